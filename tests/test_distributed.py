"""Sort-last composite on two ranks over gloo (CPU): the host-side partition / composite /
row-split logic of renderer_amd.distributed with the CPU oracle standing in for the GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleAdapter:
    """Gives the oracle renderer the buffer-owning shape of the HIP renderer."""

    def __init__(self, orc, scene, w, h):
        self.o = orc.OracleRenderer(w, h)
        self.o.set_scene(scene)
        self.vis = torch.empty(w * h, dtype=torch.int64)
        self.color = torch.zeros(h, w, 4, dtype=torch.uint8)
        self.rows, self.bands = (0, h), None

    def set_partition(self, lo, hi):
        self.o.set_partition(lo, hi)

    def set_partition_interleaved(self, chunk, rank, world):
        self.o.set_partition_interleaved(chunk, rank, world)

    def set_shade_rows(self, lo, hi):
        self.rows, self.bands = (lo, hi), None

    def set_shade_bands(self, band, rank, world):
        self.rows, self.bands = None, (band, rank, world)

    def render_visbuffer(self, camera=None, sun=None):
        v = self.o.render_visbuffer(camera, sun)
        self.vis.copy_(torch.from_numpy(v.reshape(-1).view(np.int64)))

    def shade(self):
        img = self.o.shade(self.vis.numpy().view(np.uint64), rows=self.rows, bands=self.bands)
        if self.bands:
            from renderer_amd.distributed import band_spans
            spans = band_spans(self.color.shape[0], *self.bands)
        else:
            spans = [self.rows]
        for lo, hi in spans:
            self.color[lo:hi] = torch.from_numpy(img[lo:hi])


def _worker(rank, world, port, out_dir, split):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from renderer_amd import synth
    from renderer_amd.distributed import SortLastFrame
    w, h = 128, 72
    scene = synth.make_atrium()
    a = _OracleAdapter(orc, scene, w, h)
    frame = SortLastFrame(a, scene.meshlet_visibility_count, h, rank, world, a.vis, a.color, split=split)
    frame.step()
    np.save(os.path.join(out_dir, f"vis{rank}.npy"), a.vis.numpy())
    np.save(os.path.join(out_dir, f"color{rank}.npy"), a.color.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,split", [(2, "interleaved"), (3, "interleaved"), (2, "contiguous"), (3, "contiguous")])
def test_sort_last_equals_single_rank(oracle, tmp_path, world, split):
    from renderer_amd import synth
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), split), nprocs=world, join=True)
    w, h = 128, 72
    o = oracle.OracleRenderer(w, h)
    o.set_scene(synth.make_atrium())
    full = o.render_visbuffer().reshape(-1).view(np.int64)
    color = o.shade()
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"vis{rank}.npy"), full)
        assert np.array_equal(np.load(tmp_path / f"color{rank}.npy"), color)


def test_interleaved_shares_partition_the_scene(oracle):
    """Every meshlet instance belongs to exactly one rank: the ranks' work counters add up to the whole frame's
    and the 64-bit minimum over their buffers is the whole frame, at chunk sizes that cut draws differently."""
    from renderer_amd import synth
    w, h = 96, 54
    scene = synth.make_field(40, 3, lods=1, alpha_fraction=0.3)
    o = oracle.OracleRenderer(w, h)
    o.set_scene(scene)
    full = o.render_visbuffer().copy()
    whole = {k: getattr(o.stats, k) for k in ("meshlets_tested", "triangles_tested", "fragments")}
    for world, chunk in ((2, 64), (3, 64), (8, 64), (5, 128)):
        acc = np.full_like(full, 0x7FFFFFFFFFFFFFFF)
        sums = dict.fromkeys(whole, 0)
        for rank in range(world):
            o.set_partition_interleaved(chunk, rank, world)
            acc = np.minimum(acc, o.render_visbuffer())
            for k in sums:
                sums[k] += getattr(o.stats, k)
        assert np.array_equal(acc, full), (world, chunk)
        assert sums == whole, (world, chunk)


def test_band_rows_tile_exactly():
    from renderer_amd.distributed import band_row_index, band_spans
    for h, band, world in ((2160, 32, 8), (72, 32, 2), (72, 32, 3), (31, 32, 4), (4320, 64, 8), (100, 32, 16)):
        rows = torch.cat([band_row_index(h, band, r, world) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(h))
        assert all(hi - lo <= band and lo % band == 0 for r in range(world) for lo, hi in band_spans(h, band, r, world))


def test_ranges_tile_exactly():
    from renderer_amd.distributed import instance_range, row_range
    for n, world in ((1000, 8), (7, 3), (0, 4), (2160, 8), (5, 8)):
        spans = [instance_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        rows = [row_range(n, r, world) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == n
