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
        self.rows = (0, h)

    def set_partition(self, lo, hi):
        self.o.set_partition(lo, hi)

    def set_shade_rows(self, lo, hi):
        self.rows = (lo, hi)

    def render_visbuffer(self, camera=None, sun=None):
        v = self.o.render_visbuffer(camera, sun)
        self.vis.copy_(torch.from_numpy(v.reshape(-1).view(np.int64)))

    def shade(self):
        img = self.o.shade(self.vis.numpy().view(np.uint64), rows=self.rows)
        lo, hi = self.rows
        self.color[lo:hi] = torch.from_numpy(img[lo:hi])


def _worker(rank, world, port, out_dir):
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
    frame = SortLastFrame(a, scene.meshlet_visibility_count, h, rank, world, a.vis, a.color)
    frame.step()
    np.save(os.path.join(out_dir, f"vis{rank}.npy"), a.vis.numpy())
    np.save(os.path.join(out_dir, f"color{rank}.npy"), a.color.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sort_last_equals_single_rank(oracle, tmp_path, world):
    from renderer_amd import synth
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    w, h = 128, 72
    o = oracle.OracleRenderer(w, h)
    o.set_scene(synth.make_atrium())
    full = o.render_visbuffer().reshape(-1).view(np.int64)
    color = o.shade()
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"vis{rank}.npy"), full)
        assert np.array_equal(np.load(tmp_path / f"color{rank}.npy"), color)


def test_ranges_tile_exactly():
    from renderer_amd.distributed import instance_range, row_range
    for n, world in ((1000, 8), (7, 3), (0, 4), (2160, 8), (5, 8)):
        spans = [instance_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        rows = [row_range(n, r, world) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == n
