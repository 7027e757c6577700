"""Frames in flight (SortLastPipeline) on one GPU over RCCL with a world of one rank: the stream
and buffer rotation must deliver, for every frame, exactly what a plain sequential renderer
produces.  RCCL refuses two ranks on one device, so ranks > 1 are rehearsed here with the ranks
sharing the GPU and gloo carrying the device tensors (all-reduce composite), and on the CPU by
test_distributed.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, synth
from renderer_amd.scene import Camera

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("composite", ["reduce_scatter", "all_reduce"])
def test_pipelined_frames_match_sequential(composite):
    import torch
    import torch.distributed as dist
    from renderer_amd.distributed import SortLastPipeline

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        scene = synth.make_atrium()
        w, h = 320, 180
        sun = (0.3, 0.8, -0.4)
        cams = [Camera((0.0, 1.0, 30.0), (0.0, 0.0, 1.0)), Camera((-8.0, 6.0, 0.0), (-0.7, 0.3, 0.6)),
                Camera((3.0, 0.2, -10.0), (0.3, -0.5, -0.8)), Camera((0.0, 9.5, 0.0), (0.1, 0.99, 0.05)),
                Camera((0.0, 1.0, 30.0), (0.0, 0.0, 1.0))]
        ref = Renderer(w, h, RendererConfig())
        ref.set_scene(scene)
        expected = []
        for cam in cams:
            ref.render(cam, sun)
            expected.append((ref.read_visbuffer_u64().copy(), ref.read_color().copy()))
        ref.close()

        def make():
            r = Renderer(w, h, RendererConfig())
            r.set_scene(scene)
            return r

        pipe = SortLastPipeline(make, scene.meshlet_visibility_count, h, 0, 1, depth=2, composite=composite)
        assert pipe.composite == composite
        for first in range(0, len(cams) - 1, 2):  # two frames in flight, then look at both
            frames = [pipe.step(cams[first + k], sun) for k in range(2)]
            torch.cuda.synchronize()
            assert frames[0] is not frames[1]
            for k, f in enumerate(frames):
                vis = f.vis.cpu().numpy().view(np.uint64).reshape(h, w)
                col = f.color.cpu().numpy().reshape(h, w, 4)
                assert np.array_equal(vis, expected[first + k][0])
                assert np.array_equal(col, expected[first + k][1])
        pipe.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_composite_to_the_single_gpu_image(tmp_path, world):
    """Every rank rasterises its share of the scene on the GPU, the keys are MIN-reduced, each rank
    shades its rows and gathers the rest: all ranks must end up with the single-GPU image, for
    frames in flight and for a reused slot."""
    import sort_last_worker as wk

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, wk.__file__, str(r), str(world), str(port), str(tmp_path)])
             for r in range(world)]
    try:
        codes = [p.wait(timeout=100) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world
    ref = Renderer(wk.W, wk.H, RendererConfig())
    ref.set_scene(synth.make_atrium())
    for k, cam in enumerate(wk.CAMERAS):
        ref.render(Camera(*cam), wk.SUN)
        expected = ref.read_color().reshape(wk.H, wk.W, 4)
        for r in range(world):
            got = np.load(tmp_path / f"color{r}_{k}.npy").reshape(wk.H, wk.W, 4)
            assert np.array_equal(got, expected), (k, r)
    ref.close()
