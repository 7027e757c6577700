"""The tile path as one launch (k_tile_path_rare): frames that follow a frame without large or mid-sized triangles
launch a single kernel in place of the three tile-path kernels; it leaves at once when the frame has no such work
and otherwise runs the same three steps behind grid-wide barriers.  Either way the buffers must be the oracle's."""
import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, synth
from renderer_amd.scene import Camera
from test_gpu_parity import STAT_KEYS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("occlusion", [False, True])
def test_forced_one_launch_tile_path_is_exact(oracle, monkeypatch, occlusion):
    """RDR_TILE_PATH_RARE=2: every frame takes the one-launch form, on scenes that keep all three steps busy (clipped
    hall triangles, alpha-tested fence, mid-sized sphere triangles)."""
    monkeypatch.setenv("RDR_TILE_PATH_RARE", "2")
    for sc, size in ((synth.make_atrium(), (333, 187)), (synth.make_field(16, 2, lods=1, alpha_fraction=0.5), (640, 360))):
        if len(sc.mesh_draws) == 16:
            sc.camera = Camera((0.0, 1.5, -6.0), (0.0, 0.2, -1.0))
        w, h = size
        conf = RendererConfig(cull_enabled=occlusion)
        o = oracle.OracleRenderer(w, h, conf)
        o.set_scene(sc)
        ovis = o.render_visbuffer()
        g = Renderer(w, h, conf)
        g.set_scene(sc)
        for _ in range(3):
            g.render()
            assert g.frame_status() == 0
            assert np.array_equal(g.read_visbuffer_u64(), ovis)
        if not occlusion:
            st, od = g.stats(), o.stats.as_dict()
            assert all(st[k] == od[k] for k in STAT_KEYS), (st, od)
        err = np.abs(g.read_color().astype(np.int32) - o.shade().astype(np.int32))
        assert err.max() <= 2
        g.close()


def test_frames_with_and_without_large_triangles_alternate(oracle):
    """Default mode: the launch form follows the last finished frame, so a view change from 'all small' to 'large
    triangles' is first met by the one-launch form (its slow path), then by the three launches, and back.  Every
    frame must be exact, and the work counters must not keep another frame's values."""
    sc = synth.make_field(30, 3, lods=1)
    w, h = 480, 270
    far_cam = Camera((0.0, 25.0, -60.0), (0.0, 0.35, -1.0))   # every triangle a few pixels at most
    # a third of a unit in front of one sphere's surface: its triangles are dozens of pixels across (too large for a
    # lane, many too large for a wave), some cross the near plane
    t = np.asarray(sc.global_transforms, np.float32).reshape(-1, 16)[int(sc.mesh_draws[len(sc.mesh_draws) // 2]["model_index"])]
    centre, scale = t[12:15].astype(float), float(np.linalg.norm(t[0:3]))
    near_cam = Camera((centre[0], centre[1], centre[2] - scale - 0.3), (0.0, 0.0, -1.0))
    conf = RendererConfig(cull_enabled=False)
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    expect = {}
    for name, cam in (("far", far_cam), ("near", near_cam)):
        vis = o.render_visbuffer(cam).copy()
        expect[name] = (cam, vis, o.stats.as_dict())
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    for name in ("far", "far", "far", "near", "near", "near", "far", "far", "near", "far", "near", "near"):
        cam, vis, od = expect[name]
        g.render(cam)
        g.sync()  # the next frame's launch form follows this frame's finished status
        assert g.frame_status() == 0
        assert np.array_equal(g.read_visbuffer_u64(), vis), name
        st = g.stats()
        assert all(st[k] == od[k] for k in STAT_KEYS), (name, {k: (st[k], od[k]) for k in STAT_KEYS})
        assert (st["triangles_big"] > 0) == (name == "near"), (name, st["triangles_big"])  # the views are what they are meant to be
    g.close()
