"""Dropped work is reported through the frame path itself, and the segmented queues hold what a
valid scene can put into them."""
import os

import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, RendererError, abi, synth
from renderer_amd.scene import Camera, Image, Scene, mat4_trs
from helpers import covered, world_from_fb

pytestmark = pytest.mark.gpu


def tessellated_quad_scene(nx, ny, width, height, x0, y0, x1, y1, z=5.0, alpha_mode=1, textured=True):
    """A screen-facing quad cut into nx x ny cells (two small front-facing triangles each) that covers
    the framebuffer rectangle (x0, y0)-(x1, y1); alpha-masked with a checkerboard alpha."""
    sc = Scene()
    v = np.zeros((ny + 1) * (nx + 1), abi.VERTEX)
    for j in range(ny + 1):
        for i in range(nx + 1):
            fx, fy = x0 + (x1 - x0) * i / nx, y0 + (y1 - y0) * j / ny
            k = j * (nx + 1) + i
            v[k]["p"] = (*world_from_fb(fx, fy, z, width, height), 0.0)
            v[k]["n"] = (0.0, 0.0, -1.0, 0.0)
            v[k]["uv"] = (i / nx * 3.0, j / ny * 3.0, 0.0, 0.0)
            v[k]["t"] = (1.0, 0.0, 0.0, 1.0)
    idx = []
    for j in range(ny):
        for i in range(nx):
            a, b = j * (nx + 1) + i, j * (nx + 1) + i + 1
            c, d = a + nx + 1, b + nx + 1
            idx += [a, c, b, b, c, d]
    if textured:
        tex = np.zeros((16, 16, 4), np.uint8)
        tex[..., :3] = 200
        yy, xx = np.mgrid[0:16, 0:16]
        tex[..., 3] = np.where(((xx // 4) + (yy // 4)) % 2 == 0, 255, 40)
        sc.add_image(Image([tex], srgb=False))
        sc.add_material(albedo_texture=0, emissive_factor_alpha_cutoff=(0, 0, 0, 0.5), alpha_mode=1)
    else:
        sc.add_material(emissive_factor_alpha_cutoff=(0, 0, 0, 0.5), alpha_mode=1)
    sc.add_draw(sc.append_mesh(v, [np.asarray(idx, np.uint32)]), mat4_trs(), 0, alpha_mode)
    sc.camera = Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    return sc


def _front_facing(sc, w, h, oracle):
    o = oracle.OracleRenderer(w, h, RendererConfig(cull_enabled=False))
    o.set_scene(sc)
    return o, o.render_visbuffer()


@pytest.mark.parametrize("nx,ny", [(36, 36), (44, 40), (52, 52)])
def test_many_small_alpha_masked_meshlets_fit_their_queue(oracle, nx, ny):
    w, h = 320, 200
    sc = tessellated_quad_scene(nx, ny, w, h, 20.5, 15.5, 300.5, 185.5)
    n_meshlets = int(sc.meshes[0]["lods"][0]["meshlet_count"])
    assert 33 <= n_meshlets <= 128, n_meshlets  # more meshlets than queue segments, few enough that workgroups double up unevenly
    o, ovis = _front_facing(sc, w, h, oracle)
    if covered(ovis).sum() == 0:  # wrong winding for the mask pipeline would still be drawn (cull NONE): must cover
        pytest.fail("scene covers nothing")
    g = Renderer(w, h, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    g.render()
    assert g.frame_status() == 0
    assert np.array_equal(g.read_visbuffer_u64(), ovis)
    err = np.abs(g.read_color().astype(np.int32) - o.shade().astype(np.int32))
    assert err.max() <= 2


def test_overflow_is_reported_by_the_next_render(monkeypatch):
    """A frame that drops work fails the frame path itself, not only the readbacks."""
    sc = synth.make_atrium()
    monkeypatch.setenv("RDR_DEBUG_DEFERRED_CAPACITY", "8")
    g = Renderer(256, 144, RendererConfig(cull_enabled=False))
    monkeypatch.delenv("RDR_DEBUG_DEFERRED_CAPACITY")
    g.set_scene(sc)
    g.render()  # the atrium's walls are clipped triangles: far more than 8 go to the deferred list
    assert g.frame_status() & abi.FRAME_DEFERRED_OVERFLOW
    g.render()  # frame_status() consumed the report
    g.sync()
    with pytest.raises(RendererError, match="dropped work"):
        g.render()
    with pytest.raises(RendererError, match="dropped"):
        g.read_color()
    g.close()
    # the same scene without the hook is complete
    g = Renderer(256, 144, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    g.render()
    g.render()
    assert g.frame_status() == 0
    g.read_color()
