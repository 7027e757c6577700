"""Randomised parity: triangle soups with sizes from sub-pixel to many screens, triangles behind
the camera and across the near plane, opaque and alpha-masked, at awkward resolutions.  Every one
of the three rasterisers and the clipper sees input here that the structured scenes never
produce; the visibility buffer must still be the oracle's bit for bit."""
import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, abi, synth
from renderer_amd.scene import Camera, Scene, mat4_trs

pytestmark = pytest.mark.gpu


def soup(rng, n_tris, masked, size_exponents=(-3.0, 2.0)):
    """Triangles in front of, around and behind a camera at the origin looking along +z (direction
    (0,0,-1): the view looks along -direction)."""
    sizes = 10.0 ** rng.uniform(size_exponents[0], size_exponents[1], n_tris)  # edge length: 1e-3 .. 100 world units
    centre = np.stack([rng.uniform(-6, 6, n_tris), rng.uniform(-4, 4, n_tris), rng.uniform(-3, 40, n_tris)], 1)
    tri = centre[:, None, :] + rng.normal(size=(n_tris, 3, 3)) * sizes[:, None, None]
    degenerate = rng.random(n_tris) < 0.03
    tri[degenerate, 2] = tri[degenerate, 1]                        # zero-area triangles
    v = np.zeros(3 * n_tris, abi.VERTEX)
    v["p"][:, :3] = tri.reshape(-1, 3)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True) + 1e-30
    v["n"][:, :3] = np.repeat(n, 3, 0)
    v["uv"][:, :2] = rng.uniform(-2, 3, (3 * n_tris, 2))
    v["t"] = (1, 0, 0, 1)
    sc = Scene()
    sc.env = synth.default_env()
    sc.add_material()
    mat = 0
    if masked:
        tex = sc.add_image(synth.checker_texture(64, 8, (255, 255, 255, 255), (0, 0, 0, 0)))
        mat = sc.add_material(albedo_texture=tex, alpha_mode=1, emissive_factor_alpha_cutoff=(0, 0, 0, 0.5))
    mesh = sc.append_mesh(v, [np.arange(3 * n_tris, dtype=np.uint32)])
    sc.add_draw(mesh, mat4_trs(), mat, 1 if masked else 0)
    # a second, rotated and scaled instance of the same soup
    a = rng.uniform(0, 6.28)
    q = (0.0, np.sin(a / 2), 0.0, np.cos(a / 2))
    sc.add_draw(mesh, mat4_trs((rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(5, 15)), q, rng.uniform(0.2, 1.5)), mat, 1 if masked else 0)
    sc.camera = Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    return sc


@pytest.mark.parametrize("seed,w,h,masked", [(1, 320, 180, False), (2, 317, 173, False), (3, 64, 64, True), (4, 1, 1, False),
                                             (5, 17, 3, True), (6, 1024, 64, False), (7, 250, 400, True), (8, 640, 360, False)])
def test_triangle_soup(oracle, seed, w, h, masked):
    rng = np.random.default_rng(seed)
    sc = soup(rng, 400, masked)
    for occlusion in (False, True):
        conf = RendererConfig(cull_enabled=occlusion)
        o = oracle.OracleRenderer(w, h, conf)
        o.set_scene(sc)
        g = Renderer(w, h, conf)
        g.set_scene(sc)
        for k in range(3):  # the camera turns: the occlusion state of the previous view must never leak into the result
            a = 0.4 * k
            cam = Camera((0.3 * k, 0.1 * k, -0.5 * k), (np.sin(a), 0.05 * k, -np.cos(a)))
            ovis = o.render_visbuffer(cam, (0.3, 0.8, -0.4))
            g.render(cam, (0.3, 0.8, -0.4))
            gvis = g.read_visbuffer_u64()
            assert np.array_equal(gvis, ovis), f"seed {seed} view {k} occlusion {occlusion}: {(gvis != ovis).sum()} keys differ"
            ref = o.shade()
            err = np.abs(g.read_color().astype(np.int32) - ref.astype(np.int32))
            assert err.max() <= 2, f"seed {seed} view {k}: colour error {err.max()}"
        g.close()


@pytest.mark.parametrize("masked", [False, True])
def test_more_large_triangles_than_one_round_of_the_tile_setup_takes(oracle, masked):
    """Thousands of large and clipped triangles: the tile path's set-up takes 4 096 triangles per round (four to a wave),
    so this many go through several rounds of it, and the bins hold hundreds of records per tile."""
    rng = np.random.default_rng(77)
    sc = soup(rng, 6000, masked, size_exponents=(0.3, 2.0))
    w, h = 200, 120
    o = oracle.OracleRenderer(w, h, RendererConfig(cull_enabled=False))
    o.set_scene(sc)
    g = Renderer(w, h, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    for k in range(2):
        cam = Camera((0.2 * k, 0.0, -0.3 * k), (np.sin(0.3 * k), 0.0, -np.cos(0.3 * k)))
        ovis = o.render_visbuffer(cam, (0.3, 0.8, -0.4))
        g.render(cam, (0.3, 0.8, -0.4))
        assert g.frame_status() == 0
        st = g.stats()
        assert st["triangles_big"] > 4096, st["triangles_big"]
        gvis = g.read_visbuffer_u64()
        assert np.array_equal(gvis, ovis), f"view {k}: {(gvis != ovis).sum()} keys differ"
    g.close()


def test_empty_scene_and_scene_without_visible_geometry(oracle):
    w, h = 96, 54
    sc = Scene()
    sc.env = synth.default_env()
    g = Renderer(w, h, RendererConfig(skybox=True))
    g.set_scene(sc)
    g.render()
    assert (g.read_visbuffer_u64() == abi.VIS_CLEAR).all()
    o = oracle.OracleRenderer(w, h, RendererConfig(skybox=True))
    o.set_scene(sc)
    o.render_visbuffer()
    assert np.abs(g.read_color().astype(int) - o.shade().astype(int)).max() <= 2  # all sky
    # everything behind the camera
    sc2 = synth.make_triangle_scene([(-1, -1, -5.0), (1, -1, -5.0), (0, 1, -5.0)], camera=Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0)))
    g2 = Renderer(w, h)
    g2.set_scene(sc2)
    g2.render()
    assert (g2.read_visbuffer_u64() == abi.VIS_CLEAR).all() and (g2.read_color() == 0).all()
    st = g2.stats()
    assert st["fragments"] == 0 and st["pixels_covered"] == 0
