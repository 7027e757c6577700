"""Known-answer tests that pin the CPU oracle: hand-derived coverage, depth order, culling and
uniform values.  The reference ships no tests or golden vectors for this path, so these are the
anchors ("parity unpinned" otherwise)."""
import ctypes as C
import math

import numpy as np
import pytest

from renderer_amd import abi, synth
from renderer_amd.scene import Camera, RendererConfig, mat4_trs
from helpers import H_45, covered, fb_triangle_scene, key_id


def render(oracle, scene, w, h, **cfg):
    o = oracle.OracleRenderer(w, h, RendererConfig(**cfg) if cfg else None)
    o.set_scene(scene)
    vis = o.render_visbuffer()
    return o, vis


def test_ubo_known_values(oracle):
    u = oracle.build_ubo(3840, 2160, (1.0, 2.0, 3.0), (0.0, 0.0, -1.0), (0.1, 0.9, 0.2), 7, True, False)
    view = np.array(u.view[:]).reshape(4, 4).T  # row-major view of the column-major data
    # direction (0,0,-1): the camera looks along +z, basis is the identity
    assert np.allclose(view[:3, :3], np.eye(3), atol=1e-7)
    assert np.allclose(view[:3, 3], (-1.0, -2.0, -3.0), atol=1e-6)
    proj = np.array(u.proj[:]).reshape(4, 4).T
    w = H_45 / (3840 / 2160)
    assert proj[0, 0] == pytest.approx(-w, rel=1e-6) and proj[1, 1] == pytest.approx(H_45, rel=1e-6)
    assert proj[3, 2] == 1.0 and proj[2, 3] == pytest.approx(0.01) and proj[2, 2] == 0.0
    assert u.frustum[0] == pytest.approx(-w / math.hypot(w, 1), rel=1e-6)  # negative: the x flip
    assert u.frustum[1] == pytest.approx(1 / math.hypot(w, 1), rel=1e-6)
    assert u.frustum[2] == pytest.approx(H_45 / math.hypot(H_45, 1), rel=1e-6)
    assert u.frustum[3] == pytest.approx(1 / math.hypot(H_45, 1), rel=1e-6)
    assert (u.near_plane, u.far_plane) == (pytest.approx(0.01), 500.0)
    assert (u.depth_pyramid_width, u.depth_pyramid_height) == (2048.0, 2048.0)
    assert u.lod_target == pytest.approx(1 / (H_45 * 2160), rel=1e-6)
    assert (u.draw_count, u.lod_enabled, u.cull_enabled) == (7, 1, 0)
    assert (u.screen_width, u.screen_height) == (3840.0, 2160.0)
    assert tuple(u.sun_direction[:]) == pytest.approx((0.1, 0.9, 0.2, 0.0))
    assert tuple(u.camera_position[:]) == (1.0, 2.0, 3.0)
    # power_2_floor is "largest power of two strictly below" (renderer/src/math.rs:17-25)
    for dim, expect in ((1280, 1024), (720, 512), (4096, 2048), (2049, 2048), (3, 2)):
        assert oracle.build_ubo(dim, dim, (0, 0, 0), (0, 0, -1), (0, 1, 0), 0).depth_pyramid_width == expect


def test_single_triangle_exact_coverage(oracle):
    W = H = 64
    tri = [(8, 8, 5.0), (8, 40, 5.0), (40, 8, 5.0)]
    sc = fb_triangle_scene([tri], W, H)
    o, vis = render(oracle, sc, W, H)
    cov = covered(vis)
    py, px = np.mgrid[0:H, 0:W]
    # pixel centres inside x>=8, y>=8, x+y<48; the hypotenuse is a bottom/right edge (excluded)
    expect = (px >= 8) & (py >= 8) & (px + py + 1 < 48)
    assert expect.sum() == 496
    assert np.array_equal(cov, expect)
    assert o.stats.fragments == 496 and o.stats.triangles_rasterised == 1
    # depth: z/w = near/z = 0.01/5 everywhere, id = instance 0, triangle 0
    hi = (vis[cov] >> np.uint64(32)).astype(np.uint32)
    z = (np.uint32(0x7FFFFFFF) - hi).view(np.float32)
    assert np.allclose(z, 0.002, rtol=1e-5)
    assert (key_id(vis)[cov] == 0).all()
    xy = o.read_visbuffer()
    assert (xy[cov] == (0, 0)).all() and (xy[~cov] == 0xFFFFFFFF).all()


def test_backface_is_culled(oracle):
    W = H = 64
    sc = fb_triangle_scene([[(8, 8, 5.0), (40, 8, 5.0), (8, 40, 5.0)]], W, H)  # opposite winding
    o, vis = render(oracle, sc, W, H)
    assert not covered(vis).any() and o.stats.triangles_rasterised == 0


def test_shared_edge_covers_each_pixel_once(oracle):
    W = H = 64
    a, b, c, d = (5.25, 7.5, 4.0), (50.75, 11.0, 4.0), (47.0, 55.5, 4.0), (9.5, 49.0, 4.0)
    sc = fb_triangle_scene([[a, d, c], [a, c, b]], W, H)
    o, vis = render(oracle, sc, W, H)
    cov = covered(vis)
    assert o.stats.triangles_rasterised == 2
    assert o.stats.fragments == cov.sum()  # no pixel hit twice, none missed along the diagonal
    ids = key_id(vis)[cov]
    assert set(np.unique(ids)) == {0, 1}


def test_top_left_rule_on_pixel_centres(oracle):
    W = H = 32
    # axis-aligned rectangle with edges exactly through pixel centres
    x0, y0, x1, y1 = 4.5, 6.5, 20.5, 14.5
    a, b, c, d = (x0, y0, 3.0), (x1, y0, 3.0), (x1, y1, 3.0), (x0, y1, 3.0)
    sc = fb_triangle_scene([[a, d, c], [a, c, b]], W, H)
    o, vis = render(oracle, sc, W, H)
    cov = covered(vis)
    py, px = np.mgrid[0:H, 0:W]
    expect = (px >= 4) & (px < 20) & (py >= 6) & (py < 14)  # left/top edges in, right/bottom out
    assert np.array_equal(cov, expect)


def test_nearer_triangle_wins(oracle):
    W = H = 64
    far = [(4, 4, 9.0), (4, 60, 9.0), (60, 4, 9.0)]
    near = [(10, 10, 3.0), (10, 40, 3.0), (40, 10, 3.0)]
    for order in ((far, near), (near, far)):
        sc = fb_triangle_scene(list(order), W, H)
        o, vis = render(oracle, sc, W, H)
        ids = key_id(vis)
        near_id = order.index(near)
        assert ids[15, 15] == near_id
        assert ids[5, 30] == 1 - near_id
        assert covered(vis).sum() == covered(render(oracle, fb_triangle_scene([far], W, H), W, H)[1]).sum()


def test_equal_depth_tie_goes_to_lower_id(oracle):
    W = H = 32
    t = [(4, 4, 5.0), (4, 28, 5.0), (28, 4, 5.0)]
    sc = fb_triangle_scene([t, t], W, H)
    o, vis = render(oracle, sc, W, H)
    assert (key_id(vis)[covered(vis)] == 0).all()


def test_subpixel_sliver_is_culled(oracle):
    W = H = 64
    sc = fb_triangle_scene([[(10.6, 10.6, 5.0), (10.6, 10.9, 5.0), (10.9, 10.6, 5.0)]], W, H)
    o, vis = render(oracle, sc, W, H)
    assert o.stats.triangles_tested == 1 and o.stats.triangles_rasterised == 0
    assert not covered(vis).any()


def test_near_plane_crossing_triangle_is_clipped(oracle):
    W, H = 64, 36
    sc = synth.make_triangle_scene([(-50.0, -1.0, -20.0), (50.0, -1.0, -20.0), (0.0, -1.0, 200.0)],
                                   Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0)))
    o, vis = render(oracle, sc, W, H)
    # the ground plane y=-1 seen from above its back side?  facing decides; try both windings
    sc2 = synth.make_triangle_scene([(50.0, -1.0, -20.0), (-50.0, -1.0, -20.0), (0.0, -1.0, 200.0)],
                                    Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0)))
    o2, vis2 = render(oracle, sc2, W, H)
    cov = covered(vis) | covered(vis2)
    assert covered(vis).any() != covered(vis2).any()  # exactly one winding is front-kept
    assert max(o.stats.triangles_big, o2.stats.triangles_big) == 1
    # the plane lies below the eye: bottom rows are covered, everything above the horizon is not
    assert cov[H - 1, W // 2] and cov[H - 1, 2] and cov[H - 1, W - 3]
    assert not cov[: H // 2 - 1].any()
    # depth increases (reverse-Z) towards the bottom of the screen
    v = vis if covered(vis).any() else vis2
    col = v[:, W // 2]
    rows = np.nonzero(covered(col))[0]
    hi = (col[rows] >> np.uint64(32)).astype(np.int64)
    assert (np.diff(hi) < 0).all()  # key high word falls as depth rises


def test_far_plane_sphere_reject(oracle):
    W, H = 64, 36
    v, idx, err = synth.icosphere(1, 1)

    def scene_at(z):
        sc = synth.Scene()
        sc.add_material()
        sc.add_draw(sc.append_mesh(v, idx, err), mat4_trs((0, 0, z), (0, 0, 0, 1), 2.0), 0)
        sc.camera = Camera((0, 0, 0), (0, 0, -1))
        return sc
    # centre.z + radius < far with radius 2 (unit sphere scaled by 2)
    o, vis = render(oracle, scene_at(497.9), W, H)
    assert o.stats.draws_visible == 1
    o, vis = render(oracle, scene_at(498.1), W, H)
    assert o.stats.draws_visible == 0 and not covered(vis).any()


def test_alpha_mask(oracle):
    W = H = 64
    quad = [[(8, 8, 4.0), (8, 56, 4.0), (56, 56, 4.0)], [(8, 8, 4.0), (56, 56, 4.0), (56, 8, 4.0)]]
    # constant alpha below the cutoff: everything discarded
    sc = fb_triangle_scene(quad, W, H, alpha_mode=1,
                           material=dict(base_color_factor=(1, 1, 1, 0.4), emissive_factor_alpha_cutoff=(0, 0, 0, 0.5)))
    o, vis = render(oracle, sc, W, H)
    assert o.stats.triangles_rasterised == 2 and not covered(vis).any()
    # above the cutoff: the full quad
    sc = fb_triangle_scene(quad, W, H, alpha_mode=1,
                           material=dict(base_color_factor=(1, 1, 1, 0.6), emissive_factor_alpha_cutoff=(0, 0, 0, 0.5)))
    o, vis = render(oracle, sc, W, H)
    assert covered(vis).sum() == 48 * 48
    # draws without a material never discard (material reads as zeros: 0 < 0 is false)
    sc = fb_triangle_scene(quad, W, H, alpha_mode=1)
    sc.mesh_draws["material_index"] = 0xFFFF
    o, vis = render(oracle, sc, W, H)
    assert covered(vis).sum() == 48 * 48
    # textured: alpha comes from the checkerboard, roughly half of the quad survives
    tex = synth.checker_texture(64, 4, (255, 255, 255, 255), (255, 255, 255, 0))
    sc = fb_triangle_scene(quad, W, H, alpha_mode=1, image=tex,
                           material=dict(albedo_texture=0, emissive_factor_alpha_cutoff=(0, 0, 0, 0.5)))
    o, vis = render(oracle, sc, W, H)
    frac = covered(vis).sum() / (48 * 48)
    assert 0.3 < frac < 0.7


def test_mask_draws_bypass_the_draw_cull(oracle):
    W, H = 64, 36
    v, idx, err = synth.icosphere(1, 1)
    sc = synth.Scene()
    sc.add_material()
    m = sc.append_mesh(v, idx, err)
    sc.add_draw(m, mat4_trs((0, 0, 600.0)), 0, alpha_mode=0)
    sc.add_draw(m, mat4_trs((0, 0, 600.0)), 0, alpha_mode=1)
    sc.camera = Camera((0, 0, 0), (0, 0, -1))
    o, vis = render(oracle, sc, W, H)
    # the opaque draw is rejected; the mask draw still emits commands (cull.comp.glsl:99),
    # its meshlets are then rejected by the per-meshlet far test
    assert o.stats.draws_visible == 1 and o.stats.meshlets_tested > 0 and o.stats.meshlets_visible == 0


def test_lod_selection(oracle):
    W, H = 640, 360
    sc = synth.make_field(64, 3, lods=3, with_env=False)
    sc.camera = Camera((0.0, 2.0, -9.0), (0.0, 0.196, -0.98))
    o, vis = render(oracle, sc, W, H, lod_enabled=True)
    lods = o.draw_lod[: len(sc.mesh_draws)]
    assert len(set(np.unique(lods)) - {0xFF}) >= 2
    xy = o.read_visbuffer()
    cov = covered(vis)
    draw = xy[..., 1][cov]
    meshlet = xy[..., 0][cov] >> 8
    L = sc.meshes[0]["lods"]
    lo = L["meshlet_offset"][lods[draw]]
    assert ((meshlet >= lo) & (meshlet < lo + L["meshlet_count"][lods[draw]])).all()
    o0, vis0 = render(oracle, sc, W, H, lod_enabled=False)
    assert np.isin(o0.draw_lod[: len(sc.mesh_draws)], (0, 0xFF)).all()
    assert o.stats.triangles_submitted < o0.stats.triangles_submitted


def test_partition_min_equals_full(oracle):
    W, H = 160, 90
    sc = synth.make_atrium()
    o, full = render(oracle, sc, W, H)
    n = sc.meshlet_visibility_count
    parts = []
    for lo, hi in ((0, n // 3), (n // 3, 2 * n // 3), (2 * n // 3, n)):
        o.set_partition(lo, hi)
        parts.append(o.render_visbuffer().copy())
    assert np.array_equal(np.minimum.reduce(parts), full)
    # bit 63 is never set: signed and unsigned min agree
    assert (np.stack(parts) >> np.uint64(63) == 0).all()
    assert np.array_equal(np.minimum.reduce([p.view(np.int64) for p in parts]).view(np.uint64), full)
