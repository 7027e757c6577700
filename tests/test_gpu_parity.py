"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Bar: the 64-bit visibility buffer and the resolved ID texels are bit-exact; the shaded RGBA8
image is within max-abs 2/255 and PSNR >= 40 dB (the tolerance the task states for the shaded
output; in practice the only differences are ulp-level libm vs device transcendental results)."""
import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, RendererError, abi, synth
from renderer_amd.scene import Camera, mat4_trs
from helpers import covered, fb_triangle_scene, key_id, psnr

pytestmark = pytest.mark.gpu

STAT_KEYS = ["draws_visible", "meshlets_tested", "meshlets_visible", "triangles_submitted", "triangles_tested",
             "triangles_rasterised", "fragments"]


def both(oracle, scene, w, h, shade=True, **cfg):
    conf = RendererConfig(**cfg)
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(scene)
    ovis = o.render_visbuffer()
    g = Renderer(w, h, conf)
    g.occlusion = bool(conf.cull_enabled)
    g.set_scene(scene)
    for _ in range(2 if g.occlusion else 1):  # second frame: the early phase has last frame's visibility to work with
        if shade:
            g.render()
        else:
            g.render_visbuffer()
    return o, ovis, g


def check_vis(o, ovis, g, stats=None):
    gvis = g.read_visbuffer_u64()
    diff = int((gvis != ovis).sum())
    assert diff == 0, f"{diff} of {ovis.size} visibility-buffer keys differ"
    assert np.array_equal(g.read_visbuffer(), o.read_visbuffer())
    st = g.stats()
    od = o.stats.as_dict()
    # with occlusion culling on, the work counters legitimately differ from the single-pass oracle
    occlusion = getattr(g, "occlusion", True)
    compare = (not occlusion) if stats is None else stats
    for k in STAT_KEYS:
        if compare:
            assert st[k] == od[k], (k, st[k], od[k])
    assert st["fragments"] <= od["fragments"] or not occlusion


def check_color(o, g, covered_count=None):
    ref = o.shade()
    img = g.read_color()
    err = np.abs(img.astype(np.int32) - ref.astype(np.int32))
    assert err.max() <= 2, f"max abs error {err.max()} / 255"
    assert psnr(img, ref) >= 40.0
    if covered_count is not None:
        assert g.stats()["pixels_covered"] == covered_count


@pytest.mark.parametrize("occlusion", [False, True])
@pytest.mark.parametrize("w,h", [(64, 36), (256, 144), (333, 187)])
def test_atrium(oracle, w, h, occlusion):
    sc = synth.make_atrium()
    o, ovis, g = both(oracle, sc, w, h, cull_enabled=occlusion)
    check_vis(o, ovis, g)
    check_color(o, g, int(covered(ovis).sum()))


def test_atrium_other_views(oracle):
    sc = synth.make_atrium()
    w, h = 200, 120
    conf = RendererConfig()
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    g = Renderer(w, h, conf)
    g.occlusion = bool(conf.cull_enabled)
    g.set_scene(sc)
    views = [Camera((0.0, 1.0, 30.0), (0.0, 0.0, 1.0)), Camera((-8.0, 6.0, 0.0), (-0.7, 0.3, 0.6)),
             Camera((3.0, 0.2, -10.0), (0.3, -0.5, -0.8)), Camera((0.0, 9.5, 0.0), (0.1, 0.99, 0.05))]
    for cam in views:
        ovis = o.render_visbuffer(cam, (0.3, 0.8, -0.4))
        g.render(cam, (0.3, 0.8, -0.4))
        check_vis(o, ovis, g)
        check_color(o, g)


@pytest.mark.parametrize("occlusion", [False, True])
@pytest.mark.parametrize("lod", [False, True])
def test_field(oracle, lod, occlusion):
    sc = synth.make_field(150, 4, lods=3)
    o, ovis, g = both(oracle, sc, 480, 270, lod_enabled=lod, cull_enabled=occlusion)
    check_vis(o, ovis, g)
    check_color(o, g)


def test_field_close_up_medium_triangles(oracle):
    sc = synth.make_field(16, 2, lods=1)
    sc.camera = Camera((0.0, 1.5, -6.0), (0.0, 0.2, -1.0))
    o, ovis, g = both(oracle, sc, 640, 360)
    check_vis(o, ovis, g)
    check_color(o, g)


def test_alpha_masked_field(oracle):
    sc = synth.make_field(60, 3, alpha_fraction=0.5)
    sc.camera = Camera((0.0, 4.0, -14.0), (0.0, 0.25, -1.0))
    o, ovis, g = both(oracle, sc, 320, 180)
    check_vis(o, ovis, g)
    check_color(o, g)


def test_many_materials_in_one_wave(oracle):
    """The shading pass takes the materials of a wave of 64 pixels one at a time.  Small instances with thirteen
    different materials (all material models among them) in turn put four and more materials into most waves, and an
    index past the end of the material table shades with the all-zero material."""
    sc = synth.make_field(1, 3, lods=1, with_env=True)
    rng = np.random.default_rng(11)
    kinds = [abi.MAT_METAL_ROUGHNESS, abi.MAT_SPEC_GLOSS, abi.MAT_METAL_ROUGHNESS | abi.MAT_CLEARCOAT,
             abi.MAT_METAL_ROUGHNESS | abi.MAT_SHEEN, abi.MAT_METAL_ROUGHNESS | abi.MAT_SPEC]
    for k in range(12):
        sc.add_material(material_type=kinds[k % len(kinds)], base_color_factor=tuple(rng.uniform(0.1, 1.0, 3)) + (1.0,),
                        metallic_roughness_normal_occlusion=(float(rng.uniform(0, 1)), float(rng.uniform(0.1, 1)), 1, 1),
                        clearcoat_factor=(0.7, 0.2, 0, 0), sheen_color_roughness_factor=(0.8, 0.5, 0.7, 0.5),
                        specular_factor=(1.0, 0.6, 0.5, 0.7), specular_glossiness_factor=(0.3, 0.4, 0.35, 0.6))
    side = 40
    for i in range(side * side):
        x, z = (i % side - side / 2) * 0.22, (i // side) * 0.22
        material = 0xFFFF if i % 17 == 0 else i % 13
        sc.add_draw(0, mat4_trs((x, 0.0, z), scale=(0.1, 0.1, 0.1)), material)
    sc.camera = Camera((0.0, 3.0, -2.5), (0.0, 0.45, -1.0))
    o, ovis, g = both(oracle, sc, 480, 270)
    assert covered(ovis).sum() > 20000
    check_vis(o, ovis, g)
    check_color(o, g)


def test_vertex_fetch_through_the_indices_is_result_neutral(oracle, monkeypatch):
    """The rasteriser reads a meshlet's positions and uv sets from copies laid out in meshlet_vertices order; without
    them (RDR_RASTER_INDEXED=1) it goes through the indices: the same buffer, opaque and alpha-masked."""
    monkeypatch.setenv("RDR_RASTER_INDEXED", "1")
    sc = synth.make_field(60, 3, alpha_fraction=0.5)
    sc.camera = Camera((0.0, 4.0, -14.0), (0.0, 0.25, -1.0))
    o, ovis, g = both(oracle, sc, 320, 180)
    check_vis(o, ovis, g)
    check_color(o, g)
    o, ovis, g = both(oracle, synth.make_atrium(), 256, 144, shade=False)
    check_vis(o, ovis, g)


def test_meshes_that_share_meshlets(oracle):
    """Two meshes over the same meshlets and indices but different vertices (a second vertex_offset): an entry of
    meshlet_vertices then names two vertices, the copies in meshlet order cannot exist, and the library has to notice
    that by itself and fetch through the indices."""
    sc = synth.make_field(1, 3, lods=2)
    first = sc.meshes[0].copy()
    twin = first.copy()
    twin["vertex_offset"] = len(sc.vertices)
    twin["index_offset"] = len(sc.indices)
    verts = sc.vertices[int(first["vertex_offset"]):int(first["vertex_offset"]) + int(first["vertex_count"])].copy()
    verts["p"][:, :3] *= np.float32(1.4)  # a different shape, so that fetching the first mesh's vertices would show
    verts["p"][:, 1] *= np.float32(0.6)
    twin["radius"] = first["radius"] * np.float32(1.4)
    sc.vertices = np.concatenate([sc.vertices, verts])
    lo = int(first["index_offset"])
    sc.indices = np.concatenate([sc.indices, sc.indices[lo:]])
    sc.meshes = np.concatenate([sc.meshes, twin.reshape(1)])
    sc.add_draw(1, mat4_trs((2.5, 0.0, 0.0)))
    sc.add_draw(0, mat4_trs((-2.5, 0.5, 1.0)))
    sc.add_draw(1, mat4_trs((0.0, 2.0, 3.0), scale=(0.7, 0.7, 0.7)))
    sc.camera = Camera((0.0, 1.0, -9.0), (0.0, 0.0, -1.0))
    o, ovis, g = both(oracle, sc, 320, 180)
    assert covered(ovis).sum() > 2000
    check_vis(o, ovis, g)
    check_color(o, g)


@pytest.mark.parametrize("lane_max,mid_max", [(1, 1), (1, 4096), (4096, 4096), (16, 64)])
def test_raster_routing_is_result_neutral(oracle, monkeypatch, lane_max, mid_max):
    """Which of the three rasterisers (per-lane scan, wave per triangle, tile bins) takes a
    triangle is a performance choice: every routing yields the oracle's buffer, opaque and
    alpha-masked, close up (large triangles) and from afar."""
    monkeypatch.setenv("RDR_LANE_MAX_PIXELS", str(lane_max))
    monkeypatch.setenv("RDR_MID_MAX_PIXELS", str(mid_max))
    sc = synth.make_field(40, 2, alpha_fraction=0.5)
    sc.camera = Camera((0.0, 2.0, -8.0), (0.0, 0.2, -1.0))
    o, ovis, g = both(oracle, sc, 480, 270, shade=False)
    check_vis(o, ovis, g)
    sc = synth.make_atrium()
    o, ovis, g = both(oracle, sc, 256, 144, shade=False)
    check_vis(o, ovis, g)


def test_occlusion_culling_over_a_camera_path(oracle):
    """The early phase draws what the previous frame's late phase found visible, so the state
    carried between frames is exercised by moving the camera (and the sun-independent LOD
    choice) every frame: each frame's buffer must still be the oracle's, bit for bit."""
    sc = synth.make_field(200, 3, lods=3, alpha_fraction=0.2)
    w, h = 400, 225
    conf = RendererConfig(lod_enabled=True, cull_enabled=True)
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    g = Renderer(w, h, conf)
    g.occlusion = True
    g.set_scene(sc)
    for k in range(10):
        a = 0.35 * k
        cam = Camera((18.0 * np.sin(a), 1.2 + 0.8 * k, -18.0 * np.cos(a)), (np.sin(a), 0.08 * k, -np.cos(a)))
        ovis = o.render_visbuffer(cam, (0.3, 0.8, -0.4))
        g.render_visbuffer(cam, (0.3, 0.8, -0.4))
        assert np.array_equal(g.read_visbuffer_u64(), ovis), f"frame {k}"
        if k in (3, 7):  # and again from the same place: now everything visible comes from the early phase
            g.render_visbuffer(cam, (0.3, 0.8, -0.4))
            assert np.array_equal(g.read_visbuffer_u64(), ovis), f"frame {k} repeated"


def test_occlusion_culling_skips_hidden_meshlets(oracle):
    """A wall in front of the field: same image as without culling, far fewer meshlets rasterised."""
    sc = synth.make_field(300, 3, lods=1)
    sc.camera = Camera((0.0, 1.0, -40.0), (0.0, 0.0, -1.0))
    o = oracle.OracleRenderer(480, 270, RendererConfig())
    o.set_scene(sc)
    base = o.render_visbuffer()
    fwd = np.array([o.ubo.view[2], o.ubo.view[6], o.ubo.view[10]], np.float64)  # the view matrix's forward row
    right, up = np.array([o.ubo.view[0], o.ubo.view[4], o.ubo.view[8]]), np.array([o.ubo.view[1], o.ubo.view[5], o.ubo.view[9]])
    c = np.array(sc.camera.position) + 6.0 * fwd
    quad = [c - 40 * right - 1.0 * up, c + 40 * right - 1.0 * up, c + 40 * right + 30 * up, c - 40 * right + 30 * up]
    wall = synth.make_triangle_scene([quad[0], quad[1], quad[2]], extra_tris=[[quad[0], quad[2], quad[3]], [quad[0], quad[2], quad[1]],
                                                                               [quad[0], quad[3], quad[2]]])
    mesh = sc.append_mesh(wall.vertices, [wall.indices])
    sc.add_draw(mesh, np.eye(4, dtype=np.float32).reshape(-1), 0, 0)
    results = {}
    for occl in (False, True):
        conf = RendererConfig(cull_enabled=occl)
        oo = oracle.OracleRenderer(480, 270, conf)
        oo.set_scene(sc)
        ovis = oo.render_visbuffer()
        g = Renderer(480, 270, conf)
        g.set_scene(sc)
        for _ in range(3):
            g.render_visbuffer()
        assert np.array_equal(g.read_visbuffer_u64(), ovis)
        results[occl] = g.stats()
    assert not np.array_equal(ovis, base)  # the wall is in the picture
    assert results[True]["meshlets_visible"] < 0.5 * results[False]["meshlets_visible"]
    assert results[True]["triangles_rasterised"] < 0.5 * results[False]["triangles_rasterised"]


def test_shadow_deep_stack_path(oracle, monkeypatch):
    """Rays whose traversal overflows the LDS stack are re-traced with the deep stack."""
    monkeypatch.setenv("RDR_DEBUG_STACK_LIMIT", "3")
    sc = synth.make_atrium()
    o, ovis, g = both(oracle, sc, 256, 144)
    check_vis(o, ovis, g)
    check_color(o, g)


@pytest.mark.parametrize("lods,stack", [(1, None), (6, None), (1, "2")])
def test_shadow_walk_kernel_equals_inline_walk(monkeypatch, lods, stack):
    """The walks through a ray's own closed convex instance are left to k_shadow_walk by default; with
    RDR_SHADOW_DEFER=0 they stay inside k_shadow.  Same shadow for every pixel either way, also where the pixel
    shows a coarser level of detail than the ray structure holds and where the LDS stack overflows."""
    sc = synth.make_field(300, 4, lods=lods, with_env=True, prepared=True)
    conf = RendererConfig(lod_enabled=lods > 1, cull_enabled=False)
    if stack:
        monkeypatch.setenv("RDR_DEBUG_STACK_LIMIT", stack)
    frames = {}
    for defer in ("1", "0"):
        monkeypatch.setenv("RDR_SHADOW_DEFER", defer)
        g = Renderer(1280, 720, conf)
        g.set_scene(sc)
        g.render()
        frames[defer] = (g.read_visbuffer_u64(), g.read_color())
        assert g.frame_status() == 0
        g.close()
    assert np.array_equal(frames["1"][0], frames["0"][0])
    assert np.array_equal(frames["1"][1], frames["0"][1])
    assert covered(frames["1"][0]).sum() > 20000


def test_sun_candidates_follow_sun_and_transforms(oracle, monkeypatch):
    """The per-instance lists of possible sun occluders are rebuilt when the sun direction or the transforms
    change: frames under a low sun (long shadows across neighbours), a high sun, the low sun again and moved
    instances all match the CPU check, and match the frames rendered with the lists switched off."""
    sc = synth.make_field(200, 3, lods=1, with_env=True, prepared=True)
    w, h = 480, 270
    conf = RendererConfig(lod_enabled=False, cull_enabled=False)
    suns = [(0.9, 0.25, 0.2), (-0.23623444, 0.9302103, 0.2808959), (0.9, 0.25, 0.2), (-0.3, 0.2, 0.9)]
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    monkeypatch.setenv("RDR_SUN_CANDIDATES", "0")
    plain = Renderer(w, h, conf)
    monkeypatch.delenv("RDR_SUN_CANDIDATES")
    plain.set_scene(sc)
    shadowed = []
    for k, sun in enumerate(suns):
        if k == 3:  # move every other instance up and sideways: the lists of the third frame no longer hold
            t = np.array(sc.global_transforms, dtype=np.float32).reshape(-1, 16).copy()
            t[::2, 12] += 0.8
            t[::2, 13] += 1.1
            sc.global_transforms = t
            o.set_scene(sc)
            g.update_transforms(t)
            plain.update_transforms(t)
        ovis = o.render_visbuffer(sun=sun)
        g.render(sun=sun)
        plain.render(sun=sun)
        assert np.array_equal(g.read_visbuffer_u64(), ovis)
        check_color(o, g)
        assert np.array_equal(g.read_color(), plain.read_color())
        shadowed.append(g.read_color().copy())
    assert np.array_equal(shadowed[0], shadowed[2]) and not np.array_equal(shadowed[0], shadowed[1])
    g.close()
    plain.close()


def test_sun_candidates_under_sheared_instances(oracle, monkeypatch):
    """Instances under shearing, non-uniformly scaling transforms shadow each other: the world-space radii behind the
    occluder lists have to bound what such a matrix does to the mesh (its longest column does not)."""
    rng = np.random.default_rng(5)
    sc = synth.make_field(1, 3, lods=1, with_env=True, prepared=True)  # one sphere mesh, material and environment
    mesh = 0
    for k in range(60):
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = np.diag(rng.uniform(0.2, 1.0, 3)).astype(np.float32)
        shear = np.eye(3, dtype=np.float32)
        shear[0, 1], shear[0, 2], shear[1, 2] = rng.uniform(-1.5, 1.5, 3)
        m[:3, :3] = shear @ m[:3, :3]
        m[:3, 3] = (rng.uniform(-6, 6), rng.uniform(0.0, 4.0), rng.uniform(-6, 6))
        sc.add_draw(mesh, m.T.reshape(16), 0)  # column-major
    w, h = 480, 270
    conf = RendererConfig(lod_enabled=False, cull_enabled=False)
    cam = Camera((0.0, 4.0, 14.0), (0.0, 0.2, 1.0))
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    monkeypatch.setenv("RDR_SUN_CANDIDATES", "0")
    plain = Renderer(w, h, conf)
    monkeypatch.delenv("RDR_SUN_CANDIDATES")
    plain.set_scene(sc)
    for sun in [(0.6, 0.5, 0.3), (-0.2, 0.9, -0.3)]:
        ovis = o.render_visbuffer(cam, sun)
        g.render(cam, sun)
        plain.render(cam, sun)
        assert covered(ovis).sum() > 5000
        assert np.array_equal(g.read_visbuffer_u64(), ovis)
        check_color(o, g)
        assert np.array_equal(g.read_color(), plain.read_color())
    g.close()
    plain.close()


def test_sun_candidates_at_the_edge_of_an_occluders_reach(monkeypatch):
    """An instance stays on its neighbour's list of possible sun occluders as long as its ball reaches the bundle of sun
    rays that start on the neighbour.  A sphere slides past the sun line of another, from well inside that bundle to
    well outside it in small steps, under a sheared transform as well: the image with the lists equals the image
    without them at every step, and the shadow is there at one end of the sweep and gone at the other."""
    sc = synth.make_field(1, 3, lods=1, with_env=True, prepared=True)
    sun = np.array((0.35, 0.9, 0.1))
    sun /= np.linalg.norm(sun)
    side = np.cross(sun, (0.0, 0.0, 1.0))
    side /= np.linalg.norm(side)
    base = np.array(sc.global_transforms, dtype=np.float32).reshape(-1, 16)[0].reshape(4, 4).T.copy()  # the receiver
    centre = base[:3, 3].astype(np.float64)
    scale = float(np.linalg.norm(base[:3, 0]))
    sc.add_draw(0, base.T.reshape(16), 0)  # the occluder's draw (moved below)
    w, h = 320, 180
    conf = RendererConfig(lod_enabled=False, cull_enabled=False)
    cam = Camera((float(centre[0]), float(centre[1]) + 2.0, float(centre[2]) - 5.0), (0.0, 0.3, -1.0))
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    monkeypatch.setenv("RDR_SUN_CANDIDATES", "0")
    plain = Renderer(w, h, conf)
    monkeypatch.delenv("RDR_SUN_CANDIDATES")
    plain.set_scene(sc)
    dark = []
    for shear in (0.0, 0.8):
        for off in np.linspace(0.0, 2.6 * scale, 14):
            m = base.copy()
            m[:3, :3] *= np.float32(0.6)
            m[0, 1] += np.float32(shear * 0.6 * scale)
            m[:3, 3] = (centre + sun * 3.0 * scale + side * off).astype(np.float32)
            t = np.array(sc.global_transforms, dtype=np.float32).reshape(-1, 16).copy()
            t[1] = m.T.reshape(16)
            g.update_transforms(t)
            plain.update_transforms(t)
            g.render(cam, tuple(sun))
            plain.render(cam, tuple(sun))
            assert np.array_equal(g.read_visbuffer_u64(), plain.read_visbuffer_u64())
            a, b = g.read_color(), plain.read_color()
            assert np.array_equal(a, b), (shear, off)
            receiver = g.read_visbuffer().reshape(h, w, 2)[..., 1] == 0
            dark.append((a[..., :3].astype(int).sum(axis=2), receiver))
    for first, last in ((0, 13), (14, 27)):  # the shadow on the receiver is there at one end of a sweep and gone at the other
        both = dark[first][1] & dark[last][1]
        assert ((dark[last][0] - dark[first][0])[both] > 60).sum() > 50
    g.close()
    plain.close()


def _cube_mesh():
    """A closed unit cube (half-extent 1) as six outward-facing quads in one mesh."""
    faces = [[(1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1)], [(-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1)],
             [(-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1)], [(-1, -1, 1), (-1, -1, -1), (1, -1, -1), (1, -1, 1)],
             [(-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)], [(1, -1, -1), (-1, -1, -1), (-1, 1, -1), (1, 1, -1)]]
    vs, idx = [], []
    for f in faces:
        v, i = synth.quad_mesh(f)
        n = np.cross(np.subtract(f[1], f[0]), np.subtract(f[3], f[0]))
        assert np.dot(n, np.mean(f, axis=0)) > 0  # outward
        idx.append(i + 4 * len(vs))
        vs.append(v)
    return np.concatenate(vs), np.concatenate(idx).astype(np.uint32)


@pytest.mark.parametrize("elevation", [1 / 150.0, 1 / 400.0])
def test_far_shadows_of_small_closed_meshes(oracle, elevation):
    """A low sun: the shadows of a small sphere and a small cube (closed convex meshes, scaled instances) fall on the
    ground hundreds of radii away, so the shadow rays meet the instances' ball tests with the origin very far from a
    small ball, where a discriminant of the form b^2 - a c cancels to nothing.  The checker traces every ray through
    every triangle; a ray wrongly judged to miss (or to be swallowed by) a ball would be a lit (or dark) pixel on the
    shadow's edge, 65 % of its brightness."""
    from renderer_amd.scene import Scene
    sc = Scene()
    sc.env = synth.default_env()
    m_ground = sc.add_material(base_color_factor=(0.9, 0.9, 0.9, 1.0), metallic_roughness_normal_occlusion=(0.0, 1.0, 1, 1))
    gv, gi = synth.grid_mesh([(-520.0, 0.0, -6.0), (-520.0, 0.0, 6.0), (20.0, 0.0, 6.0), (20.0, 0.0, -6.0)], 64, 4)
    sc.add_draw(sc.append_mesh(gv, [gi]), mat4_trs(), m_ground)
    v, lods, _ = synth.icosphere(3, lods=1)
    ball = sc.append_mesh(v, [lods[0]])
    cv, ci = _cube_mesh()
    cube = sc.append_mesh(cv, [ci])
    q = synth._shoemake(0.3, 0.7, 0.2)
    sc.add_draw(ball, mat4_trs((0.0, 1.0, -1.5), q, (0.25, 0.25, 0.25)), m_ground)
    sc.add_draw(cube, mat4_trs((2.0, 1.2, 1.5), q, (0.3, 0.3, 0.3)), m_ground)
    # the sun just above the horizon in +x: an object at height y shades the ground y / elevation further along -x
    sun = (1.0, elevation, 0.0)
    reach = 1.0 / elevation
    for back in (0.85, 1.15):  # two places along the shadows (they span 0.75 .. 1.25 and more of `reach`)
        sc.camera = Camera((-reach * back, 5.0, 0.0), (0.05, 1.0, 0.02))  # looks straight down (along -direction)
        o, ovis, g = both(oracle, sc, 480, 270, cull_enabled=False)
        g.render(sc.camera, sun)
        o.render_visbuffer(sc.camera, sun)
        ref = o.shade()
        img = g.read_color()
        # the frame must hold both lit and shadowed ground, or the test tests nothing
        lum = ref[..., :3].astype(int).sum(axis=2)
        assert (lum < 0.75 * lum.max()).sum() > 200 and (lum > 0.9 * lum.max()).sum() > 200  # (0.35 linear is 0.63 after the sRGB curve)
        err = np.abs(img.astype(np.int32) - ref.astype(np.int32))
        assert err.max() <= 2, f"elevation {elevation}, at {back}: {int((err.max(axis=2) > 2).sum())} pixels differ by up to {err.max()} / 255"
        g.close()


def test_known_answer_triangle():
    W = H = 64
    sc = fb_triangle_scene([[(8, 8, 5.0), (8, 40, 5.0), (40, 8, 5.0)]], W, H)
    g = Renderer(W, H)
    g.set_scene(sc)
    g.render()
    vis = g.read_visbuffer_u64()
    py, px = np.mgrid[0:H, 0:W]
    assert np.array_equal(covered(vis), (px >= 8) & (py >= 8) & (px + py + 1 < 48))
    xy = g.read_visbuffer()
    assert (xy[~covered(vis)] == 0xFFFFFFFF).all() and (xy[covered(vis)] == 0).all()
    col = g.read_color()
    assert (col[~covered(vis)] == 0).all() and (col[covered(vis)][:, 3] == 255).all()


def test_partitions_composite_to_full_frame(oracle):
    sc = synth.make_atrium()
    w, h = 320, 180
    g = Renderer(w, h)
    g.set_scene(sc)
    g.render_visbuffer()
    full = g.read_visbuffer_u64()
    n = sc.meshlet_visibility_count
    parts = []
    for k in range(4):
        g.set_partition(n * k // 4, n * (k + 1) // 4)
        g.render_visbuffer()
        parts.append(g.read_visbuffer_u64())
    assert np.array_equal(np.minimum.reduce(parts), full)
    o = oracle.OracleRenderer(w, h)
    o.set_scene(sc)
    o.set_partition(n // 4, n // 2)
    assert np.array_equal(o.render_visbuffer(), parts[1])


@pytest.mark.parametrize("world,chunk", [(2, 64), (3, 64), (8, 64), (5, 128)])
def test_interleaved_partitions_composite_to_full_frame(oracle, world, chunk):
    """Chunks of every draw's meshlets dealt round-robin (rdr_set_partition_interleaved): each rank's buffer is
    the oracle's for the same share, the shares' work counters add up to the frame's, and the minimum over the
    ranks is the full frame; with level-of-detail selection on, chunks are counted within the selected level."""
    sc = synth.make_field(60, 4, lods=3, alpha_fraction=0.3)
    w, h = 320, 180
    conf = RendererConfig(lod_enabled=True, cull_enabled=False)
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    g.render_visbuffer()
    full = g.read_visbuffer_u64().copy()
    whole = g.stats()
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    acc = np.full_like(full, abi.VIS_CLEAR)
    sums = dict.fromkeys(STAT_KEYS[1:], 0)
    for rank in range(world):
        g.set_partition_interleaved(chunk, rank, world)
        g.render_visbuffer()
        part = g.read_visbuffer_u64()
        if rank in (0, world - 1):
            o.set_partition_interleaved(chunk, rank, world)
            assert np.array_equal(o.render_visbuffer(), part)
        st = g.stats()
        for k in sums:
            sums[k] += st[k]
        acc = np.minimum(acc, part)
    assert np.array_equal(acc, full)
    assert sums == {k: whole[k] for k in sums}
    # the range still applies on top of the interleaving
    n = sc.meshlet_visibility_count
    g.set_partition(n // 3, n)
    g.set_partition_interleaved(chunk, 0, world)
    g.render_visbuffer()
    o.set_partition(n // 3, n)
    o.set_partition_interleaved(chunk, 0, world)
    assert np.array_equal(o.render_visbuffer(), g.read_visbuffer_u64())
    with pytest.raises(RendererError, match="multiple of 64"):
        g.set_partition_interleaved(48, 0, 2)
    with pytest.raises(RendererError, match="below world"):
        g.set_partition_interleaved(64, 2, 2)
    g.close()


@pytest.mark.parametrize("world,size,occlusion", [(2, (320, 180), False), (3, (333, 187), False), (8, (256, 300), False), (3, (320, 180), True)])
def test_draws_every_rank_takes_compose_to_the_full_frame(world, size, occlusion):
    """rdr_set_partition_replication: the hall's few screen-filling triangles are drawn by every rank, each on the
    rows it shades.  What the exchange reads of a rank's buffer are its own rows, so: the minimum over the ranks is
    the full frame, every rank's own rows hold nothing the full frame does not, and the pixel work of the large
    triangles is spread over the ranks instead of sitting with one."""
    sc = synth.make_atrium()
    w, h = size
    conf = RendererConfig(cull_enabled=occlusion)
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    frames = 3 if occlusion else 1
    for _ in range(frames):
        g.render_visbuffer()
    full = g.read_visbuffer_u64().copy()
    frag = {}
    for px in (0.0, 64.0):
        acc = np.full_like(full, abi.VIS_CLEAR)
        frag[px] = []
        for rank in range(world):
            g.set_partition_interleaved(64, rank, world)
            g.set_shade_bands(32, rank, world)
            g.set_partition_replication(px)
            for _ in range(frames):
                g.render_visbuffer()
            assert g.frame_status() == 0
            part = g.read_visbuffer_u64()
            frag[px].append(g.stats()["fragments"])
            mine = ((np.arange(h) // 32) % world) == rank
            assert (part[mine] >= full[mine]).all()
            if px > 0.0:  # the rank's own rows are complete for the draws every rank takes: the hall is one of them
                assert (part[mine] != abi.VIS_CLEAR).all()
            acc = np.minimum(acc, part)
        assert np.array_equal(acc, full), px
    if not occlusion:
        assert max(frag[64.0]) / (sum(frag[64.0]) / world) < max(frag[0.0]) / (sum(frag[0.0]) / world)
    # shading each rank's rows of its own composited buffer gives the single-GPU image (the shading passes find every
    # draw's per-frame record whichever rank drew it)
    g.set_partition_replication(0.0)
    g.set_partition_interleaved(64, 0, 1)
    g.set_shade_rows(0, h)
    g.close()


@pytest.mark.parametrize("world,band,size", [(2, 32, (320, 180)), (3, 32, (333, 187)), (8, 32, (256, 300)), (2, 64, (320, 180))])
def test_shade_bands_compose_to_the_full_image(world, band, size):
    """Bands of rows dealt round-robin (rdr_set_shade_bands): a rank shades exactly its rows, and the ranks'
    rows together are the full image (skybox pixels included)."""
    sc = synth.make_field(40, 3, lods=1, with_env=True)
    w, h = size
    g = Renderer(w, h, RendererConfig(cull_enabled=False, skybox=True))
    g.set_scene(sc)
    g.render()
    full = g.read_color().copy()
    covered_all = g.stats()["pixels_covered"]
    acc = np.zeros_like(full)
    covered_sum = 0
    for rank in range(world):
        g.set_shade_bands(band, rank, world)
        g.render()
        img = g.read_color()
        mine = ((np.arange(h) // band) % world) == rank
        acc[mine] = img[mine]
        covered_sum += g.stats()["pixels_covered"]
    assert np.array_equal(acc, full)
    assert covered_sum == covered_all
    g.set_shade_rows(0, h)  # back to a plain range
    g.render()
    assert np.array_equal(g.read_color(), full)
    with pytest.raises(RendererError, match="power of two"):
        g.set_shade_bands(48, 0, 2)
    g.close()


def test_update_transforms(oracle):
    sc = synth.make_field(30, 2)
    sc.camera = Camera((0.0, 4.0, -14.0), (0.0, 0.25, -1.0))
    w, h = 256, 144
    g = Renderer(w, h)
    g.set_scene(sc)
    g.render()
    before = g.read_visbuffer_u64()
    moved = sc.global_transforms.copy()
    moved[:, 13] += 1.5  # lift every instance
    g.update_transforms(moved)
    g.render()
    sc.global_transforms = moved
    o = oracle.OracleRenderer(w, h)
    o.set_scene(sc)
    ovis = o.render_visbuffer()
    assert not np.array_equal(before, ovis)
    check_vis(o, ovis, g)
    check_color(o, g)


def test_render_is_idempotent():
    sc = synth.make_atrium()
    g = Renderer(256, 144)
    g.set_scene(sc)
    g.render()
    a, ca = g.read_visbuffer_u64(), g.read_color()
    for _ in range(3):
        g.render()
    assert np.array_equal(g.read_visbuffer_u64(), a) and np.array_equal(g.read_color(), ca)


def test_errors():
    g = Renderer(64, 36)
    with pytest.raises(RendererError, match="no scene"):
        g.render(Camera((0, 0, 0), (0, 0, -1)), (0, 1, 0))
    sc = synth.make_field(2, 1, with_env=False)
    bad = synth.make_field(2, 1, with_env=False)
    bad.mesh_draws["mesh_index"] = 7
    with pytest.raises(RendererError, match="mesh_index"):
        g.set_scene(bad)
    bad = synth.make_field(2, 1, with_env=False)
    bad.meshlet_vertices[3] = 10 ** 6
    with pytest.raises(RendererError, match="vertex id"):
        g.set_scene(bad)
    g.set_scene(sc)
    with pytest.raises(RendererError, match="whole transform array"):
        g.update_transforms(np.zeros((1, 16), np.float32))
    g.render()
    assert g.read_color().shape == (36, 64, 4)


@pytest.mark.parametrize("occlusion", [False, True])
def test_bench_scene_at_low_resolution(oracle, occlusion):
    """The scene the benchmark times (1250 instances, 164 000 visible meshlets: far more than the rasteriser's
    resident waves take by their place in the grid, so nearly all of them are handed out by ticket, in both
    phases of a frame with occlusion culling) at a resolution the CPU check renders in a blink: same keys,
    same work counters, same colours."""
    sc = synth.make_field(1250, 5, lods=1, with_env=True, prepared=True)
    o, ovis, g = both(oracle, sc, 480, 270, lod_enabled=False, cull_enabled=occlusion)
    check_vis(o, ovis, g)
    check_color(o, g)
    assert g.frame_status() == 0
    g.close()


def test_full_size_properties():
    """At the benchmark's size the oracle is too slow to run in a test; check size-independent
    properties instead, on the scene the benchmark times (meshlets from the library's mesh preparation):
    sort-last partitions min-compose to the full frame (contiguous thirds and interleaved eighths), every key
    decodes to a valid (draw, meshlet, triangle), shading covers exactly the keyed pixels, a second frame repeats
    the first."""
    sc = synth.make_field(1250, 5, lods=1, with_env=True, prepared=True)
    w, h = 3840, 2160
    g = Renderer(w, h, RendererConfig(lod_enabled=False, cull_enabled=False))
    g.set_scene(sc)
    g.render()
    assert g.frame_status() == 0
    full = g.read_visbuffer_u64().copy()
    st = g.stats()
    assert st["triangles_submitted"] == st["draws_visible"] * 20480 <= sc.triangle_count() == 1250 * 20480
    cov = covered(full)
    assert st["pixels_covered"] == int(cov.sum()) > 0
    _valid_keys(sc, g, cov)
    col = g.read_color().copy()
    assert (col[..., 3][cov] == 255).all() and (col[~cov] == 0).all()
    g.render()
    assert np.array_equal(g.read_visbuffer_u64(), full) and np.array_equal(g.read_color(), col)
    n = sc.meshlet_visibility_count
    acc = np.full_like(full, abi.VIS_CLEAR)
    for k in range(3):
        g.set_partition(n * k // 3, n * (k + 1) // 3)
        g.render_visbuffer()
        acc = np.minimum(acc, g.read_visbuffer_u64())
    assert np.array_equal(acc, full)
    g.set_partition(0, n)
    acc = np.full_like(full, abi.VIS_CLEAR)
    for k in range(8):
        g.set_partition_interleaved(64, k, 8)
        g.render_visbuffer()
        acc = np.minimum(acc, g.read_visbuffer_u64())
    assert np.array_equal(acc, full)
    g.close()


SECONDARY_FIELDS = {
    # the benchmark's secondary field series, by the same constructors and configurations
    "alpha": (dict(lods=1, alpha_fraction=0.5), dict(lod_enabled=False, cull_enabled=False)),
    "lod": (dict(lods=8), dict(lod_enabled=True, cull_enabled=False)),
    "occlusion": (dict(lods=1), dict(lod_enabled=False, cull_enabled=True)),
}


@pytest.mark.parametrize("series", sorted(SECONDARY_FIELDS))
def test_secondary_field_series_4k_properties(series):
    """The other field workloads the benchmark times, at its size: complete frame, valid keys, shading covers
    exactly the keyed pixels, a second frame repeats the first, interleaved partitions min-compose to the full
    frame, and two-phase occlusion culling changes neither the keys nor the image."""
    scene_kw, conf_kw = SECONDARY_FIELDS[series]
    sc = synth.make_field(1250, 5, with_env=True, prepared=True, **scene_kw)
    w, h = 3840, 2160
    g = Renderer(w, h, RendererConfig(**conf_kw))
    g.set_scene(sc)
    for _ in range(2 if conf_kw["cull_enabled"] else 1):  # second frame: the early phase has something to draw
        g.render()
    assert g.frame_status() == 0
    full = g.read_visbuffer_u64().copy()
    col = g.read_color().copy()
    st = g.stats()
    cov = covered(full)
    assert st["pixels_covered"] == int(cov.sum()) > w * h // 8
    _valid_keys(sc, g, cov)
    assert (col[..., 3][cov] == 255).all() and (col[~cov] == 0).all()
    if series == "lod":
        assert 0 < st["triangles_submitted"] < sc.triangle_count()
    g.render()
    assert g.frame_status() == 0
    assert np.array_equal(g.read_visbuffer_u64(), full) and np.array_equal(g.read_color(), col)
    acc = np.full_like(full, abi.VIS_CLEAR)
    for k in range(4):
        g.set_partition_interleaved(64, k, 4)
        for _ in range(2 if conf_kw["cull_enabled"] else 1):
            g.render_visbuffer()
        assert g.frame_status() == 0
        acc = np.minimum(acc, g.read_visbuffer_u64())
    assert np.array_equal(acc, full)
    g.close()
    # the other setting of the occlusion switch: bit for bit the same keys and image
    other = dict(conf_kw, cull_enabled=not conf_kw["cull_enabled"])
    c = Renderer(w, h, RendererConfig(**other))
    c.set_scene(sc)
    for _ in range(3):
        c.render()
    assert c.frame_status() == 0
    assert np.array_equal(c.read_visbuffer_u64(), full) and np.array_equal(c.read_color(), col)
    c.close()


def test_bench_camera_path_4k_properties():
    """The benchmark's moving camera at its size: along the path (every eighth of its 120 poses, so that the view changes by
    whole spheres from frame to frame) two-phase occlusion culling gives the keys and the image of the single pass bit for
    bit, every frame is complete and its keys are valid."""
    import bench
    sc = synth.make_field(1250, 5, lods=1, with_env=True, prepared=True)
    w, h = 3840, 2160
    one = Renderer(w, h, RendererConfig(lod_enabled=False, cull_enabled=False))
    two = Renderer(w, h, RendererConfig(lod_enabled=False, cull_enabled=True))
    one.set_scene(sc)
    two.set_scene(sc)
    seen = None
    for pos, direction in bench.camera_path(120)[::8]:
        cam = Camera(pos, direction)
        one.render(cam)
        two.render(cam)
        assert one.frame_status() == 0 and two.frame_status() == 0
        keys = one.read_visbuffer_u64()
        assert np.array_equal(two.read_visbuffer_u64(), keys)
        assert np.array_equal(two.read_color(), one.read_color())
        cov = covered(keys)
        assert cov.sum() > w * h // 8
        _valid_keys(sc, one, cov)
        assert seen is None or not np.array_equal(keys, seen)  # the camera did move
        seen = keys.copy()
    one.close()
    two.close()


def _valid_keys(sc, g, cov):
    xy = g.read_visbuffer()
    draw, meshlet, tri = xy[..., 1][cov], xy[..., 0][cov] >> 8, xy[..., 0][cov] & 0xFF
    assert draw.max() < len(sc.mesh_draws) and meshlet.max() < len(sc.meshlets)
    assert (tri < sc.meshlets["triangle_count"][meshlet]).all()


def test_field_xl_8k_properties():
    """The largest configuration of the measurement plan: 50 000 instances (1.02 G triangles submitted) at
    7680 x 4320 on one GPU.  No list overflows, every key decodes to a valid triangle, shading covers exactly
    the keyed pixels, two sort-last partitions min-compose to the full frame, a second frame repeats the first."""
    sc = synth.make_field(50000, 5, lods=1, with_env=True, prepared=True)
    w, h = 7680, 4320
    g = Renderer(w, h, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    g.render()
    assert g.frame_status() == 0
    full = g.read_visbuffer_u64().copy()
    st = g.stats()
    tris = int(sc.meshlets["triangle_count"][:int(sc.meshes[0]["lods"][0]["meshlet_count"])].sum())
    assert tris == 20480 and sc.triangle_count() == 50000 * tris
    assert st["triangles_submitted"] == st["draws_visible"] * tris and 0 < st["draws_visible"] <= 50000  # the far test drops the rest
    cov = covered(full)
    assert st["pixels_covered"] == int(cov.sum()) > w * h // 4
    _valid_keys(sc, g, cov)
    col = g.read_color()
    assert (col[..., 3][cov] == 255).all() and (col[~cov] == 0).all()
    del col
    g.render()
    assert np.array_equal(g.read_visbuffer_u64(), full)
    n = sc.meshlet_visibility_count
    acc = np.full_like(full, abi.VIS_CLEAR)
    for k in range(2):
        g.set_partition(n * k // 2, n * (k + 1) // 2)
        g.render_visbuffer()
        assert g.frame_status() == 0
        acc = np.minimum(acc, g.read_visbuffer_u64())
    assert np.array_equal(acc, full)
    g.close()


def test_atrium_4k_properties():
    """The interior scene (textured, every pixel covered, tile path) at the 4K target of the single-GPU
    configurations: complete frame, valid keys, deterministic, partitions min-compose, occlusion culling neutral."""
    sc = synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5)
    w, h = 3840, 2160
    g = Renderer(w, h, RendererConfig(cull_enabled=False, skybox=True))
    g.set_scene(sc)
    g.render()
    assert g.frame_status() == 0
    full = g.read_visbuffer_u64().copy()
    col = g.read_color().copy()
    st = g.stats()
    assert st["bin_entries"] > 0 and st["pixels_covered"] == w * h == int(covered(full).sum())
    _valid_keys(sc, g, covered(full))
    assert (col[..., 3] == 255).all()
    g.render()
    assert np.array_equal(g.read_visbuffer_u64(), full) and np.array_equal(g.read_color(), col)
    n = sc.meshlet_visibility_count
    acc = np.full_like(full, abi.VIS_CLEAR)
    for k in range(4):
        g.set_partition(n * k // 4, n * (k + 1) // 4)
        g.render_visbuffer()
        acc = np.minimum(acc, g.read_visbuffer_u64())
    assert np.array_equal(acc, full)
    g.close()
    c = Renderer(w, h, RendererConfig(cull_enabled=True, skybox=True))
    c.set_scene(sc)
    for _ in range(3):
        c.render()
    assert c.frame_status() == 0
    assert np.array_equal(c.read_visbuffer_u64(), full) and np.array_equal(c.read_color(), col)
    c.close()


@pytest.mark.parametrize("lod", [False, True])
def test_prepared_meshes_with_simplified_lods(oracle, lod):
    """Meshes that went through the library's mesh preparation (cache-ordered meshlets, simplified
    LOD chain): instances near and far pick different levels when LOD selection is on."""
    from renderer_amd.scene import Scene
    v, lods, _ = synth.icosphere(4, lods=1)
    gv, gi = synth.grid_mesh(np.array([[-30, -1, -5], [30, -1, -5], [30, -1, 120], [-30, -1, 120]], np.float32), 24, 48)
    sc = Scene()
    ball, ground = sc.append_mesh_auto(v, lods[0]), sc.append_mesh_auto(gv, gi)
    assert int(sc.meshes[ball]["lod_count"]) >= 4
    sc.add_draw(ground, mat4_trs(), 0xFFFF)
    rng = np.random.default_rng(5)
    for k in range(40):
        z = 2.0 + 2.8 * k
        sc.add_draw(ball, mat4_trs((rng.uniform(-8, 8), rng.uniform(0, 3), z), scale=(rng.uniform(0.5, 1.5),) * 3), 0xFFFF)
    sc.env = synth.default_env()
    sc.camera = Camera((0.0, 3.0, -6.0), (0.0, -0.1, 1.0))
    o, ovis, g = both(oracle, sc, 384, 216, lod_enabled=lod, cull_enabled=False)
    check_vis(o, ovis, g)
    check_color(o, g)
    if lod:  # fewer triangles submitted than with every instance at level 0
        full = sc.triangle_count()
        assert 0 < g.stats()["triangles_submitted"] < full
    g.close()


def test_8k_properties_with_large_clipped_triangles():
    """7680 x 4320 (the largest target the measurement plan names) on the scene that exercises the
    tile path: rendering twice gives the same buffers, sort-last partitions min-compose to the full
    frame, every key decodes to a valid triangle and every pixel is covered (a closed hall)."""
    sc = synth.make_atrium()
    w, h = 7680, 4320
    g = Renderer(w, h, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    g.render()
    full = g.read_visbuffer_u64().copy()
    col = g.read_color().copy()
    st = g.stats()
    assert st["bin_entries"] > 0 and st["pixels_covered"] == w * h == int(covered(full).sum())
    g.render()
    assert np.array_equal(g.read_visbuffer_u64(), full) and np.array_equal(g.read_color(), col)
    xy = g.read_visbuffer()
    draw, meshlet, tri = xy[..., 1], xy[..., 0] >> 8, xy[..., 0] & 0xFF
    assert draw.max() < len(sc.mesh_draws) and meshlet.max() < len(sc.meshlets)
    assert (tri < sc.meshlets["triangle_count"][meshlet]).all()
    n = sc.meshlet_visibility_count
    acc = np.full_like(full, abi.VIS_CLEAR)
    for k in range(2):
        g.set_partition(n * k // 2, n * (k + 1) // 2)
        g.render_visbuffer()
        acc = np.minimum(acc, g.read_visbuffer_u64())
    assert np.array_equal(acc, full)
    g.close()


def test_shading_the_same_visibility_buffer_twice_gives_the_same_image(oracle):
    """The first shading pass after a geometry pass relies on the list heads the clear zeroed; a
    second pass over the same visibility buffer has to reset them itself."""
    sc = synth.make_field(40, 2, lods=1, alpha_fraction=0.3, with_env=True)
    g = Renderer(320, 200, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    g.render()
    first = g.read_color().copy()
    covered_first = g.stats()["pixels_covered"]
    g.shade()
    assert np.array_equal(g.read_color(), first)
    assert g.stats()["pixels_covered"] == covered_first
    g.render_visbuffer()
    g.shade()
    g.shade()
    assert np.array_equal(g.read_color(), first)
    g.close()
