"""Skybox pass (renderer/src/renderer/pass.rs:397-445, cubemap.vert/frag.glsl): pixels without
geometry show the environment cubemap along their view ray."""
import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, synth
from renderer_amd.scene import Camera, Cubemap
from helpers import covered, psnr

# linear face colours whose sRGB encodes are far apart: +X,-X,+Y,-Y,+Z,-Z
FACE_RGB = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1]], np.float32)


def face_coloured_env(n=64):
    faces = np.zeros((6, n, n, 4), np.float32)
    faces[..., :3] = FACE_RGB[:, None, None, :]
    faces[..., 3] = 0.5
    env = synth.default_env()
    env.skybox = Cubemap([faces])
    return env


def major_face(d):
    a = int(np.argmax(np.abs(d)))
    return 2 * a + (0 if d[a] >= 0 else 1)


def expected_face_image(ubo, w, h):
    """Independent derivation: un-project each pixel centre with the inverse of view_proj (float64)
    and take the dominant axis of the ray from the camera."""
    vp = np.array(ubo.view_proj, np.float64).reshape(4, 4).T  # column-major storage
    inv = np.linalg.inv(vp)
    cam = np.array(ubo.camera_position, np.float64)
    out = np.zeros((h, w), np.int32)
    margin = np.zeros((h, w))
    for y in range(h):
        for x in range(w):
            ndc = np.array([2 * (x + 0.5) / w - 1, 1 - 2 * (y + 0.5) / h, 0.5, 1.0])
            p = inv @ ndc
            d = p[:3] / p[3] - cam
            out[y, x] = major_face(d)
            s = np.sort(np.abs(d))
            margin[y, x] = (s[2] - s[1]) / s[2]
    return out, margin


@pytest.mark.parametrize("direction", [(0.0, 0.0, 1.0), (0.0, 0.0, -1.0), (1.0, 0.1, 0.0), (-0.9, 0.2, 0.1),
                                       (0.1, 0.99, 0.05), (0.2, -0.95, 0.1), (0.6, 0.5, -0.6)])
def test_oracle_skybox_shows_the_face_each_ray_points_at(oracle, direction):
    w, h = 48, 27
    sc = synth.make_triangle_scene([(2.0, 1.0, -49.0), (4.0, 1.0, -49.0), (3.0, 3.0, -49.0)])
    sc.env = face_coloured_env()
    sc.camera = Camera((3.0, 2.0, -50.0), direction)
    o = oracle.OracleRenderer(w, h, RendererConfig(skybox=True))
    o.set_scene(sc)
    vis = o.render_visbuffer()
    img = o.shade()
    want, margin = expected_face_image(o.ubo, w, h)
    sel = ~covered(vis) & (margin > 0.05)  # away from cube edges, where bilinear filtering blends two faces
    assert sel.sum() > w * h // 2
    assert np.array_equal(img[sel][:, :3], (FACE_RGB[want[sel]] * 255).astype(np.uint8))
    assert (img[sel][:, 3] == 128).all()  # alpha is stored as is: floor(0.5 * 255 + 0.5)
    # without the flag the same pixels stay clear
    o2 = oracle.OracleRenderer(w, h, RendererConfig())
    o2.set_scene(sc)
    o2.render_visbuffer()
    assert (o2.shade()[~covered(vis)] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["atrium", "field"])
def test_hip_skybox_matches_oracle(oracle, scene_name):
    w, h = 320, 180
    sc = synth.make_atrium() if scene_name == "atrium" else synth.make_field(40, 2, with_env=True)
    if scene_name == "atrium":
        sc.camera = Camera((40.0, 25.0, -60.0), (0.4, 0.3, -1.0))  # outside the building: sky around it
    assert sc.env is not None and sc.env.skybox is not None
    conf = RendererConfig(skybox=True)
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    vis = o.render_visbuffer()
    ref = o.shade()
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    g.render()
    assert np.array_equal(g.read_visbuffer_u64(), vis)
    img = g.read_color()
    empty = ~covered(vis)
    assert empty.sum() > 0 and ref[empty][:, 3].min() == 255  # the synthetic sky is opaque
    assert np.abs(img.astype(np.int32) - ref.astype(np.int32)).max() <= 2
    assert psnr(img, ref) >= 40.0
    # rows shaded separately (sort-last strips) give the same image
    g.set_shade_rows(0, h // 2)
    g.render()
    top = g.read_color()[: h // 2].copy()
    g.set_shade_rows(h // 2, h)
    g.shade()
    both = np.concatenate([top, g.read_color()[h // 2:]])
    assert np.array_equal(both, img)


def _cube_face(d):
    x, y, z = d
    ax, ay, az = abs(x), abs(y), abs(z)
    if ax >= ay and ax >= az:
        return (0, -z, -y, ax) if x >= 0 else (1, z, -y, ax)
    if ay >= az:
        return (2, x, z, ay) if y >= 0 else (3, x, -z, ay)
    return (4, x, -y, az) if z >= 0 else (5, -x, -y, az)


def _cube_dir(face, a, b):
    return [(1.0, -b, -a), (-1.0, -b, a), (a, 1.0, b), (a, -1.0, -b), (a, -b, 1.0), (-a, -b, -1.0)][face]


def _sample_by_reprojection(faces, d):
    """Bilinear cube sample where a tap outside the face is fetched from the texel that contains its
    re-projected centre (float64, written independently of the integer rule the library uses).
    Returns (rgba, touches_a_corner)."""
    n = faces.shape[1]
    face, sc, tc, ma = _cube_face(d)
    x, y = 0.5 * (sc / ma + 1) * n - 0.5, 0.5 * (tc / ma + 1) * n - 0.5
    i0, j0 = int(np.floor(x)), int(np.floor(y))
    ax, ay = x - i0, y - j0
    taps, corner = [], False
    for j in (j0, j0 + 1):
        for i in (i0, i0 + 1):
            f, ii, jj = face, i, j
            out_i, out_j = not 0 <= i < n, not 0 <= j < n
            corner |= out_i and out_j
            if out_i or out_j:
                a, b = (i + 0.5) / n * 2 - 1, (j + 0.5) / n * 2 - 1
                f, s2, t2, m2 = _cube_face(_cube_dir(face, a, b))
                ii = min(max(int(np.floor(0.5 * (s2 / m2 + 1) * n)), 0), n - 1)
                jj = min(max(int(np.floor(0.5 * (t2 / m2 + 1) * n)), 0), n - 1)
            taps.append(faces[f, jj, ii].astype(np.float64))
    top = taps[0] * (1 - ax) + taps[1] * ax
    bot = taps[2] * (1 - ax) + taps[3] * ax
    return top * (1 - ay) + bot * ay, corner


@pytest.mark.parametrize("direction", [(1.0, 1.0, 0.05), (-1.0, 0.1, 1.0), (0.05, -1.0, -1.0), (1.0, -0.1, -1.0),
                                       (-0.1, 1.0, 1.0), (-1.0, -1.0, 0.1)])
def test_oracle_seamless_edges_take_the_reprojected_texel(oracle, direction):
    """Looking at a cube edge with a small random cubemap: every sky pixel whose footprint stays clear
    of a cube corner must equal the bilinear blend over re-projected texels."""
    rng = np.random.default_rng(11)
    n = 8
    faces = rng.random((6, n, n, 4)).astype(np.float32)
    w, h = 40, 24
    sc = synth.make_triangle_scene([(2.0, 1.0, -49.0), (4.0, 1.0, -49.0), (3.0, 3.0, -49.0)])
    env = synth.default_env()
    env.skybox = Cubemap([faces])
    sc.env = env
    sc.camera = Camera((3.0, 2.0, 50.0), direction)
    o = oracle.OracleRenderer(w, h, RendererConfig(skybox=True))
    o.set_scene(sc)
    vis = o.render_visbuffer()
    img = o.shade().astype(np.int32)
    vp = np.array(o.ubo.view_proj, np.float64).reshape(4, 4).T
    inv, cam = np.linalg.inv(vp), np.array(o.ubo.camera_position, np.float64)
    checked = crossing = 0
    for y in range(h):
        for x in range(w):
            if covered(vis)[y, x]:
                continue
            p = inv @ np.array([2 * (x + 0.5) / w - 1, 1 - 2 * (y + 0.5) / h, 0.5, 1.0])
            d = p[:3] / p[3] - cam
            rgba, corner = _sample_by_reprojection(faces, d)
            if corner:
                continue
            c = np.clip(rgba[:3], 0, 1)
            srgb = np.where(c <= 0.0031308, 12.92 * c, 1.055 * c ** (1 / 2.4) - 0.055)
            want = np.floor(np.append(srgb, rgba[3]) * 255 + 0.5)
            assert np.abs(img[y, x] - want).max() <= 1, (x, y, img[y, x], want)
            checked += 1
            face, sc_, tc_, ma = _cube_face(d)
            fx, fy = 0.5 * (sc_ / ma + 1) * n - 0.5, 0.5 * (tc_ / ma + 1) * n - 0.5
            crossing += not (0 <= fx < n - 1 and 0 <= fy < n - 1)
    assert checked > w * h // 2 and crossing > 20  # a good share of the footprints really cross an edge
