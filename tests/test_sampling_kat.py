"""Known answers for the parts of the deferred shading pass that the flat, untextured cases of
test_shading_kat.py leave out: analytic barycentrics and their screen-space derivatives
(assets/shaders/shading.frag.glsl:81-120), ``textureGrad`` on a mip-mapped, non-constant, sRGB
texture with REPEAT wrap (:173-179), the normal map with a negative tangent handedness (:416-419),
the texture-transform and uv-set quirks (:373-414), and a prefiltered cubemap tap at a fractional
level (:207-225).

Everything here is float64 and derived from what the GLSL means and from the Vulkan specification's
texel-filtering rules, not from the C restatement under test:
  * barycentrics come from a ray / plane intersection through the pixel centre, and their
    derivatives are the differences to the pixels one to the right and one down (which is what the
    shader's analytic formula evaluates: lambda at ndc + (2 / W, 0) and at ndc - (0, 2 / H));
  * the level of detail is log2 of the longer of the two footprint axes in texels (Vulkan 1.3,
    "Scale Factor Operation" with explicit gradients), clamped to the chain; VK_FILTER_LINEAR on
    the two nearest levels, blended linearly; texel centres at integer + 0.5; REPEAT; sRGB texels
    are converted to linear before filtering;
  * cube face selection and (s, t) follow the specification's table; levels as for 2D images.
"""
import math

import numpy as np
import pytest

from renderer_amd import RendererConfig, abi
from renderer_amd.scene import Camera, Cubemap, EnvMap, Image, Scene, mat4_trs
from helpers import covered
from test_shading_kat import (LUT, LAMBERTIAN, GGX, CHARLIE, f_schlick, geometric_occlusion, microfacet_distribution, mix,
                              normalize)

W, H = 112, 64
SUN = (0.3, 0.8, -0.4)
CAM = Camera((0.0, 14.0, -70.0), (0.0, 0.25, -1.0))
# a large ground triangle, tilted in depth so that perspective matters; per-vertex data all different
POS = np.array([(40.0, -2.0, -30.0), (-40.0, -2.0, -30.0), (0.0, 6.0, 60.0)])
NRM = np.array([(0.10, 1.0, -0.05), (-0.12, 1.0, 0.02), (0.03, 1.0, 0.15)])
TAN = np.array([(1.0, 0.05, 0.0, -1.0), (1.0, -0.04, 0.1, -1.0), (1.0, 0.0, -0.08, -1.0)])  # handedness -1 at all three
UV0 = np.array([(0.25, 0.10), (6.40, 0.55), (3.10, 9.30)])   # several repeats across the triangle
UV1 = np.array([(91.0, -7.0), (13.0, 55.0), (-40.0, 2.5)])   # never used: uv1 is a copy of uv0 in the shader


# ---- float64 texture filtering, from the Vulkan specification ---------------------------------
def srgb_to_linear(c):
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def texel(level, x, y, srgb):
    t = level[y % level.shape[0], x % level.shape[1]].astype(np.float64)
    if level.dtype == np.uint8:
        t = t / 255.0
        if srgb:
            t[:3] = srgb_to_linear(t[:3])
    return t


def bilinear(level, u, v, srgb):
    h, w = level.shape[:2]
    x, y = u * w - 0.5, v * h - 0.5
    x0, y0 = math.floor(x), math.floor(y)
    a, b = x - x0, y - y0
    return ((1 - a) * (1 - b) * texel(level, x0, y0, srgb) + a * (1 - b) * texel(level, x0 + 1, y0, srgb) +
            (1 - a) * b * texel(level, x0, y0 + 1, srgb) + a * b * texel(level, x0 + 1, y0 + 1, srgb))


def texture_grad(img, uv, ddx, ddy):
    h, w = img.mips[0].shape[:2]
    rho = max(math.hypot(ddx[0] * w, ddx[1] * h), math.hypot(ddy[0] * w, ddy[1] * h))
    lod = min(max(math.log2(rho), 0.0), len(img.mips) - 1.0) if rho > 0 else 0.0
    l0 = int(math.floor(lod))
    l1 = min(l0 + 1, len(img.mips) - 1)
    t = lod - l0
    return (1 - t) * bilinear(img.mips[l0], uv[0], uv[1], img.srgb) + t * bilinear(img.mips[l1], uv[0], uv[1], img.srgb), lod


def cube_face_st(d):
    """Face index and (s, t) in [0, 1] of a direction: Vulkan 1.3, "Cube Map Face Selection"."""
    ax, ay, az = abs(d[0]), abs(d[1]), abs(d[2])
    if ax >= ay and ax >= az:
        f, sc, tc, ma = (0, -d[2], -d[1], ax) if d[0] >= 0 else (1, d[2], -d[1], ax)
    elif ay >= az:
        f, sc, tc, ma = (2, d[0], d[2], ay) if d[1] >= 0 else (3, d[0], -d[2], ay)
    else:
        f, sc, tc, ma = (4, d[0], -d[1], az) if d[2] >= 0 else (5, -d[0], -d[1], az)
    return f, 0.5 * (sc / ma + 1.0), 0.5 * (tc / ma + 1.0)


def cube_lod(cm, d, lod):
    """textureLod on a cubemap; None when a bilinear footprint would cross a face edge (the seamless
    edge rule is pinned by tests/test_skybox.py, not here)."""
    lod = min(max(lod, 0.0), len(cm.mips) - 1.0)
    l0 = int(math.floor(lod))
    l1 = min(l0 + 1, len(cm.mips) - 1)
    f, s, t = cube_face_st(d)
    out = []
    for l in (l0, l1):
        n = cm.mips[l].shape[1]
        x, y = s * n - 0.5, t * n - 0.5
        if x < 0 or y < 0 or x > n - 1 or y > n - 1:
            return None
        x0, y0 = min(int(math.floor(x)), n - 2) if n > 1 else 0, min(int(math.floor(y)), n - 2) if n > 1 else 0
        a, b = x - x0, y - y0
        face = cm.mips[l][f].astype(np.float64)
        x1, y1 = min(x0 + 1, n - 1), min(y0 + 1, n - 1)
        out.append((1 - a) * (1 - b) * face[y0, x0] + a * (1 - b) * face[y0, x1] + (1 - a) * b * face[y1, x0] + a * b * face[y1, x1])
    w = lod - l0
    return (1 - w) * out[0] + w * out[1]


# ---- the scene ---------------------------------------------------------------------------------
def random_image(seed, size, srgb, mips=None):
    rng = np.random.default_rng(seed)
    levels = []
    n = size
    while n >= 1 and (mips is None or len(levels) < mips):
        # low-frequency content plus noise, every level drawn independently so that the blend between levels shows
        base = rng.integers(0, 256, (max(n // 4, 1), max(n // 4, 1), 4), dtype=np.uint8)
        lvl = np.kron(base, np.ones((min(4, n), min(4, n), 1), np.uint8))[:n, :n].astype(np.int32)
        lvl = np.clip(lvl + rng.integers(-20, 21, lvl.shape), 0, 255).astype(np.uint8)
        levels.append(lvl)
        n //= 2
    return Image(levels, srgb=srgb)


def random_cubemap(seed, size, mips):
    rng = np.random.default_rng(seed)
    return Cubemap([(rng.random((6, max(size >> l, 1), max(size >> l, 1), 4)) * 0.5).astype(np.float32) for l in range(mips)])


def constant_cube(rgb, mips):
    return Cubemap([np.tile(np.append(rgb, 1.0).astype(np.float32), (6, max(1, 4 >> l), max(1, 4 >> l), 1)) for l in range(mips)])


MATERIAL = dict(
    base_color_factor=(0.9, 0.8, 0.7, 1.0), metallic_roughness_normal_occlusion=(0.8, 0.9, 1.0, 1.0),
    emissive_factor_alpha_cutoff=(0.30, 0.25, 0.35, 0.5),
    albedo_texture=0, albedo_texture_uv=1,                       # uv set 1 requested: the shader still reads uv0
    albedo_texture_transform=(1.7, 0.6, 0.21, -0.37),            # applied
    normal_texture=1, normal_texture_uv=0, normal_texture_transform=(0.8, 1.3, 0.05, 0.4),
    metallic_roughness_texture=2, metallic_roughness_texture_uv=0, metallic_roughness_texture_transform=(1.0, 1.0, 0.0, 0.0),
    occlusion_texture=3, occlusion_texture_uv=0, occlusion_texture_transform=(0.5, 0.5, 0.3, 0.3),
    emissive_texture=4, emissive_texture_uv=0, emissive_texture_transform=(5.0, 7.0, 0.5, 0.25),  # ignored by the shader
    specular_color_texture=5, specular_texture=5, specular_factor=(0.9, 0.7, 0.8, 0.85),
    material_type=abi.MAT_METAL_ROUGHNESS | abi.MAT_SPEC,
)


def make_scene(ggx_cube=None):
    sc = Scene()
    v = np.zeros(3, abi.VERTEX)
    for k in range(3):
        v[k]["p"] = (*POS[k], 0.0)
        v[k]["n"] = (*NRM[k], 0.0)
        v[k]["uv"] = (*UV0[k], *UV1[k])
        v[k]["t"] = tuple(TAN[k])
    sc.images = [random_image(1, 64, True), random_image(2, 32, False), random_image(3, 32, False), random_image(4, 16, False),
                 random_image(5, 64, False), random_image(6, 32, False, mips=3)]
    sc.add_material(**MATERIAL)
    sc.add_draw(sc.append_mesh(v, [np.arange(3, dtype=np.uint32)]), mat4_trs(), 0, 0)
    sc.camera, sc.sun_direction = CAM, SUN
    sc.env = EnvMap(brdf_lut=Image([np.tile(LUT.astype(np.float32), (4, 4, 1))]), lambertian=constant_cube(LAMBERTIAN, 1),
                    ggx=ggx_cube or constant_cube(GGX, 3), charlie=constant_cube(CHARLIE, 3), skybox=constant_cube(np.zeros(3), 1))
    return sc


# ---- float64 evaluation of one pixel -----------------------------------------------------------
def barycentrics(inv_vp, cam, x_fb, y_fb):
    n = np.cross(POS[1] - POS[0], POS[2] - POS[0])
    # a point far along the pixel's ray (reverse-Z: ndc z = near / view z, so 1e-4 is 100 units out): the
    # float32 rounding of the matrix then moves the direction by parts in 1e8, not in 1e4
    q = inv_vp @ np.array([2 * x_fb / W - 1, 1 - 2 * y_fb / H, 1e-4, 1.0])
    d = q[:3] / q[3] - cam
    p = cam + ((POS[0] - cam) @ n) / (d @ n) * d
    # p = l0 P0 + l1 P1 + l2 P2 with l0 + l1 + l2 = 1: areas of the sub-triangles
    area = n @ n
    l0 = np.cross(POS[1] - p, POS[2] - p) @ n / area
    l1 = np.cross(POS[2] - p, POS[0] - p) @ n / area
    return np.array([l0, l1, 1 - l0 - l1]), p


def expected_pixel(sc, ubo, x, y, ggx_cube=None):
    vp = np.array(ubo.view_proj, np.float64).reshape(4, 4).T
    inv, cam = np.linalg.inv(vp), np.array(ubo.camera_position, np.float64)
    lam, w_pos = barycentrics(inv, cam, x + 0.5, y + 0.5)
    ddx = barycentrics(inv, cam, x + 1.5, y + 0.5)[0] - lam
    ddy = barycentrics(inv, cam, x + 0.5, y + 1.5)[0] - lam
    uv, duvdx, duvdy = lam @ UV0, ddx @ UV0, ddy @ UV0
    mat = abi.default_material()
    for k, val in MATERIAL.items():
        mat[k] = val
    lods = []

    def tap(index, transform):
        s, o = np.array(transform[:2], np.float64), np.array(transform[2:], np.float64)
        val, lod = texture_grad(sc.images[index], uv * s + o, duvdx * s, duvdy * s)
        lods.append(lod)
        return val

    ident = (1.0, 1.0, 0.0, 0.0)
    albedo = tap(0, MATERIAL["albedo_texture_transform"]) * np.array(MATERIAL["base_color_factor"])
    nmap = tap(1, MATERIAL["normal_texture_transform"])[:3] * 2 - 1
    mr = tap(2, MATERIAL["metallic_roughness_texture_transform"])
    occ = tap(3, MATERIAL["occlusion_texture_transform"])[:3]
    emissive = np.array(MATERIAL["emissive_factor_alpha_cutoff"][:3]) * tap(4, ident)[:3]
    sf = np.array(MATERIAL["specular_factor"], np.float64)
    spec_rgb, spec_w = sf[:3] * tap(5, ident)[:3], sf[3] * tap(5, ident)[0]

    n_i, t_i = normalize(lam @ NRM), normalize(lam @ TAN[:, :3])
    bitangent = np.cross(n_i, t_i) * (TAN[0, 3] * TAN[1, 3] * TAN[2, 3])
    n = normalize(nmap[0] * t_i + nmap[1] * bitangent + nmap[2] * n_i)

    L = normalize(np.array(SUN))
    mrno = np.array(MATERIAL["metallic_roughness_normal_occlusion"], np.float64)
    metallic = np.clip(mrno[0] * mr[2], 0, 1)
    rough = np.clip(mr[1] * mrno[1], 0.04, 1)
    alpha = rough * rough
    base = albedo[:3]
    ior = float(mat["ior"])
    f0 = np.full(3, ((ior - 1) / (ior + 1)) ** 2)
    diffuse, specular = mix(base, 0.0, metallic), mix(f0, base, metallic)
    f0 = mix(np.minimum(f0 * spec_rgb, 1.0), base, metallic)   # KHR_materials_specular; `specular` keeps the plain mix
    weight = spec_w
    r0, r90 = specular, np.full(3, np.clip(specular.max() * 25, 0, 1))
    v = -normalize(w_pos - cam)
    h = normalize(L + v)
    ndv, ndl = np.clip(abs(n @ v), 0.001, 1), np.clip(n @ L, 0.001, 1)
    ndh, ldh, vdh = np.clip(n @ h, 0, 1), np.clip(L @ h, 0, 1), np.clip(v @ h, 0, 1)
    f = f_schlick(r0, r90, vdh)
    g, d = geometric_occlusion(ndl, ndv, alpha), microfacet_distribution(alpha, ndh)
    f90 = 2 * ldh * ldh * alpha - 0.5
    burley = (diffuse / np.pi) * (1 + f90 * (1 - ndl) ** 5) * (1 + f90 * (1 - ndv) ** 5)
    color = ndl * 5.0 * ((1 - f) * burley + f * g * d / (4 * ndl * ndv))
    if ggx_cube is None:
        ggx_light = GGX
    else:
        refl = normalize(2 * (n @ v) * n - v)  # reflect(-v, n)
        tapped = cube_lod(ggx_cube, refl, rough * (len(ggx_cube.mips) - 1))
        if tapped is None:
            return None, lods
        ggx_light = tapped[:3]
    fr = np.maximum(1 - rough, r0) - r0
    ks = r0 + fr * (1 - ndv) ** 5
    ibl_s = weight * ggx_light * (ks * LUT[0] + LUT[1])
    fss = weight * ks * LUT[0] + LUT[1]
    ems = 1 - (LUT[0] + LUT[1])
    favg = weight * (r0 + (1 - r0) / 21)
    fms = ems * fss * favg / (1 - favg * ems)
    ibl_d = (fms + diffuse * (1 - fss + fms)) * LAMBERTIAN
    color = np.clip(ibl_s + ibl_d + color + emissive, 0, 1) * occ[0]
    c = np.clip(color, 0, 1)
    srgb = np.where(c <= 0.0031308, 12.92 * c, 1.055 * c ** (1 / 2.4) - 0.055)
    return np.floor(srgb * 255 + 0.5), lods


def check(img, vis, ubo, sc, tol, ggx_cube=None):
    cov = covered(vis).reshape(H, W)
    checked, all_lods, worst = 0, [], 0
    for y in range(1, H - 1, 2):
        for x in range(1, W - 1, 3):
            # interior pixels only: the pixel and its right / lower neighbours inside the triangle
            if not (cov[y, x] and cov[y, x + 1] and cov[y + 1, x] and cov[y - 1, x] and cov[y, x - 1]):
                continue
            want, lods = expected_pixel(sc, ubo, x, y, ggx_cube)
            if want is None:
                continue
            got = img[y, x, :3].astype(int)
            worst = max(worst, int(np.abs(got - want).max()))
            assert np.abs(got - want).max() <= tol, (x, y, got, want, lods)
            checked += 1
            all_lods += lods
    assert checked > (80 if ggx_cube is not None else 150)  # cube taps whose footprint crosses a face edge are left out
    # the cases must exercise what they claim: magnified and minified taps, fractional levels
    lods = np.array(all_lods)
    assert (lods == 0).any() and (lods > 1.5).any() and ((lods % 1.0 > 0.2) & (lods % 1.0 < 0.8)).any()
    return checked, worst


def test_reference_geometry_is_what_the_test_assumes():
    """The float64 barycentrics reproduce the vertices' uv at the vertices and sum to one."""
    from renderer_amd.renderer import build_ubo
    ubo = build_ubo(W, H, CAM.position, CAM.direction, SUN, 1)
    vp = np.array(ubo.view_proj, np.float64).reshape(4, 4).T
    for k in range(3):
        clip = vp @ np.append(POS[k], 1.0)
        x_fb, y_fb = (clip[0] / clip[3] * 0.5 + 0.5) * W, (0.5 - clip[1] / clip[3] * 0.5) * H
        lam, p = barycentrics(np.linalg.inv(vp), np.array(ubo.camera_position, np.float64), x_fb, y_fb)
        assert np.allclose(lam, np.eye(3)[k], atol=1e-6) and np.allclose(p, POS[k], atol=1e-4)


def test_oracle_sampling_matches_float64(oracle):
    sc = make_scene()
    o = oracle.OracleRenderer(W, H, RendererConfig(cull_enabled=False))
    o.set_scene(sc)
    vis = o.render_visbuffer()
    check(o.shade(), vis, o.ubo, sc, tol=1)


def test_oracle_cubemap_tap_at_fractional_level(oracle):
    cube = random_cubemap(9, 32, 6)
    sc = make_scene(cube)
    o = oracle.OracleRenderer(W, H, RendererConfig(cull_enabled=False))
    o.set_scene(sc)
    vis = o.render_visbuffer()
    check(o.shade(), vis, o.ubo, sc, tol=1, ggx_cube=cube)


@pytest.mark.gpu
@pytest.mark.parametrize("with_cube", [False, True])
def test_hip_sampling_matches_float64(with_cube):
    from renderer_amd import Renderer
    cube = random_cubemap(9, 32, 6) if with_cube else None
    sc = make_scene(cube)
    g = Renderer(W, H, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    g.render()
    check(g.read_color(), g.read_visbuffer_u64(), g.ubo(), sc, tol=2, ggx_cube=cube)
    g.close()
