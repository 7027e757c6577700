"""Known answers for the deferred shading pass (assets/shaders/shading.frag.glsl:180-531): the
oracle's output on a flat, untextured triangle under a *constant* environment is compared with the
shader's formulas evaluated here in float64, written from the GLSL and independent of the C
restatement.  The surface point comes from a ray / plane intersection (not from barycentrics), the
environment taps are constants, so what is pinned is the material model, the light terms, the
clamp, the shadow factor and the sRGB encode.  A second, GPU-marked test holds the HIP path to the
same numbers."""
import numpy as np
import pytest

from renderer_amd import RendererConfig, abi, synth
from renderer_amd.scene import Camera, Cubemap, EnvMap, Image
from helpers import covered

LUT = np.array([0.62, 0.11, 0.37, 0.0])        # brdf_lut: (scale, bias, charlie, -) everywhere
LAMBERTIAN = np.array([0.30, 0.34, 0.42])
GGX = np.array([0.22, 0.25, 0.31])
CHARLIE = np.array([0.12, 0.10, 0.15])
SUN = (0.3, 0.8, -0.4)
TRI = [(40.0, -2.0, -30.0), (-40.0, -2.0, -30.0), (0.0, 6.0, 60.0)]  # tilted ground plane, normal about +y
W, H = 96, 54


def constant_env():
    def cube(rgb, mips):
        return Cubemap([np.tile(np.append(rgb, 1.0).astype(np.float32), (6, max(1, 4 >> l), max(1, 4 >> l), 1)) for l in range(mips)])
    return EnvMap(brdf_lut=Image([np.tile(LUT.astype(np.float32), (4, 4, 1))]), lambertian=cube(LAMBERTIAN, 1),
                  ggx=cube(GGX, 3), charlie=cube(CHARLIE, 3), skybox=cube(np.zeros(3), 1))


def mix(a, b, t):
    return a * (1 - t) + b * t


def normalize(v):
    return v / np.linalg.norm(v)


def f_schlick(f0, f90, vdh):
    return f0 + (f90 - f0) * np.clip(1 - vdh, 0, 1) ** 5


def geometric_occlusion(ndl, ndv, a):
    r2 = a * a
    return (2 * ndl / (ndl + np.sqrt(r2 + (1 - r2) * ndl * ndl))) * (2 * ndv / (ndv + np.sqrt(r2 + (1 - r2) * ndv * ndv)))


def microfacet_distribution(a, ndh):
    r2 = a * a
    f = (ndh * r2 - ndh) * ndh + 1
    return r2 / (np.pi * f * f)


def ibl_ggx(ndv, rough, r0, weight, light):
    fr = np.maximum(1 - rough, r0) - r0
    ks = r0 + fr * (1 - ndv) ** 5
    return weight * light * (ks * LUT[0] + LUT[1])


def lambda_sheen_helper(x, ag):
    om = (1 - ag) ** 2
    a, b, c = mix(21.5473, 25.3245, om), mix(3.82987, 3.32435, om), mix(0.19823, 0.16801, om)
    d, e = mix(-1.97760, -1.27393, om), mix(-4.32054, -4.85967, om)
    return a / (1 + b * x ** c) + d * x + e


def lambda_sheen(ct, ag):
    if abs(ct) < 0.5:
        return np.exp(lambda_sheen_helper(ct, ag))
    return np.exp(2 * lambda_sheen_helper(0.5, ag) - lambda_sheen_helper(1 - ct, ag))


def expected_pixel(mat, n, w_pos, cam, shadowed):
    """shading.frag.glsl:421-531 for an untextured material (every sample_or_default returns its default)."""
    L = normalize(np.array(SUN))
    mrno = np.array(mat["metallic_roughness_normal_occlusion"], np.float64)
    base = np.array(mat["base_color_factor"], np.float64)[:3]
    mtype = int(mat["material_type"])
    metallic = np.clip(mrno[0] * 1.0, 0, 1)
    rough = 1.0 * mrno[1]
    sg = np.array(mat["specular_glossiness_factor"], np.float64)
    if mtype & abi.MAT_SPEC_GLOSS:
        rough = 1 - 1.0 * sg[3]
    else:
        rough = np.clip(rough, 0.04, 1)
    alpha = rough * rough
    ior = float(mat["ior"])
    f0 = np.full(3, ((ior - 1) / (ior + 1)) ** 2)
    diffuse, specular, weight = mix(base, 0.0, metallic), mix(f0, base, metallic), 1.0
    if mtype & abi.MAT_SPEC:
        sf = np.array(mat["specular_factor"], np.float64)
        f0 = mix(np.minimum(f0 * sf[:3], 1.0), base, metallic)
        weight = sf[3]
    if mtype & abi.MAT_SPEC_GLOSS:
        f0 = sg[:3] * 1.0
        diffuse = base * (1 - f0.max())
        specular = f0
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
    ibl_s = ibl_ggx(ndv, rough, r0, weight, GGX)
    fr = np.maximum(1 - rough, r0) - r0
    ks = r0 + fr * (1 - ndv) ** 5
    fss = weight * ks * LUT[0] + LUT[1]
    ems = 1 - (LUT[0] + LUT[1])
    favg = weight * (r0 + (1 - r0) / 21)
    fms = ems * fss * favg / (1 - favg * ems)
    ibl_d = (fms + diffuse * (1 - fss + fms)) * LAMBERTIAN
    emissive = np.array(mat["emissive_factor_alpha_cutoff"], np.float64)[:3]
    color = np.clip(ibl_s + ibl_d + color + emissive, 0, 1) * 1.0
    if mtype & abi.MAT_CLEARCOAT:
        cc = np.array(mat["clearcoat_factor"], np.float64)
        contrib = ibl_ggx(ndv, cc[1], np.full(3, ((ior - 1) / (ior + 1)) ** 2), 1.0, GGX)  # the clearcoat normal is n
        color = color * (1 - cc[0] * f_schlick(r0, r90, ndv)) + contrib
        cndl, cndv, cndh = np.clip(n @ L, 0.001, 1), np.clip(n @ v, 0.001, 1), np.clip(n @ h, 0.001, 1)
        cf = f_schlick(np.full(3, 0.04), np.ones(3), vdh)
        cg, cd = geometric_occlusion(cndl, cndv, cc[1] ** 2), microfacet_distribution(cc[1] ** 2, cndh)
        color = color + 5.0 * cndl * (cf * cg * cd / (4 * cndl * cndv))
    if mtype & abi.MAT_SHEEN:
        sh = np.array(mat["sheen_color_roughness_factor"], np.float64)
        color = color + CHARLIE * sh[:3] * LUT[2]
        sr = max(sh[3], 0.000001)
        ag = sr * sr
        dch = (2 + 1 / ag) * (1 - ndh * ndh) ** (0.5 / ag) / (2 * np.pi)
        vis = np.clip(1 / ((1 + 2 * lambda_sheen(ndl, ag)) * (4 * ndv * ndl)), 0, 1)
        color = color + 5.0 * ndl * sh[:3] * dch * vis
    color = color * (0.35 if shadowed else 1.0)
    c = np.clip(color, 0, 1)
    srgb = np.where(c <= 0.0031308, 12.92 * c, 1.055 * c ** (1 / 2.4) - 0.055)
    return np.floor(srgb * 255 + 0.5)


MATERIALS = {
    "default": {},
    "rough coloured dielectric": dict(base_color_factor=(0.25, 0.18, 0.10, 1.0), metallic_roughness_normal_occlusion=(0.0, 0.8, 1.0, 1.0)),
    "smooth metal": dict(base_color_factor=(0.30, 0.22, 0.08, 1.0), metallic_roughness_normal_occlusion=(1.0, 0.3, 1.0, 1.0)),
    "emissive, other ior": dict(base_color_factor=(0.1, 0.1, 0.1, 1.0), emissive_factor_alpha_cutoff=(0.05, 0.20, 0.02, 0.5), ior=1.33,
                                metallic_roughness_normal_occlusion=(0.2, 0.6, 1.0, 1.0)),
    "specular extension": dict(base_color_factor=(0.2, 0.2, 0.25, 1.0), metallic_roughness_normal_occlusion=(0.1, 0.5, 1.0, 1.0),
                               specular_factor=(0.9, 0.6, 0.3, 0.7), material_type=abi.MAT_METAL_ROUGHNESS | abi.MAT_SPEC),
    "specular glossiness": dict(base_color_factor=(0.2, 0.15, 0.1, 1.0), specular_glossiness_factor=(0.10, 0.12, 0.08, 0.55),
                                material_type=abi.MAT_SPEC_GLOSS),
    "clearcoat": dict(base_color_factor=(0.15, 0.05, 0.05, 1.0), metallic_roughness_normal_occlusion=(0.0, 0.7, 1.0, 1.0),
                      clearcoat_factor=(0.8, 0.25, 0.0, 0.0), material_type=abi.MAT_METAL_ROUGHNESS | abi.MAT_CLEARCOAT),
    "sheen": dict(base_color_factor=(0.10, 0.12, 0.2, 1.0), metallic_roughness_normal_occlusion=(0.0, 0.9, 1.0, 1.0),
                  sheen_color_roughness_factor=(0.3, 0.2, 0.25, 0.6), material_type=abi.MAT_METAL_ROUGHNESS | abi.MAT_SHEEN),
}


def kat_scene(name, shadowed):
    extra = [[(-200.0, 40.0, -200.0), (0.0, 40.0, 300.0), (200.0, 40.0, -200.0)]] if shadowed else []  # a roof between plane and sun
    sc = synth.make_triangle_scene(TRI, Camera((0.0, 14.0, -70.0), (0.0, 0.25, -1.0)), 0, extra)
    sc.materials = sc.materials[:0]
    sc.add_material(**MATERIALS[name])
    sc.env = constant_env()
    sc.sun_direction = SUN
    return sc


def check(img, vis, ubo, name, shadowed, tol):
    p = np.array(TRI, np.float64)
    n = normalize(np.cross(p[1] - p[0], p[2] - p[0]))
    vp = np.array(ubo.view_proj, np.float64).reshape(4, 4).T
    inv, cam = np.linalg.inv(vp), np.array(ubo.camera_position, np.float64)
    mat = abi.default_material()
    for k, val in MATERIALS[name].items():
        mat[k] = val
    ids = (np.asarray(vis).reshape(H, W) & np.uint64(0x7F)).astype(int) + ((np.asarray(vis).reshape(H, W) >> np.uint64(7)) & np.uint64(0x1FFFFFF)).astype(int) * 128
    checked, lit_values = 0, set()
    for y in range(0, H, 3):
        for x in range(0, W, 3):
            if not covered(vis).reshape(H, W)[y, x] or ids[y, x] != 0:  # only the ground triangle (meshlet instance 0, triangle 0)
                continue
            q = inv @ np.array([2 * (x + 0.5) / W - 1, 1 - 2 * (y + 0.5) / H, 0.5, 1.0])
            d = q[:3] / q[3] - cam
            t = ((p[0] - cam) @ n) / (d @ n)
            want = expected_pixel(mat, n, cam + t * d, cam, shadowed)
            got = img[y, x, :3].astype(int)
            assert np.abs(got - want).max() <= tol, (name, x, y, got, want)
            assert img[y, x, 3] == 255
            checked += 1
            lit_values.add(tuple(got))
    assert checked > 40 and len(lit_values) >= 2  # the view-dependent terms vary over the plane
    return checked


@pytest.mark.parametrize("shadowed", [False, True])
@pytest.mark.parametrize("name", list(MATERIALS))
def test_oracle_shading_matches_the_shader_formulas(oracle, name, shadowed):
    sc = kat_scene(name, shadowed)
    o = oracle.OracleRenderer(W, H, RendererConfig(cull_enabled=False))
    o.set_scene(sc)
    vis = o.render_visbuffer()
    check(o.shade(), vis, o.ubo, name, shadowed, tol=1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "smooth metal", "clearcoat", "sheen", "specular glossiness"])
def test_hip_shading_matches_the_shader_formulas(name):
    from renderer_amd import Renderer
    sc = kat_scene(name, True)
    g = Renderer(W, H, RendererConfig(cull_enabled=False))
    g.set_scene(sc)
    g.render()
    check(g.read_color(), g.read_visbuffer_u64(), g.ubo(), name, True, tol=2)
    g.close()
