"""Synthetic scenes and environment maps.

The reference ships no assets (its scenes are external glTF files and KTX2 environment maps),
so every runnable workload here is procedural: the instanced icosphere field, a box atrium with
large, near-plane-crossing triangles, and tiny known-answer scenes.
"""
import math

import numpy as np

from . import abi
from .scene import Camera, Cubemap, EnvMap, Image, Scene, mat4_trs, mat4_trs_batch


# ------------------------------------------------------------------------------
# PCG32 (XSH-RR 64/32), the generator the field layout is specified with
# ------------------------------------------------------------------------------
class PCG32:
    MULT = 6364136223846793005
    MASK = (1 << 64) - 1

    def __init__(self, seed=42, stream=54):
        self.state = 0
        self.inc = ((stream << 1) | 1) & self.MASK
        self.next_u32()
        self.state = (self.state + seed) & self.MASK
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.MASK
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (self.next_u32() / 4294967296.0)


# ------------------------------------------------------------------------------
# Meshes
# ------------------------------------------------------------------------------
def _sphere_vertex(p):
    v = np.zeros((), abi.VERTEX)
    n = p / np.linalg.norm(p)
    v["p"] = (n[0], n[1], n[2], 0.0)
    v["n"] = (n[0], n[1], n[2], 0.0)
    phi = math.atan2(n[2], n[0])
    u = phi / (2 * math.pi) + 0.5
    t = math.acos(max(-1.0, min(1.0, n[1]))) / math.pi
    v["uv"] = (u * 4.0, t * 2.0, u, t)
    tx, tz = -math.sin(phi), math.cos(phi)
    v["t"] = (tx, 0.0, tz, 1.0)
    return v


def icosphere(subdivisions=5, lods=1):
    """Unit icosphere.  Returns (vertices, [lod0_indices, lod1_indices, ...], errors): LOD i is
    subdivision level ``subdivisions - i`` over the same vertex array, triangles emitted
    depth-first so that neighbouring triangles are neighbours in the index list."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    base = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
            (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    pts = [np.array(p, np.float64) / np.linalg.norm(p) for p in base]
    cache = {}

    def mid(a, b):
        k = (a, b) if a < b else (b, a)
        if k not in cache:
            m = pts[a] + pts[b]
            pts.append(m / np.linalg.norm(m))
            cache[k] = len(pts) - 1
        return cache[k]

    def subdiv(tri, depth, out):
        if depth == 0:
            out.extend(tri)
            return
        a, b, c = tri
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        for child in ((a, ab, ca), (ab, b, bc), (ab, bc, ca), (ca, bc, c)):
            subdiv(child, depth - 1, out)

    lod_indices, errors = [], []
    for li in range(lods):
        level = subdivisions - li
        assert level >= 0
        out = []
        for f in faces:
            subdiv(f, level, out)
        lod_indices.append(np.asarray(out, np.uint32))
        edge = 1.0514622 / (2 ** level)  # icosahedron edge on the unit sphere
        errors.append(0.0 if li == 0 else 1.0 - math.sqrt(max(0.0, 1.0 - (edge / 2) ** 2)))
    verts = np.array([_sphere_vertex(p) for p in pts], abi.VERTEX)
    return verts, lod_indices, errors


def quad_mesh(corners, uv_scale=1.0, two_tris=True):
    """A planar quad (4 corners, counter-clockwise seen from its front)."""
    c = np.asarray(corners, np.float64)
    n = np.cross(c[1] - c[0], c[3] - c[0])
    n /= np.linalg.norm(n)
    tg = (c[1] - c[0]) / np.linalg.norm(c[1] - c[0])
    v = np.zeros(4, abi.VERTEX)
    uvs = [(0, 0), (1, 0), (1, 1), (0, 1)]
    for i in range(4):
        v[i]["p"] = (*c[i], 0.0)
        v[i]["n"] = (*n, 0.0)
        v[i]["uv"] = (uvs[i][0] * uv_scale, uvs[i][1] * uv_scale, uvs[i][0], uvs[i][1])
        v[i]["t"] = (*tg, 1.0)
    idx = np.array([0, 1, 2, 0, 2, 3] if two_tris else [0, 1, 2], np.uint32)
    return v, idx


def grid_mesh(corners, nx, ny, uv_scale=1.0):
    """A planar quad tessellated into nx*ny cells (2 triangles each)."""
    c = np.asarray(corners, np.float64)
    n = np.cross(c[1] - c[0], c[3] - c[0])
    n /= np.linalg.norm(n)
    tg = (c[1] - c[0]) / np.linalg.norm(c[1] - c[0])
    v = np.zeros((ny + 1) * (nx + 1), abi.VERTEX)
    for j in range(ny + 1):
        for i in range(nx + 1):
            s, t = i / nx, j / ny
            p = c[0] * (1 - s) * (1 - t) + c[1] * s * (1 - t) + c[2] * s * t + c[3] * (1 - s) * t
            k = j * (nx + 1) + i
            v[k]["p"] = (*p, 0.0)
            v[k]["n"] = (*n, 0.0)
            v[k]["uv"] = (s * uv_scale, t * uv_scale, s, t)
            v[k]["t"] = (*tg, 1.0)
    idx = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            b, d, e = a + 1, a + nx + 1, a + nx + 2
            idx += [a, b, e, a, e, d]
    return v, np.asarray(idx, np.uint32)


def cylinder_mesh(radius, height, segments=24, rings=8):
    """Open cylinder around +Y, outward-facing."""
    v = np.zeros((rings + 1) * (segments + 1), abi.VERTEX)
    for j in range(rings + 1):
        for i in range(segments + 1):
            a = 2 * math.pi * i / segments
            cx, cz = math.cos(a), math.sin(a)
            k = j * (segments + 1) + i
            v[k]["p"] = (radius * cx, height * j / rings, radius * cz, 0.0)
            v[k]["n"] = (cx, 0.0, cz, 0.0)
            v[k]["uv"] = (i / segments * 2.0, j / rings * 4.0, i / segments, j / rings)
            v[k]["t"] = (-cz, 0.0, cx, 1.0)
    idx = []
    for j in range(rings):
        for i in range(segments):
            a = j * (segments + 1) + i
            b, d, e = a + 1, a + segments + 1, a + segments + 2
            idx += [a, d, e, a, e, b]
    return v, np.asarray(idx, np.uint32)


# ------------------------------------------------------------------------------
# Textures and environment
# ------------------------------------------------------------------------------
def _mip_chain(img):
    mips = [img]
    while mips[-1].shape[0] > 1 or mips[-1].shape[1] > 1:
        m = mips[-1].astype(np.float32)
        h, w = m.shape[:2]
        if h > 1:
            m = (m[0::2] + m[1::2]) * 0.5
        if w > 1:
            m = (m[:, 0::2] + m[:, 1::2]) * 0.5
        mips.append(np.round(m).astype(img.dtype) if img.dtype == np.uint8 else m.astype(np.float32))
    return mips


def checker_texture(size=256, cells=8, c0=(220, 200, 180, 255), c1=(60, 80, 120, 0), srgb=True):
    """Checkerboard RGBA8 with a full mip chain; c1's alpha of 0 makes it an alpha-test mask."""
    y, x = np.mgrid[0:size, 0:size]
    sel = ((x * cells // size) + (y * cells // size)) % 2
    img = np.where(sel[..., None] == 0, np.array(c0, np.uint8), np.array(c1, np.uint8)).astype(np.uint8)
    return Image(_mip_chain(img), srgb=srgb)


def noise_texture(size=128, seed=7, srgb=False, lo=0, hi=255):
    rng = np.random.default_rng(seed)
    img = rng.integers(lo, hi + 1, size=(size, size, 4), dtype=np.uint8)
    return Image(_mip_chain(img), srgb=srgb)


def normal_texture(size=128, bumps=6):
    y, x = np.mgrid[0:size, 0:size].astype(np.float64)
    fx = 0.35 * np.cos(2 * math.pi * bumps * x / size)
    fy = 0.35 * np.cos(2 * math.pi * bumps * y / size)
    n = np.stack([-fx, -fy, np.ones_like(fx)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    img = np.concatenate([np.round((n * 0.5 + 0.5) * 255), np.full((size, size, 1), 255.0)], -1).astype(np.uint8)
    return Image(_mip_chain(img), srgb=False)


def _cube_dirs(n):
    """(6, n, n, 3) directions through texel centres, Vulkan face order."""
    a = (np.arange(n) + 0.5) / n * 2 - 1
    s, t = np.meshgrid(a, a)  # s along x (columns), t along y (rows)
    one = np.ones_like(s)
    faces = [
        np.stack([one, -t, -s], -1), np.stack([-one, -t, s], -1),
        np.stack([s, one, t], -1), np.stack([s, -one, -t], -1),
        np.stack([s, -t, one], -1), np.stack([-s, -t, -one], -1),
    ]
    d = np.stack(faces, 0)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def _sky(d, sun, blur):
    """Analytic sky radiance: gradient + ground + a sun lobe widened by ``blur``."""
    up = d[..., 1]
    horizon = np.array([0.75, 0.8, 0.9])
    zenith = np.array([0.15, 0.3, 0.7])
    ground = np.array([0.25, 0.22, 0.2])
    tsky = np.clip(up, 0, 1)[..., None] ** 0.5
    col = np.where(up[..., None] >= 0, horizon * (1 - tsky) + zenith * tsky,
                   horizon * (1 - np.clip(-up * 4, 0, 1)[..., None]) + ground * np.clip(-up * 4, 0, 1)[..., None])
    cs = np.clip((d * np.asarray(sun)).sum(-1), 0, 1)
    lobe = np.exp((cs - 1.0) / (0.002 + blur * blur * 0.5))
    return (col * (1 - 0.5 * blur) + 0.5 * blur * np.array([0.5, 0.5, 0.55])
            + lobe[..., None] * np.array([4.0, 3.6, 3.0]) * (0.02 + (1 - blur)) * 0.5).astype(np.float32)


def sky_cubemap(size, mips, sun, blur_lo=0.0, blur_hi=1.0):
    out = []
    for l in range(mips):
        n = max(1, size >> l)
        blur = blur_lo + (blur_hi - blur_lo) * (l / max(1, mips - 1))
        rgb = _sky(_cube_dirs(n), sun, blur)
        out.append(np.concatenate([rgb, np.ones((6, n, n, 1), np.float32)], -1).astype(np.float32))
    return Cubemap(out)


def brdf_lut(size=64, samples=64):
    """Split-sum GGX (r, g) and Charlie (b) LUT, the table bake's ``brdf-lut`` produces
    (assets/shaders/brdf.comp.glsl), at reduced sample count; RGBA32F."""
    ndv = ((np.arange(size) + 0.5) / size)[None, :, None]
    rough = ((np.arange(size) + 0.5) / size)[:, None, None]
    i = np.arange(samples)
    u1 = ((i + 0.5) / samples)[None, None, :]
    bits = np.array([int(f"{k:032b}"[::-1], 2) for k in i], np.float64) * 2.3283064365386963e-10
    u2 = bits[None, None, :]
    a = rough * rough
    phi = 2 * math.pi * u1
    cos_t = np.sqrt((1 - u2) / (1 + (a * a - 1) * u2))
    sin_t = np.sqrt(1 - cos_t * cos_t)
    hx, hz = sin_t * np.cos(phi), cos_t
    vx, vz = np.sqrt(1 - ndv * ndv), ndv
    vdh = vx * hx + vz * hz
    lz = 2 * vdh * hz - vz
    ndl, ndh, vdh = np.clip(lz, 0, 1), np.clip(hz, 0, 1), np.clip(vdh, 0, 1)
    k = a / 2
    g = (ndl / (ndl * (1 - k) + k + 1e-9)) * (ndv / (ndv * (1 - k) + k))
    gv = g * vdh / (ndh * ndv + 1e-9)
    fc = (1 - vdh) ** 5
    ok = ndl > 0
    A = np.where(ok, (1 - fc) * gv, 0).mean(-1)
    B = np.where(ok, fc * gv, 0).mean(-1)
    Cc = np.clip(0.3 * (1 - rough[..., 0]) * (1 - ndv[..., 0]) + 0.05, 0, 1) * np.ones((size, size))
    lut = np.stack([A, B, Cc, np.ones_like(A)], -1).astype(np.float32)
    return Image([lut], srgb=False)


def default_env(sun=(-0.23623444, 0.9302103, 0.2808959), size=32, lut_size=64):
    s = np.asarray(sun, np.float64)
    s = s / np.linalg.norm(s)
    return EnvMap(
        brdf_lut=brdf_lut(lut_size),
        lambertian=sky_cubemap(max(8, size // 2), 1, s, 1.0, 1.0),
        ggx=sky_cubemap(size, int(math.log2(size)) + 1, s, 0.0, 1.0),
        charlie=sky_cubemap(size, int(math.log2(size)) + 1, s, 0.3, 1.0),
        skybox=sky_cubemap(size * 2, 1, s, 0.0, 1.0),
    )


# ------------------------------------------------------------------------------
# Scenes
# ------------------------------------------------------------------------------
def _shoemake(u1, u2, u3):
    a, b = np.sqrt(1 - u1), np.sqrt(u1)
    return np.stack([a * np.sin(2 * math.pi * u2), a * np.cos(2 * math.pi * u2),
                     b * np.sin(2 * math.pi * u3), b * np.cos(2 * math.pi * u3)], -1)


def field_transforms(n_instances, seed=42, stream=54):
    """Instance transforms of the icosphere field: jittered centred square XZ grid
    (side = ceil(sqrt(N)), spacing 2, jitter U(-0.3,0.3)), scale U(0.3,0.9), uniform random
    rotation, drawn row-major from PCG32(seed, stream)."""
    rng = PCG32(seed, stream)
    side = int(math.ceil(math.sqrt(n_instances)))
    vals = np.array([rng.next_u32() for _ in range(n_instances * 6)], np.float64).reshape(n_instances, 6) / 4294967296.0
    k = np.arange(n_instances)
    gx, gz = k % side, k // side
    x = (gx - (side - 1) / 2) * 2.0 + (vals[:, 0] * 0.6 - 0.3)
    z = (gz - (side - 1) / 2) * 2.0 + (vals[:, 1] * 0.6 - 0.3)
    scale = 0.3 + 0.6 * vals[:, 2]
    quat = _shoemake(vals[:, 3], vals[:, 4], vals[:, 5])
    return mat4_trs_batch(np.stack([x, np.zeros_like(x), z], -1), quat, scale[:, None])


def make_field(n_instances=1250, subdivisions=5, lods=1, alpha_fraction=0.0, with_env=True, seed=42, prepared=False):
    """Field-S (1 250 instances, 25.6 M triangles) / Field-XL (50 000, 1.02 G triangles).
    ``prepared``: run the sphere through the library's mesh preparation (cache-ordered meshlets,
    simplified LOD chain of up to ``lods`` levels) instead of the plain index-order meshlets over
    the coarser subdivision levels."""
    sc = Scene()
    verts, lod_idx, errs = icosphere(subdivisions, 1 if prepared else lods)
    mesh = sc.append_mesh_auto(verts, lod_idx[0], lods) if prepared else sc.append_mesh(verts, lod_idx, errs)
    sc.add_material()  # Material::default()
    xf = field_transforms(n_instances, seed)
    if alpha_fraction > 0:
        tex = sc.add_image(checker_texture(256, 8))
        sc.add_material(albedo_texture=tex, alpha_mode=1, emissive_factor_alpha_cutoff=(0, 0, 0, 0.5))
        for i in range(n_instances):
            masked = (i % max(1, int(round(1 / alpha_fraction)))) == 0
            sc.add_draw(mesh, xf[i], 1 if masked else 0, 1 if masked else 0)
    else:
        sc.add_draws(mesh, xf, 0, 0)
    if with_env:
        sc.env = default_env()
    d = np.array([0.0, 0.35, -1.0])
    d /= np.linalg.norm(d)
    sc.camera = Camera((0.0, 25.0, -60.0), tuple(d))
    return sc


def make_atrium(with_textures=True, columns=6, spheres=12, sphere_subdiv=3, seed=3):
    """A box hall (floor / walls / ceiling as a handful of very large triangles, some crossing
    the near plane behind the camera), rows of columns, an alpha-masked fence and a few
    spheres with varied material models.  Stand-in for an interior scene such as Sponza."""
    sc = Scene()
    sc.env = default_env()
    tex_checker = tex_noise = tex_normal = tex_mask = abi.NO_TEXTURE
    if with_textures:
        tex_checker = sc.add_image(checker_texture(256, 16, (200, 190, 170, 255), (90, 80, 70, 255)))
        tex_noise = sc.add_image(noise_texture(128, 7, False, 40, 255))
        tex_normal = sc.add_image(normal_texture(128, 6))
        tex_mask = sc.add_image(checker_texture(128, 16, (40, 160, 60, 255), (0, 0, 0, 0)))
    m_floor = sc.add_material(albedo_texture=tex_checker, normal_texture=tex_normal,
                              metallic_roughness_texture=tex_noise,
                              albedo_texture_transform=(8.0, 8.0, 0.25, 0.5),
                              metallic_roughness_normal_occlusion=(0.2, 0.9, 1, 1))
    m_wall = sc.add_material(base_color_factor=(0.8, 0.75, 0.7, 1.0),
                             metallic_roughness_normal_occlusion=(0.0, 0.8, 1, 1))
    m_col = sc.add_material(albedo_texture=tex_checker, occlusion_texture=tex_noise,
                            metallic_roughness_normal_occlusion=(0.0, 0.6, 1, 1))
    m_metal = sc.add_material(base_color_factor=(1.0, 0.8, 0.4, 1.0),
                              metallic_roughness_normal_occlusion=(1.0, 0.25, 1, 1))
    m_sg = sc.add_material(material_type=abi.MAT_SPEC_GLOSS, base_color_factor=(0.5, 0.2, 0.2, 1),
                           specular_glossiness_factor=(0.3, 0.3, 0.35, 0.7),
                           metallic_roughness_texture=tex_noise)
    m_cc = sc.add_material(material_type=abi.MAT_METAL_ROUGHNESS | abi.MAT_CLEARCOAT,
                           base_color_factor=(0.1, 0.3, 0.7, 1), clearcoat_factor=(0.8, 0.1, 0, 0),
                           metallic_roughness_normal_occlusion=(0.0, 0.5, 1, 1), clearcoat_texture=tex_normal)
    m_sheen = sc.add_material(material_type=abi.MAT_METAL_ROUGHNESS | abi.MAT_SHEEN,
                              base_color_factor=(0.5, 0.1, 0.4, 1), sheen_color_roughness_factor=(0.9, 0.6, 0.8, 0.4),
                              metallic_roughness_normal_occlusion=(0.0, 0.9, 1, 1))
    m_spec = sc.add_material(material_type=abi.MAT_METAL_ROUGHNESS | abi.MAT_SPEC,
                             base_color_factor=(0.2, 0.6, 0.3, 1), specular_factor=(1.0, 0.5, 0.5, 0.6),
                             emissive_factor_alpha_cutoff=(0.05, 0.02, 0.0, 1.0), emissive_texture=tex_noise,
                             specular_color_texture=tex_noise, specular_texture=tex_noise, ior=1.8)
    m_mask = sc.add_material(albedo_texture=tex_mask, alpha_mode=1, emissive_factor_alpha_cutoff=(0, 0, 0, 0.5),
                             albedo_texture_transform=(3.0, 1.0, 0.0, 0.0))
    ident = mat4_trs()
    L, Wd, Hh = 40.0, 12.0, 10.0
    # floor (+Y up), ceiling (-Y), two long walls, far wall, near wall — front faces look inward
    floor_v, floor_i = quad_mesh([(-Wd, 0, -L), (-Wd, 0, L), (Wd, 0, L), (Wd, 0, -L)], 1.0)
    sc.add_draw(sc.append_mesh(floor_v, [floor_i]), ident, m_floor)
    ceil_v, ceil_i = quad_mesh([(-Wd, Hh, -L), (Wd, Hh, -L), (Wd, Hh, L), (-Wd, Hh, L)], 1.0)
    sc.add_draw(sc.append_mesh(ceil_v, [ceil_i]), ident, m_wall)
    wl_v, wl_i = grid_mesh([(-Wd, 0, L), (-Wd, 0, -L), (-Wd, Hh, -L), (-Wd, Hh, L)], 8, 2)
    sc.add_draw(sc.append_mesh(wl_v, [wl_i]), ident, m_wall)
    wr_v, wr_i = grid_mesh([(Wd, 0, -L), (Wd, 0, L), (Wd, Hh, L), (Wd, Hh, -L)], 8, 2)
    sc.add_draw(sc.append_mesh(wr_v, [wr_i]), ident, m_wall)
    wf_v, wf_i = quad_mesh([(-Wd, 0, -L), (Wd, 0, -L), (Wd, Hh, -L), (-Wd, Hh, -L)])
    sc.add_draw(sc.append_mesh(wf_v, [wf_i]), ident, m_wall)
    wn_v, wn_i = quad_mesh([(Wd, 0, L), (-Wd, 0, L), (-Wd, Hh, L), (Wd, Hh, L)])
    sc.add_draw(sc.append_mesh(wn_v, [wn_i]), ident, m_wall)
    cyl_v, cyl_i = cylinder_mesh(0.6, Hh, 32, 12)
    cyl = sc.append_mesh(cyl_v, [cyl_i])
    for i in range(columns):
        z = -L + (i + 0.5) * (2 * L / columns)
        for x in (-Wd * 0.55, Wd * 0.55):
            sc.add_draw(cyl, mat4_trs((x, 0, z)), m_col)
    fence_v, fence_i = grid_mesh([(-Wd * 0.5, 0, -6.0), (Wd * 0.5, 0, -6.0), (Wd * 0.5, 3.0, -6.0), (-Wd * 0.5, 3.0, -6.0)], 6, 2, 1.0)
    # two-sided fence: the mask pipeline has cull NONE, but the per-triangle test still drops
    # back faces, so add the reverse winding too
    fence_i2 = np.concatenate([fence_i, fence_i.reshape(-1, 3)[:, ::-1].reshape(-1)])
    sc.add_draw(sc.append_mesh(fence_v, [fence_i2]), ident, m_mask, alpha_mode=1)
    sv, si, se = icosphere(sphere_subdiv, 1)
    sph = sc.append_mesh(sv, si, se)
    rng = np.random.default_rng(seed)
    mats = [m_metal, m_sg, m_cc, m_sheen, m_spec, m_col]
    for i in range(spheres):
        p = (rng.uniform(-Wd * 0.4, Wd * 0.4), rng.uniform(0.8, 4.0), rng.uniform(-L * 0.8, L * 0.3))
        q = _shoemake(rng.uniform(), rng.uniform(), rng.uniform())
        sc.add_draw(sph, mat4_trs(p, q, rng.uniform(0.5, 1.2)), mats[i % len(mats)])
    # a draw without a material (material_index 0xFFFF reads zeros)
    sc.add_draw(sph, mat4_trs((0.0, 6.0, -10.0), (0, 0, 0, 1), 0.7), 0xFFFF)
    d = np.array([0.05, 0.12, 1.0])
    d /= np.linalg.norm(d)
    sc.camera = Camera((0.5, 3.0, 18.0), tuple(d))
    sc.sun_direction = (-0.23623444, 0.9302103, 0.2808959)
    return sc


def make_triangle_scene(points, camera=None, alpha_mode=0, extra_tris=()):
    """Known-answer scene: one mesh made of the given world-space triangles, identity model."""
    sc = Scene()
    tris = [points] + list(extra_tris)
    v = np.zeros(3 * len(tris), abi.VERTEX)
    for ti, tri in enumerate(tris):
        p = np.asarray(tri, np.float64)
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n = n / (np.linalg.norm(n) + 1e-30)
        for k in range(3):
            v[3 * ti + k]["p"] = (*p[k], 0.0)
            v[3 * ti + k]["n"] = (*n, 0.0)
            v[3 * ti + k]["uv"] = ((0, 1, 0)[k], (0, 0, 1)[k], 0, 0)
            v[3 * ti + k]["t"] = (1, 0, 0, 1)
    idx = np.arange(3 * len(tris), dtype=np.uint32)
    sc.add_material()
    sc.add_draw(sc.append_mesh(v, [idx]), mat4_trs(), 0, alpha_mode)
    sc.camera = camera or Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    return sc
