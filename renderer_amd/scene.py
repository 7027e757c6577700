"""Host-side scene model: the arrays the reference's ``Scene`` / ``Geometry`` hold
(scenery/src/scene.rs:36-60, scenery/src/mesh.rs:51-59) as numpy arrays in the GPU layouts,
a meshlet builder with the reference's limits (64 vertices / 124 triangles,
scenery/src/mesh.rs:213-215) and the ``Camera`` of scenery/src/camera.rs.
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi

MAX_MESHLET_VERTICES = 64
MAX_MESHLET_TRIANGLES = 124
# Scene::default sun direction (scenery/src/scene.rs:115)
DEFAULT_SUN = (-0.23623444, 0.9302103, 0.2808959)


@dataclass
class Camera:
    """scenery/src/camera.rs:6-9 — the view looks along ``-direction``."""
    position: tuple = (0.0, 0.0, 0.0)
    direction: tuple = (0.0, 0.0, 1.0)


@dataclass
class RendererConfig:
    """scenery/src/lib.rs:7-14, the fields that reach the GPU."""
    lod_enabled: bool = False
    cull_enabled: bool = True
    collect_stats: bool = True
    time_kernels: bool = False
    skybox: bool = False  # run the skybox pass (renderer/src/renderer/pass.rs:397-445) on uncovered pixels


@dataclass
class Image:
    """Decoded texture: ``mips[i]`` is (h, w, 4) uint8 or float32."""
    mips: list
    srgb: bool = False

    @property
    def format(self):
        if self.mips[0].dtype == np.float32:
            return abi.FORMAT_RGBA32F
        return abi.FORMAT_RGBA8_SRGB if self.srgb else abi.FORMAT_RGBA8_UNORM


@dataclass
class BlockImage:
    """Block-compressed texture as a .dds file holds it: ``data`` is the mip chain's blocks (uint8),
    ``format`` one of ``abi.FORMAT_BC1`` .. ``FORMAT_BC7_SRGB``.  The library expands it to RGBA8 on the
    GPU when the scene is uploaded."""
    data: np.ndarray
    width: int
    height: int
    mip_count: int
    format: int

    def level_bytes(self, level):
        w, h = max(1, self.width >> level), max(1, self.height >> level)
        return ((w + 3) // 4) * ((h + 3) // 4) * (8 if self.format in (abi.FORMAT_BC1, abi.FORMAT_BC4) else 16)


@dataclass
class Cubemap:
    """``mips[i]`` is (6, n, n, 4), faces in +X,-X,+Y,-Y,+Z,-Z order."""
    mips: list

    @property
    def format(self):
        return abi.FORMAT_RGBA32F if self.mips[0].dtype == np.float32 else abi.FORMAT_RGBA8_UNORM


@dataclass
class EnvMap:
    brdf_lut: Image = None
    lambertian: Cubemap = None
    ggx: Cubemap = None
    charlie: Cubemap = None
    skybox: Cubemap = None


@dataclass
class Scene:
    vertices: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.VERTEX))
    indices: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    meshlet_vertices: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    meshlet_triangles: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint8))
    meshlets: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.MESHLET))
    meshes: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.MESH))
    mesh_draws: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.MESH_DRAW))
    materials: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.MATERIAL))
    global_transforms: np.ndarray = field(default_factory=lambda: np.zeros((0, 16), np.float32))
    images: list = field(default_factory=list)
    env: EnvMap = None
    meshlet_visibility_count: int = 0
    camera: Camera = field(default_factory=Camera)
    sun_direction: tuple = DEFAULT_SUN

    # ---- construction helpers -------------------------------------------------
    def append_mesh(self, vertices, lod_indices, lod_errors=None):
        """Counterpart of ``Geometry::append_mesh`` (scenery/src/mesh.rs:197-349) for
        pre-simplified LOD index lists: builds meshlets for every LOD.  Returns the mesh index."""
        assert vertices.dtype == abi.VERTEX
        mesh = np.zeros((), abi.MESH)
        pos = vertices["p"][:, :3].astype(np.float32)
        center = (pos.sum(axis=0, dtype=np.float32) / np.float32(len(pos))).astype(np.float32)
        mesh["center"] = center
        mesh["radius"] = np.sqrt(((pos - center) ** 2).sum(axis=1)).max().astype(np.float32)
        mesh["vertex_offset"] = len(self.vertices)
        mesh["vertex_count"] = len(vertices)
        mesh["index_offset"] = len(self.indices)
        mesh["lod_count"] = len(lod_indices)
        assert 1 <= len(lod_indices) <= 8
        mv_all, mt_all, ml_all = [self.meshlet_vertices], [self.meshlet_triangles], [self.meshlets]
        mv_off, mt_off, ml_off = len(self.meshlet_vertices), len(self.meshlet_triangles), len(self.meshlets)
        for li, idx in enumerate(lod_indices):
            idx = np.asarray(idx, np.uint32)
            mv, mt, ml = build_meshlets(pos, idx)
            ml["vertex_offset"] += mv_off
            ml["triangle_offset"] += mt_off
            mesh["lods"][li]["meshlet_offset"] = ml_off
            mesh["lods"][li]["meshlet_count"] = len(ml)
            mesh["lods"][li]["error"] = 0.0 if li == 0 or lod_errors is None else lod_errors[li]
            mv_all.append(mv); mt_all.append(mt); ml_all.append(ml)
            mv_off += len(mv); mt_off += len(mt); ml_off += len(ml)
        self.meshlet_vertices = np.concatenate(mv_all).astype(np.uint32)
        self.meshlet_triangles = np.concatenate(mt_all).astype(np.uint8)
        self.meshlets = np.concatenate(ml_all)
        self.vertices = np.concatenate([self.vertices, vertices])
        self.indices = np.concatenate([self.indices, np.asarray(lod_indices[0], np.uint32)])
        self.meshes = np.concatenate([self.meshes, mesh.reshape(1)])
        return len(self.meshes) - 1

    def append_mesh_auto(self, vertices, indices, max_lods=8, fast_mode=False):
        """``Geometry::append_mesh`` (scenery/src/mesh.rs:197-349) from a raw triangle list: the
        library de-duplicates and reorders the vertices, builds the meshlets and simplifies the
        LOD chain (``fast_mode``: the reference's ``RendererConfig::fast_mode``, a cheaper index ordering).
        Returns the mesh index."""
        from .meshprep import build_mesh
        b = build_mesh(vertices, indices, max_lods, fast_mode)
        mesh = b.mesh.copy()
        mesh["vertex_offset"] = len(self.vertices)
        mesh["index_offset"] = len(self.indices)
        ml = b.meshlets.copy()
        ml["vertex_offset"] += len(self.meshlet_vertices)
        ml["triangle_offset"] += len(self.meshlet_triangles)
        for li in range(int(mesh["lod_count"])):
            mesh["lods"][li]["meshlet_offset"] += len(self.meshlets)
        self.meshlet_vertices = np.concatenate([self.meshlet_vertices, b.meshlet_vertices]).astype(np.uint32)
        self.meshlet_triangles = np.concatenate([self.meshlet_triangles, b.meshlet_triangles]).astype(np.uint8)
        self.meshlets = np.concatenate([self.meshlets, ml])
        self.vertices = np.concatenate([self.vertices, b.vertices])
        self.indices = np.concatenate([self.indices, b.indices])
        self.meshes = np.concatenate([self.meshes, mesh.reshape(1)])
        return len(self.meshes) - 1

    def add_draw(self, mesh_index, transform, material_index=0xFFFF, alpha_mode=0):
        """One node + one primitive, as ``build_scene_mesh_draws`` pushes them
        (scenery/src/scene.rs:601-651)."""
        model_index = len(self.global_transforms)
        self.global_transforms = np.concatenate(
            [self.global_transforms, np.asarray(transform, np.float32).reshape(1, 16)])
        d = np.zeros(1, abi.MESH_DRAW)
        d["alpha_mode"] = alpha_mode
        d["meshlet_visibility_offset"] = self.meshlet_visibility_count
        d["model_index"] = model_index
        d["mesh_index"] = mesh_index
        d["material_index"] = material_index
        self.mesh_draws = np.concatenate([self.mesh_draws, d])
        self.meshlet_visibility_count += int(self.meshes[mesh_index]["lods"][0]["meshlet_count"])
        return len(self.mesh_draws) - 1

    def add_draws(self, mesh_index, transforms, material_index=0xFFFF, alpha_mode=0):
        """Vectorised ``add_draw`` for many instances of one mesh."""
        transforms = np.asarray(transforms, np.float32).reshape(-1, 16)
        n = len(transforms)
        base = len(self.global_transforms)
        self.global_transforms = np.concatenate([self.global_transforms, transforms])
        d = np.zeros(n, abi.MESH_DRAW)
        mc = int(self.meshes[mesh_index]["lods"][0]["meshlet_count"])
        d["alpha_mode"] = alpha_mode
        d["meshlet_visibility_offset"] = self.meshlet_visibility_count + mc * np.arange(n, dtype=np.uint32)
        d["model_index"] = base + np.arange(n, dtype=np.uint32)
        d["mesh_index"] = mesh_index
        d["material_index"] = material_index
        self.mesh_draws = np.concatenate([self.mesh_draws, d])
        self.meshlet_visibility_count += mc * n

    def add_material(self, **kw):
        m = abi.default_material().reshape(1).copy()
        for k, v in kw.items():
            m[k] = v
        self.materials = np.concatenate([self.materials, m])
        return len(self.materials) - 1

    def add_image(self, image):
        self.images.append(image)
        return len(self.images) - 1

    def triangle_count(self, lod=0):
        """Triangles submitted per frame when every draw selects ``lod``."""
        total = 0
        per_mesh = []
        for m in self.meshes:
            L = m["lods"][min(lod, int(m["lod_count"]) - 1)]
            o, c = int(L["meshlet_offset"]), int(L["meshlet_count"])
            per_mesh.append(int(self.meshlets["triangle_count"][o:o + c].sum()))
        per_mesh = np.asarray(per_mesh, np.int64)
        if len(self.mesh_draws):
            total = int(per_mesh[self.mesh_draws["mesh_index"]].sum())
        return total

    # ---- C ABI ----------------------------------------------------------------
    def to_desc(self):
        """Build an ``rdr_scene_desc`` over this scene's arrays.  Returns (desc, keepalive)."""
        keep = []

        def ptr(a):
            a = np.ascontiguousarray(a)
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p) if a.size else None

        def image_desc(img):
            if img is None:  # an image the loader skipped (embedded / missing file): data == NULL samples as zero
                return abi.ImageDesc(None, 0, 0, 0, 0)
            if isinstance(img, BlockImage):
                blob = np.ascontiguousarray(img.data, np.uint8).reshape(-1)
                need = sum(img.level_bytes(l) for l in range(img.mip_count))
                if blob.size < need:
                    raise ValueError(f"block-compressed image holds {blob.size} bytes, its mip chain needs {need}")
                keep.append(blob)
                return abi.ImageDesc(blob.ctypes.data_as(C.c_void_p), img.width, img.height, img.mip_count, img.format)
            blob = np.concatenate([np.ascontiguousarray(m).reshape(-1).view(np.uint8) for m in img.mips])
            keep.append(blob)
            h, w = img.mips[0].shape[:2]
            return abi.ImageDesc(blob.ctypes.data_as(C.c_void_p), w, h, len(img.mips), img.format)

        def cube_desc(cm):
            if cm is None:
                return abi.CubemapDesc(None, 0, 0, 0, 0)
            blob = np.concatenate([np.ascontiguousarray(m).reshape(-1).view(np.uint8) for m in cm.mips])
            keep.append(blob)
            return abi.CubemapDesc(blob.ctypes.data_as(C.c_void_p), cm.mips[0].shape[1], len(cm.mips), cm.format, 0)

        d = abi.SceneDesc()
        d.vertices = ptr(self.vertices)
        d.indices = ptr(self.indices.astype(np.uint32))
        d.meshlet_vertices = ptr(self.meshlet_vertices.astype(np.uint32))
        d.meshlet_triangles = ptr(self.meshlet_triangles.astype(np.uint8))
        d.meshlets = ptr(self.meshlets)
        d.meshes = ptr(self.meshes)
        d.mesh_draws = ptr(self.mesh_draws)
        d.materials = ptr(self.materials)
        d.global_transforms = ptr(self.global_transforms.astype(np.float32))
        if self.images:
            arr = (abi.ImageDesc * len(self.images))(*[image_desc(i) for i in self.images])
            keep.append(arr)
            d.images = C.cast(arr, C.POINTER(abi.ImageDesc))
        if self.env is not None:
            e = abi.EnvDesc()
            if self.env.brdf_lut is not None:
                e.brdf_lut = image_desc(self.env.brdf_lut)
            e.skybox = cube_desc(self.env.skybox)
            e.lambertian = cube_desc(self.env.lambertian)
            e.ggx = cube_desc(self.env.ggx)
            e.charlie = cube_desc(self.env.charlie)
            keep.append(e)
            d.env = C.pointer(e)
        d.vertex_count = len(self.vertices)
        d.index_count = len(self.indices)
        d.meshlet_vertex_count = len(self.meshlet_vertices)
        d.meshlet_triangle_bytes = len(self.meshlet_triangles)
        d.meshlet_count = len(self.meshlets)
        d.mesh_count = len(self.meshes)
        d.draw_count = len(self.mesh_draws)
        d.material_count = len(self.materials)
        d.transform_count = len(self.global_transforms)
        d.image_count = len(self.images)
        d.meshlet_visibility_count = self.meshlet_visibility_count
        return d, keep


# ------------------------------------------------------------------------------
# Meshlet builder
# ------------------------------------------------------------------------------

def _bounding_sphere(p):
    """Ritter's bounding sphere."""
    p = p.astype(np.float64)
    a = p[np.argmax(((p - p[0]) ** 2).sum(1))]
    b = p[np.argmax(((p - a) ** 2).sum(1))]
    c = (a + b) / 2
    r = math.sqrt(((a - b) ** 2).sum()) / 2
    for q in p:
        d = math.sqrt(((q - c) ** 2).sum())
        if d > r:
            nr = (r + d) / 2
            c = c + (q - c) * ((nr - r) / d)
            r = nr
    return c.astype(np.float32), np.float32(r * (1 + 1e-6) + 1e-12)


def _meshlet_bounds(pos, tris):
    """Bounding sphere + normal cone, quantised like meshoptimizer's
    ``meshopt_computeMeshletBounds`` (cone_axis_s8 / cone_cutoff_s8, used at
    scenery/src/mesh.rs:351-372): conservative after quantisation."""
    center, radius = _bounding_sphere(pos[np.unique(tris)])
    a, b, c = pos[tris[:, 0]].astype(np.float64), pos[tris[:, 1]].astype(np.float64), pos[tris[:, 2]].astype(np.float64)
    n = np.cross(b - a, c - a)
    ln = np.sqrt((n ** 2).sum(1))
    ok = ln > 0
    axis_s8 = np.zeros(3, np.int8)
    cutoff_s8 = 127
    if ok.any():
        n = n[ok] / ln[ok, None]
        axis = n.sum(0)
        la = math.sqrt((axis ** 2).sum())
        if la > 1e-12:
            axis /= la
            mindp = float((n @ axis).min())
            if mindp > 0.1:
                cutoff = math.sqrt(max(0.0, 1.0 - mindp * mindp))
                q = np.clip(np.round(axis * 127.0), -127, 127)
                err = float(np.abs(q / 127.0 - axis).sum())
                axis_s8 = q.astype(np.int8)
                cutoff_s8 = min(127, int(127.0 * cutoff + 127.0 * err + 1))
    return center, radius, axis_s8, cutoff_s8


def build_meshlets(positions, indices):
    """Greedy scan in index order: a triangle joins the open meshlet unless it would exceed 64
    vertices or 124 triangles.  Each meshlet's triangle bytes are padded to a multiple of 4, as
    meshoptimizer lays them out.  Returns (meshlet_vertices, meshlet_triangles, meshlets) with
    offsets local to the returned arrays."""
    idx = np.asarray(indices, np.uint32).reshape(-1, 3)
    mv, mt, out = [], [], []
    local = {}
    cur_v, cur_t = [], []

    def flush():
        nonlocal local, cur_v, cur_t
        if not cur_t:
            return
        m = np.zeros((), abi.MESHLET)
        tri_global = np.asarray([[cur_v[a], cur_v[b], cur_v[c]] for a, b, c in cur_t], np.int64)
        ctr, rad, ax, co = _meshlet_bounds(positions, tri_global)
        m["center"], m["radius"], m["cone_axis"], m["cone_cutoff"] = ctr, rad, ax, co
        m["vertex_offset"] = len(mv)
        m["triangle_offset"] = len(mt)
        m["triangle_count"] = len(cur_t)
        m["vertex_count"] = len(cur_v)
        mv.extend(cur_v)
        for t in cur_t:
            mt.extend(t)
        while len(mt) % 4:
            mt.append(0)
        out.append(m)
        local, cur_v, cur_t = {}, [], []

    for a, b, c in idx.tolist():
        new = len({a, b, c} - local.keys())
        if len(cur_v) + new > MAX_MESHLET_VERTICES or len(cur_t) + 1 > MAX_MESHLET_TRIANGLES:
            flush()
        tri = []
        for v in (a, b, c):
            if v not in local:
                local[v] = len(cur_v)
                cur_v.append(v)
            tri.append(local[v])
        cur_t.append(tri)
    flush()
    meshlets = np.array(out, abi.MESHLET) if out else np.zeros(0, abi.MESHLET)
    return np.asarray(mv, np.uint32), np.asarray(mt, np.uint8), meshlets


# ------------------------------------------------------------------------------
# Transform helpers (column-major mat4 as 16 floats)
# ------------------------------------------------------------------------------

def mat4_trs(translation=(0, 0, 0), quat=(0, 0, 0, 1), scale=(1, 1, 1)):
    """``Mat4::from_scale_rotation_translation`` (glam), column-major."""
    x, y, z, w = [np.float32(v) for v in quat]
    sx, sy, sz = [np.float32(v) for v in (scale if np.ndim(scale) else (scale,) * 3)]
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    m = np.zeros(16, np.float32)
    m[0:3] = np.array([1 - (yy + zz), xy + wz, xz - wy], np.float32) * sx
    m[4:7] = np.array([xy - wz, 1 - (xx + zz), yz + wx], np.float32) * sy
    m[8:11] = np.array([xz + wy, yz - wx, 1 - (xx + yy)], np.float32) * sz
    m[12:15] = np.asarray(translation, np.float32)
    m[15] = 1.0
    return m


def mat4_trs_batch(translations, quats, scales):
    n = len(translations)
    q = np.asarray(quats, np.float32)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    s = np.asarray(scales, np.float32).reshape(n, -1)
    if s.shape[1] == 1:
        s = np.repeat(s, 3, axis=1)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    m = np.zeros((n, 16), np.float32)
    m[:, 0], m[:, 1], m[:, 2] = (1 - (yy + zz)) * s[:, 0], (xy + wz) * s[:, 0], (xz - wy) * s[:, 0]
    m[:, 4], m[:, 5], m[:, 6] = (xy - wz) * s[:, 1], (1 - (xx + zz)) * s[:, 1], (yz + wx) * s[:, 1]
    m[:, 8], m[:, 9], m[:, 10] = (xz + wy) * s[:, 2], (yz - wx) * s[:, 2], (1 - (xx + yy)) * s[:, 2]
    m[:, 12:15] = np.asarray(translations, np.float32)
    m[:, 15] = 1.0
    return m
