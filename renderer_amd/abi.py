"""ctypes / numpy mirror of ``include/rdr_abi.h``.

Struct layouts are the reference's ``#[repr(C)]`` types (scenery/src/mesh.rs:17-123,
scenery/src/vertex.rs:4-11, renderer/src/renderer.rs:56-76); sizes are asserted at import.
"""
import ctypes as C

import numpy as np

VERTEX = np.dtype([("p", "<f4", 4), ("n", "<f4", 4), ("uv", "<f4", 4), ("t", "<f4", 4)])
MESHLET = np.dtype(
    [
        ("center", "<f4", 3),
        ("radius", "<f4"),
        ("cone_axis", "i1", 3),
        ("cone_cutoff", "i1"),
        ("vertex_offset", "<u4"),
        ("triangle_offset", "<u4"),
        ("triangle_count", "u1"),
        ("vertex_count", "u1"),
        ("_pad", "u1", 2),
    ]
)
MESH_LOD = np.dtype([("meshlet_offset", "<u4"), ("meshlet_count", "<u4"), ("error", "<f4"), ("_pad", "<u4")])
MESH = np.dtype(
    [
        ("center", "<f4", 3),
        ("radius", "<f4"),
        ("vertex_offset", "<u4"),
        ("vertex_count", "<u4"),
        ("lod_count", "<u4"),
        ("index_offset", "<u4"),
        ("lods", MESH_LOD, 8),
    ]
)
MESH_DRAW = np.dtype(
    [
        ("alpha_mode", "<u4"),
        ("meshlet_visibility_offset", "<u4"),
        ("model_index", "<u4"),
        ("mesh_index", "<u2"),
        ("material_index", "<u2"),
    ]
)
_MAT_VEC4 = [
    "base_color_factor",
    "metallic_roughness_normal_occlusion",
    "emissive_factor_alpha_cutoff",
    "specular_glossiness_factor",
    "clearcoat_factor",
    "sheen_color_roughness_factor",
    "albedo_texture_transform",
    "normal_texture_transform",
    "metallic_roughness_texture_transform",
    "emissive_texture_transform",
    "occlusion_texture_transform",
    "specular_factor",
]
_MAT_U32 = [
    "albedo_texture", "albedo_texture_uv", "normal_texture", "normal_texture_uv",
    "metallic_roughness_texture", "metallic_roughness_texture_uv", "emissive_texture", "emissive_texture_uv",
    "occlusion_texture", "occlusion_texture_uv", "clearcoat_texture", "clearcoat_texture_uv",
    "specular_color_texture", "specular_color_texture_uv", "specular_texture", "specular_texture_uv",
]
MATERIAL = np.dtype(
    [(n, "<f4", 4) for n in _MAT_VEC4]
    + [(n, "<u4") for n in _MAT_U32]
    + [("ior", "<f4"), ("alpha_mode", "<u4"), ("material_type", "<u4"), ("_pad", "<u4")]
)

assert VERTEX.itemsize == 64 and MESHLET.itemsize == 32 and MESH.itemsize == 160
assert MESH_DRAW.itemsize == 16 and MATERIAL.itemsize == 272

MAT_METAL_ROUGHNESS, MAT_SPEC_GLOSS, MAT_CLEARCOAT, MAT_SHEEN, MAT_SPEC = 1, 2, 4, 8, 16
NO_TEXTURE = 0xFFFFFFFF
FORMAT_RGBA8_UNORM, FORMAT_RGBA8_SRGB, FORMAT_RGBA32F = 0, 1, 2
# rdr_frame_status flags
(FRAME_DEFERRED_OVERFLOW, FRAME_SETUP_OVERFLOW, FRAME_BIN_OVERFLOW, FRAME_MID_OVERFLOW, FRAME_FRAG_OVERFLOW,
 FRAME_LIST_OVERFLOW, FRAME_SHADOW_STACK, FRAME_PEER_TIMEOUT) = 1, 2, 4, 8, 16, 32, 64, 128
FORMAT_BC1, FORMAT_BC2, FORMAT_BC3, FORMAT_BC4 = 3, 4, 5, 6  # block-compressed, decoded to RGBA8_UNORM at upload
FORMAT_BC7, FORMAT_BC7_SRGB = 7, 8  # decoded to RGBA8_UNORM / RGBA8_SRGB at upload
VIS_CLEAR = 0x7FFFFFFFFFFFFFFF


def default_material():
    """``Material::default()`` (scenery/src/mesh.rs:125-161)."""
    m = np.zeros((), MATERIAL)
    one = (1.0, 1.0, 1.0, 1.0)
    for n in ("base_color_factor", "metallic_roughness_normal_occlusion", "specular_glossiness_factor",
              "clearcoat_factor", "sheen_color_roughness_factor", "specular_factor"):
        m[n] = one
    m["emissive_factor_alpha_cutoff"] = (0.0, 0.0, 0.0, 1.0)
    for n in ("albedo_texture_transform", "normal_texture_transform", "metallic_roughness_texture_transform",
              "emissive_texture_transform", "occlusion_texture_transform"):
        m[n] = (1.0, 1.0, 0.0, 0.0)
    for n in _MAT_U32:
        m[n] = NO_TEXTURE if n.endswith("_texture") else 0
    m["ior"] = 1.5
    m["alpha_mode"] = 0
    m["material_type"] = MAT_METAL_ROUGHNESS
    return m


class Ubo(C.Structure):
    _fields_ = [
        ("view_proj", C.c_float * 16),
        ("view", C.c_float * 16),
        ("proj", C.c_float * 16),
        ("camera_position", C.c_float * 3),
        ("_pad0", C.c_uint32),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
        ("depth_pyramid_width", C.c_float),
        ("depth_pyramid_height", C.c_float),
        ("frustum", C.c_float * 4),
        ("draw_count", C.c_uint32),
        ("lod_enabled", C.c_uint32),
        ("cull_enabled", C.c_uint32),
        ("lod_target", C.c_float),
        ("screen_width", C.c_float),
        ("screen_height", C.c_float),
        ("_pad1", C.c_uint32 * 2),
        ("sun_direction", C.c_float * 4),
    ]


assert C.sizeof(Ubo) == 288 and Ubo.sun_direction.offset == 272 and Ubo.frustum.offset == 224


class ImageDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("mip_count", C.c_uint32), ("format", C.c_uint32)]


class CubemapDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_uint32), ("mip_count", C.c_uint32),
                ("format", C.c_uint32), ("_pad", C.c_uint32)]


class EnvDesc(C.Structure):
    _fields_ = [("brdf_lut", ImageDesc), ("skybox", CubemapDesc), ("lambertian", CubemapDesc),
                ("ggx", CubemapDesc), ("charlie", CubemapDesc)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p),
        ("indices", C.c_void_p),
        ("meshlet_vertices", C.c_void_p),
        ("meshlet_triangles", C.c_void_p),
        ("meshlets", C.c_void_p),
        ("meshes", C.c_void_p),
        ("mesh_draws", C.c_void_p),
        ("materials", C.c_void_p),
        ("global_transforms", C.c_void_p),
        ("images", C.POINTER(ImageDesc)),
        ("env", C.POINTER(EnvDesc)),
        ("vertex_count", C.c_uint32),
        ("index_count", C.c_uint32),
        ("meshlet_vertex_count", C.c_uint32),
        ("meshlet_triangle_bytes", C.c_uint32),
        ("meshlet_count", C.c_uint32),
        ("mesh_count", C.c_uint32),
        ("draw_count", C.c_uint32),
        ("material_count", C.c_uint32),
        ("transform_count", C.c_uint32),
        ("image_count", C.c_uint32),
        ("meshlet_visibility_count", C.c_uint32),
        ("_pad", C.c_uint32),
    ]


class Config(C.Structure):
    _fields_ = [("lod_enabled", C.c_uint32), ("cull_enabled", C.c_uint32),
                ("collect_stats", C.c_uint32), ("time_kernels", C.c_uint32), ("skybox", C.c_uint32)]


class MeshLod(C.Structure):
    _fields_ = [("meshlet_offset", C.c_uint32), ("meshlet_count", C.c_uint32), ("error", C.c_float), ("_pad", C.c_uint32)]


class Mesh(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("vertex_offset", C.c_uint32), ("vertex_count", C.c_uint32),
                ("lod_count", C.c_uint32), ("index_offset", C.c_uint32), ("lods", MeshLod * 8)]


class BuiltMesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("indices", C.c_void_p), ("meshlet_vertices", C.c_void_p),
                ("meshlet_triangles", C.c_void_p), ("meshlets", C.c_void_p),
                ("vertex_count", C.c_uint32), ("index_count", C.c_uint32), ("meshlet_vertex_count", C.c_uint32),
                ("meshlet_triangle_bytes", C.c_uint32), ("meshlet_count", C.c_uint32), ("_pad", C.c_uint32),
                ("mesh", Mesh)]


assert C.sizeof(Mesh) == 160

STAT_KERNELS = 8
KERNEL_NAMES = ["clear", "draw_cull", "meshlet_cull", "raster", "shadow", "big_raster", "shade", "classify"]


class Stats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_float * STAT_KERNELS),
        ("draws_visible", C.c_uint64),
        ("commands", C.c_uint64),
        ("meshlets_tested", C.c_uint64),
        ("meshlets_visible", C.c_uint64),
        ("triangles_submitted", C.c_uint64),
        ("triangles_tested", C.c_uint64),
        ("triangles_rasterised", C.c_uint64),
        ("triangles_big", C.c_uint64),
        ("fragments", C.c_uint64),
        ("pixels_covered", C.c_uint64),
        ("vertices_transformed", C.c_uint64),
        ("frames_timed", C.c_uint64),
        ("bin_entries", C.c_uint64),
        ("bin_max", C.c_uint64),
    ]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[1:]}
        d["kernel_ms"] = {KERNEL_NAMES[i]: float(self.kernel_ms[i]) for i in range(STAT_KERNELS)}
        return d
