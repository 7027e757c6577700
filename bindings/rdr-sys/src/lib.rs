//! Raw bindings to `include/rdr_abi.h`, one item per declaration of the header, in the header's order.
//!
//! The scene structs are the reference's own `#[repr(C)]` types (`scenery::vertex::Vertex`,
//! `scenery::mesh::{Meshlet, Mesh, MeshLod, MeshDraw, Material}`): a host that links `scenery` casts
//! pointers to its vectors; the definitions below have the same layouts for hosts that do not.
//! Not compiled in the environment this library was built in (no Rust toolchain there);
//! `tests/test_bindings.py` checks that every exported symbol of the library is declared here with the
//! argument count of the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_vertex { pub p: [f32; 4], pub n: [f32; 4], pub uv: [f32; 4], pub t: [f32; 4] }
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_meshlet {
    pub center: [f32; 3], pub radius: f32, pub cone_axis: [i8; 3], pub cone_cutoff: i8,
    pub vertex_offset: u32, pub triangle_offset: u32, pub triangle_count: u8, pub vertex_count: u8, pub _pad: [u8; 2],
}
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_mesh_lod { pub meshlet_offset: u32, pub meshlet_count: u32, pub error: f32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_mesh {
    pub center: [f32; 3], pub radius: f32, pub vertex_offset: u32, pub vertex_count: u32, pub lod_count: u32,
    pub index_offset: u32, pub lods: [rdr_mesh_lod; 8],
}
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_mesh_draw {
    pub alpha_mode: u32, pub meshlet_visibility_offset: u32, pub model_index: u32, pub mesh_index: u16, pub material_index: u16,
}
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_material {
    pub base_color_factor: [f32; 4], pub metallic_roughness_normal_occlusion: [f32; 4], pub emissive_factor_alpha_cutoff: [f32; 4],
    pub specular_glossiness_factor: [f32; 4], pub clearcoat_factor: [f32; 4], pub sheen_color_roughness_factor: [f32; 4],
    pub albedo_texture_transform: [f32; 4], pub normal_texture_transform: [f32; 4], pub metallic_roughness_texture_transform: [f32; 4],
    pub emissive_texture_transform: [f32; 4], pub occlusion_texture_transform: [f32; 4], pub specular_factor: [f32; 4],
    pub albedo_texture: u32, pub albedo_texture_uv: u32, pub normal_texture: u32, pub normal_texture_uv: u32,
    pub metallic_roughness_texture: u32, pub metallic_roughness_texture_uv: u32, pub emissive_texture: u32, pub emissive_texture_uv: u32,
    pub occlusion_texture: u32, pub occlusion_texture_uv: u32, pub clearcoat_texture: u32, pub clearcoat_texture_uv: u32,
    pub specular_color_texture: u32, pub specular_color_texture_uv: u32, pub specular_texture: u32, pub specular_texture_uv: u32,
    pub ior: f32, pub alpha_mode: u32, pub material_type: u32, pub _pad: u32,
}
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_ubo {
    pub view_proj: [f32; 16], pub view: [f32; 16], pub proj: [f32; 16], pub camera_position: [f32; 3], pub _pad0: u32,
    pub near_plane: f32, pub far_plane: f32, pub depth_pyramid_width: f32, pub depth_pyramid_height: f32, pub frustum: [f32; 4],
    pub draw_count: u32, pub lod_enabled: u32, pub cull_enabled: u32, pub lod_target: f32, pub screen_width: f32, pub screen_height: f32,
    pub _pad1: [u32; 2], pub sun_direction: [f32; 4],
}

pub const RDR_FORMAT_RGBA8_UNORM: u32 = 0;
pub const RDR_FORMAT_RGBA8_SRGB: u32 = 1;
pub const RDR_FORMAT_RGBA32F: u32 = 2;
pub const RDR_FORMAT_BC1: u32 = 3;
pub const RDR_FORMAT_BC2: u32 = 4;
pub const RDR_FORMAT_BC3: u32 = 5;
pub const RDR_FORMAT_BC4: u32 = 6;
pub const RDR_FORMAT_BC7: u32 = 7;
pub const RDR_FORMAT_BC7_SRGB: u32 = 8;
pub const RDR_NO_TEXTURE: u32 = 0xFFFF_FFFF;
pub const RDR_VIS_CLEAR: u64 = 0x7FFF_FFFF_FFFF_FFFF;

#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_image_desc { pub data: *const c_void, pub width: u32, pub height: u32, pub mip_count: u32, pub format: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_cubemap_desc { pub data: *const c_void, pub size: u32, pub mip_count: u32, pub format: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_env_desc {
    pub brdf_lut: rdr_image_desc, pub skybox: rdr_cubemap_desc, pub lambertian: rdr_cubemap_desc, pub ggx: rdr_cubemap_desc, pub charlie: rdr_cubemap_desc,
}
#[repr(C)] pub struct rdr_scene_desc {
    pub vertices: *const rdr_vertex, pub indices: *const u32, pub meshlet_vertices: *const u32, pub meshlet_triangles: *const u8,
    pub meshlets: *const rdr_meshlet, pub meshes: *const rdr_mesh, pub mesh_draws: *const rdr_mesh_draw, pub materials: *const rdr_material,
    pub global_transforms: *const f32, pub images: *const rdr_image_desc, pub env: *const rdr_env_desc,
    pub vertex_count: u32, pub index_count: u32, pub meshlet_vertex_count: u32, pub meshlet_triangle_bytes: u32, pub meshlet_count: u32,
    pub mesh_count: u32, pub draw_count: u32, pub material_count: u32, pub transform_count: u32, pub image_count: u32,
    pub meshlet_visibility_count: u32, pub _pad: u32,
}
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct rdr_config { pub lod_enabled: u32, pub cull_enabled: u32, pub collect_stats: u32, pub time_kernels: u32, pub skybox: u32 }

pub const RDR_STAT_KERNELS: usize = 8;
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct rdr_stats {
    pub kernel_ms: [f32; RDR_STAT_KERNELS], pub draws_visible: u64, pub commands: u64, pub meshlets_tested: u64, pub meshlets_visible: u64,
    pub triangles_submitted: u64, pub triangles_tested: u64, pub triangles_rasterised: u64, pub triangles_big: u64, pub fragments: u64,
    pub pixels_covered: u64, pub vertices_transformed: u64, pub frames_timed: u64, pub bin_entries: u64, pub bin_max: u64,
}

pub const RDR_FRAME_DEFERRED_OVERFLOW: u32 = 1;
pub const RDR_FRAME_SETUP_OVERFLOW: u32 = 2;
pub const RDR_FRAME_BIN_OVERFLOW: u32 = 4;
pub const RDR_FRAME_MID_OVERFLOW: u32 = 8;
pub const RDR_FRAME_FRAG_OVERFLOW: u32 = 16;
pub const RDR_FRAME_LIST_OVERFLOW: u32 = 32;
pub const RDR_FRAME_SHADOW_STACK: u32 = 64;
pub const RDR_FRAME_PEER_TIMEOUT: u32 = 128;

pub const RDR_COMPOSITE_MAX_RANKS: usize = 16;
#[repr(C)] #[derive(Clone, Copy)] pub struct rdr_composite_handle { pub vis: [u8; 64], pub color: [u8; 64], pub ctrl: [u8; 64], pub info: [u8; 16] }

#[repr(C)] pub struct rdr_renderer { _private: [u8; 0] }
#[repr(C)] pub struct rdr_mesh_builder { _private: [u8; 0] }
#[repr(C)] pub struct rdr_built_mesh {
    pub vertices: *const rdr_vertex, pub indices: *const u32, pub meshlet_vertices: *const u32, pub meshlet_triangles: *const u8,
    pub meshlets: *const rdr_meshlet, pub vertex_count: u32, pub index_count: u32, pub meshlet_vertex_count: u32,
    pub meshlet_triangle_bytes: u32, pub meshlet_count: u32, pub _pad: u32, pub mesh: rdr_mesh,
}

extern "C" {
    pub fn rdr_decode_bc(device: c_int, format: u32, width: u32, height: u32, mip_count: u32, blocks: *const c_void, block_bytes: usize, out_rgba8: *mut c_void) -> c_int;
    pub fn rdr_decode_bc_noalloc(device: c_int, format: u32, width: u32, height: u32, mip_count: u32, blocks: *const c_void, block_bytes: usize, out_rgba8: *mut c_void) -> c_int;
    pub fn rdr_create(cfg: *const rdr_config, width: u32, height: u32, hip_devices: *const c_int, n_devices: c_int, out: *mut *mut rdr_renderer) -> c_int;
    pub fn rdr_set_scene(r: *mut rdr_renderer, scene: *const rdr_scene_desc) -> c_int;
    pub fn rdr_update_transforms(r: *mut rdr_renderer, mat4s: *const f32, count: u32) -> c_int;
    pub fn rdr_render(r: *mut rdr_renderer, cam_pos: *const f32, cam_dir: *const f32, sun_dir: *const f32) -> c_int;
    pub fn rdr_frame_status(r: *mut rdr_renderer, flags: *mut u32) -> c_int;
    pub fn rdr_render_visbuffer(r: *mut rdr_renderer, cam_pos: *const f32, cam_dir: *const f32, sun_dir: *const f32) -> c_int;
    pub fn rdr_shade(r: *mut rdr_renderer) -> c_int;
    pub fn rdr_set_partition(r: *mut rdr_renderer, instance_lo: u32, instance_hi: u32) -> c_int;
    pub fn rdr_set_shade_rows(r: *mut rdr_renderer, row_lo: u32, row_hi: u32) -> c_int;
    pub fn rdr_set_partition_interleaved(r: *mut rdr_renderer, chunk: u32, rank: u32, world: u32) -> c_int;
    pub fn rdr_set_shade_bands(r: *mut rdr_renderer, band: u32, rank: u32, world: u32) -> c_int;
    pub fn rdr_set_partition_replication(r: *mut rdr_renderer, min_triangle_pixels: f32) -> c_int;
    pub fn rdr_visbuffer_device_ptr(r: *mut rdr_renderer) -> *mut c_void;
    pub fn rdr_color_device_ptr(r: *mut rdr_renderer) -> *mut c_void;
    pub fn rdr_composite_export(r: *mut rdr_renderer, out: *mut rdr_composite_handle) -> c_int;
    pub fn rdr_composite_connect(r: *mut rdr_renderer, rank: u32, world: u32, handles: *const rdr_composite_handle) -> c_int;
    pub fn rdr_composite_exchange(r: *mut rdr_renderer) -> c_int;
    pub fn rdr_composite_gather(r: *mut rdr_renderer) -> c_int;
    pub fn rdr_composite_disconnect(r: *mut rdr_renderer) -> c_int;
    pub fn rdr_set_stream(r: *mut rdr_renderer, hip_stream: *mut c_void) -> c_int;
    pub fn rdr_sync(r: *mut rdr_renderer) -> c_int;
    pub fn rdr_read_visbuffer(r: *mut rdr_renderer, out_xy: *mut u32) -> c_int;
    pub fn rdr_read_visbuffer_u64(r: *mut rdr_renderer, out: *mut u64) -> c_int;
    pub fn rdr_read_color(r: *mut rdr_renderer, out_rgba8_srgb: *mut u8) -> c_int;
    pub fn rdr_get_stats(r: *mut rdr_renderer, out: *mut rdr_stats) -> c_int;
    pub fn rdr_build_ubo(width: u32, height: u32, cam_pos: *const f32, cam_dir: *const f32, sun_dir: *const f32, draw_count: u32, lod_enabled: u32, cull_enabled: u32, out: *mut rdr_ubo) -> c_int;
    pub fn rdr_get_ubo(r: *mut rdr_renderer, out: *mut rdr_ubo) -> c_int;
    pub fn rdr_destroy(r: *mut rdr_renderer);
    pub fn rdr_last_error() -> *const c_char;
    pub fn rdr_bake_brdf_lut(hip_device: c_int, size: u32, samples: u32, out_rgba32f: *mut f32) -> c_int;
    pub fn rdr_bake_env_map(hip_device: c_int, skybox: *const rdr_cubemap_desc, distribution: u32, size: u32, mips: u32, samples: u32, out_rgba32f: *mut f32) -> c_int;
    pub fn rdr_mesh_build(vertices: *const rdr_vertex, vertex_count: u32, indices: *const u32, index_count: u32, max_lods: u32, fast_mode: u32, out: *mut *mut rdr_mesh_builder) -> c_int;
    pub fn rdr_mesh_result(b: *const rdr_mesh_builder, out: *mut rdr_built_mesh) -> c_int;
    pub fn rdr_mesh_free(b: *mut rdr_mesh_builder);
    pub fn rdr_generate_tangents(vertices: *mut rdr_vertex, vertex_count: u32, indices: *const u32, index_count: u32) -> c_int;
}

const _: () = {
    assert!(std::mem::size_of::<rdr_vertex>() == 64);
    assert!(std::mem::size_of::<rdr_meshlet>() == 32);
    assert!(std::mem::size_of::<rdr_mesh>() == 160);
    assert!(std::mem::size_of::<rdr_mesh_draw>() == 16);
    assert!(std::mem::size_of::<rdr_material>() == 272);
    assert!(std::mem::size_of::<rdr_ubo>() == 288);
    assert!(std::mem::size_of::<rdr_composite_handle>() == 208);
};
