//! Raw bindings of `librt_hip.so` (the C ABI declared in include/rt_api.h and include/rt_types.h), one item per C item,
//! same names, same order of fields and arguments. tests/test_rust_binding.py parses this file and the two headers and fails
//! on any difference in name, arity, pointer constness or scalar width; the file has not been compiled (no Rust toolchain
//! in the build image).
//!
//! The wire records (`RtBvhNode`, `RtTriangle`, `RtMaterial`, `RtLight`, `RtEmissiveTri`, `RtCamera`) are byte-identical to
//! the application's own `repr(C)` types (src/types/*.rs), so a pointer to a `crate::CameraBufferObject` may be passed
//! where `*const RtCamera` is expected, and `Scene::write_buffers` output where a scene blob is expected.
#![allow(dead_code)]
use std::os::raw::{c_char, c_void};

pub const RT_OK: i32 = 0;
pub const RT_ERR_INVALID: i32 = -1;
pub const RT_ERR_DEVICE: i32 = -2;
pub const RT_ERR_NO_SCENE: i32 = -3;
pub const RT_ERR_UNSUPPORTED: i32 = -4;
pub const RT_ERR_OOM: i32 = -5;

pub const RT_REF_RGBA8_ACCUM: u32 = 1 << 0;
pub const RT_REF_CDF_PAD: u32 = 1 << 1;
pub const RT_FIX_ABSORPTION: u32 = 1 << 2;
pub const RT_COUNT_TRAVERSAL: u32 = 1 << 3;
pub const RT_KERNEL_TIMING: u32 = 1 << 4;
pub const RT_SINGLE_STREAM: u32 = 1 << 5;
pub const RT_EXACT_ORDER: u32 = 1 << 6;
pub const RT_EXACT_ARITH: u32 = 1 << 7;

pub const RT_NO_TEXTURE: u32 = 0xFFFF_FFFF;
pub const RT_STATIC_MAX_RAYS: u32 = 1 << 18;

/// Opaque renderer context (`rt_ctx`).
#[repr(C)]
pub struct RtCtx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtBvhNode {
    pub min: [f32; 3],
    pub idx: u32,
    pub max: [f32; 3],
    pub amt: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtTriangle {
    pub material_idx: u32,
    pub is_sphere: u32,
    pub _pad0: [u32; 2],
    pub v: [[f32; 4]; 3],
    pub n: [[f32; 4]; 3],
    pub uv: [[f32; 2]; 3],
    pub _pad1: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtMaterial {
    pub base_colour: [f32; 4],
    pub emission: [f32; 4],
    pub metallic: f32,
    pub roughness: f32,
    pub subsurface: f32,
    pub anisotropic: f32,
    pub specular_tint: f32,
    pub sheen: f32,
    pub sheen_tint: f32,
    pub clearcoat: f32,
    pub clearcoat_roughness: f32,
    pub spec_trans: f32,
    pub ior: f32,
    pub shade_smooth: u32,
    pub base_color_tex: u32,
    pub metallic_roughness_tex: u32,
    pub normal_tex: u32,
    pub emission_tex: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtLight {
    pub position: [f32; 4],
    pub emission: [f32; 4],
    pub type_: u32,
    pub _pad: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtEmissiveTri {
    pub tri_index: u32,
    pub mesh_index: u32,
    pub area: f32,
    pub power: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCamera {
    pub resolution: [u32; 2],
    pub view_port_uv: [f32; 2],
    pub focal_length: f32,
    pub focus_distance: f32,
    pub aperture_radius: f32,
    pub frame: u32,
    pub location: [f32; 4],
    pub rotation: [f32; 16],
    pub _pad: [u32; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtTexture {
    pub width: u32,
    pub height: u32,
    pub pixels: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtConfig {
    pub device: i32,
    pub width: u32,
    pub height: u32,
    pub max_bounces: u32,
    pub flags: u32,
    pub frames_per_pass: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtTimings {
    pub total_ms: f32,
    pub raygen_ms: f32,
    pub extend_ms: f32,
    pub shade_ms: f32,
    pub connect_ms: f32,
    pub accumulate_ms: f32,
    pub raygen_launches: u32,
    pub extend_launches: u32,
    pub shade_launches: u32,
    pub connect_launches: u32,
    pub accumulate_launches: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCounters {
    pub camera_rays: u64,
    pub extend_rays: u64,
    pub shadow_rays: u64,
    pub extend_node_pops: u64,
    pub extend_tri_tests: u64,
    pub shadow_node_pops: u64,
    pub shadow_tri_tests: u64,
    pub extend_leaf_pops: u64,
    pub shadow_leaf_pops: u64,
    pub tail_paths: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtTuneReport {
    pub sampled_visits: [u64; 3],
    pub resident_records: [u32; 3],
    pub resident_share_before: [f32; 3],
    pub resident_share_after: [f32; 3],
    pub records_moved: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtHit {
    pub t: f32,
    pub u: f32,
    pub v: f32,
    pub prim: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtShadeVertex {
    pub ray_o: [f32; 3],
    pub prev_pdf: f32,
    pub ray_d: [f32; 3],
    pub seed: u32,
    pub throughput: [f32; 3],
    pub reserved: u32,
    pub hit: RtHit,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtShadeResult {
    pub emitted: [f32; 3],
    pub seed: u32,
    pub throughput: [f32; 3],
    pub pdf: f32,
    pub next_o: [f32; 3],
    pub alive: u32,
    pub next_d: [f32; 3],
    pub has_shadow: u32,
    pub shadow_o: [f32; 3],
    pub shadow_tmax: f32,
    pub shadow_d: [f32; 3],
    pub light_index: i32,
    pub shadow_contrib: [f32; 3],
    pub lobe: i32,
    pub light_tri: i32,
    pub light_pdf: f32,
    pub bsdf_pdf_light: f32,
    pub roughness: f32,
    pub medium_throughput: [f32; 3],
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtBvhParams {
    pub traversal_cost: f32,
    pub intersect_cost: f32,
    pub min_leaf: u32,
    pub max_leaf: u32,
    pub threads: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtBvhStats {
    pub sah_cost: f32,
    pub node_count: u32,
    pub leaf_count: u32,
    pub max_depth: u32,
    pub max_leaf_size: u32,
    pub avg_leaf_size: f32,
    pub build_ms: f32,
}

#[link(name = "rt_hip")]
extern "C" {
    pub fn rt_version() -> *const c_char;
    pub fn rt_last_error(ctx: *const RtCtx) -> *const c_char;
    pub fn rt_bvh_build(
        tris: *mut RtTriangle,
        n_tri: u32,
        params: *const RtBvhParams,
        nodes: *mut RtBvhNode,
        n_nodes: *mut u32,
        stats: *mut RtBvhStats,
    ) -> i32;
    pub fn rt_bvh_build_sweep(
        tris: *mut RtTriangle,
        n_tri: u32,
        nodes: *mut RtBvhNode,
        n_nodes: *mut u32,
        stats: *mut RtBvhStats,
    ) -> i32;
    pub fn rt_bvh_build_spatial(
        tris: *const RtTriangle,
        n_tri: u32,
        params: *const RtBvhParams,
        alpha: f32,
        budget_fraction: f32,
        tris_out: *mut RtTriangle,
        tri_capacity: u32,
        n_tris_out: *mut u32,
        nodes: *mut RtBvhNode,
        node_capacity: u32,
        n_nodes: *mut u32,
        stats: *mut RtBvhStats,
    ) -> i32;
    pub fn rt_build_lights(
        tris: *const RtTriangle,
        n_tri: u32,
        mats: *const RtMaterial,
        n_mat: u32,
        emissive: *mut RtEmissiveTri,
        cdf: *mut f32,
        n_emissive: *mut u32,
        total_power: *mut f32,
    ) -> i32;
    pub fn rt_scene_buffer_sizes(n_bvh: u32, n_mat: u32, n_tri: u32, n_light: u32, n_emissive: u32, sizes: *mut u64) -> i32;
    pub fn rt_scene_write_buffers(
        nodes: *const RtBvhNode,
        n_bvh: u32,
        mats: *const RtMaterial,
        n_mat: u32,
        tris: *const RtTriangle,
        n_tri: u32,
        lights: *const RtLight,
        n_light: u32,
        emissive: *const RtEmissiveTri,
        cdf: *const f32,
        n_emissive: u32,
        blob: *mut c_void,
        blob_capacity: u64,
    ) -> i32;
    pub fn rt_camera_look_at(
        cam: *mut RtCamera,
        width: u32,
        height: u32,
        location: *const f32,
        look_at: *const f32,
        focal_length: f32,
        focus_distance: f32,
        aperture_radius: f32,
    ) -> i32;
    pub fn rt_create(cfg: *const RtConfig, out: *mut *mut RtCtx) -> i32;
    pub fn rt_destroy(ctx: *mut RtCtx);
    pub fn rt_upload_scene(
        ctx: *mut RtCtx,
        blob: *const c_void,
        sizes: *const u64,
        n_bvh: u32,
        n_mat: u32,
        n_tri: u32,
        n_light: u32,
        n_emissive: u32,
    ) -> i32;
    pub fn rt_upload_textures(ctx: *mut RtCtx, textures: *const RtTexture, n_textures: u32, skybox: *const RtTexture) -> i32;
    pub fn rt_upload_skybox_f32(ctx: *mut RtCtx, width: u32, height: u32, rgba: *const f32) -> i32;
    pub fn rt_tune_resident_set(ctx: *mut RtCtx, report: *mut RtTuneReport) -> i32;
    pub fn rt_set_camera(ctx: *mut RtCtx, cam: *const RtCamera) -> i32;
    pub fn rt_resize(ctx: *mut RtCtx, width: u32, height: u32) -> i32;
    pub fn rt_reset_accumulation(ctx: *mut RtCtx) -> i32;
    pub fn rt_set_frame_slice(ctx: *mut RtCtx, first: u32, stride: u32) -> i32;
    pub fn rt_set_tile_slice(ctx: *mut RtCtx, first: u32, stride: u32) -> i32;
    pub fn rt_render(ctx: *mut RtCtx, n_samples: u32) -> i32;
    pub fn rt_sync(ctx: *mut RtCtx) -> i32;
    pub fn rt_poll(ctx: *mut RtCtx, done: *mut i32) -> i32;
    pub fn rt_sample_count(ctx: *const RtCtx) -> u32;
    pub fn rt_read_accum_f32(ctx: *mut RtCtx, rgba: *mut f32, n_floats: usize) -> i32;
    pub fn rt_read_sum_f32(ctx: *mut RtCtx, rgba: *mut f32, n_floats: usize) -> i32;
    pub fn rt_add_sum_f32(ctx: *mut RtCtx, rgba_sum: *const f32, n_floats: usize) -> i32;
    pub fn rt_accum_device_buffer(ctx: *mut RtCtx, device_ptr: *mut *mut c_void, n_bytes: *mut usize) -> i32;
    pub fn rt_read_framebuffer_rgba8(ctx: *mut RtCtx, rgba: *mut u8, n_bytes: usize) -> i32;
    pub fn rt_last_timings(ctx: *const RtCtx, out: *mut RtTimings) -> i32;
    pub fn rt_get_counters(ctx: *mut RtCtx, out: *mut RtCounters) -> i32;
    pub fn rt_reset_counters(ctx: *mut RtCtx) -> i32;
    pub fn rt_trace_closest(ctx: *mut RtCtx, n: u32, origins: *const f32, directions: *const f32, hits: *mut RtHit) -> i32;
    pub fn rt_trace_shadow(
        ctx: *mut RtCtx,
        n: u32,
        origins: *const f32,
        directions: *const f32,
        tmax: *const f32,
        visible: *mut u8,
    ) -> i32;
    pub fn rt_bench_trace(
        ctx: *mut RtCtx,
        n: u32,
        origins: *const f32,
        directions: *const f32,
        shadow: i32,
        repeat: u32,
        ms_per_launch: *mut f32,
    ) -> i32;
    pub fn rt_trace_closest_static(
        ctx: *mut RtCtx,
        n: u32,
        origins: *const f32,
        directions: *const f32,
        hits: *mut RtHit,
    ) -> i32;
    pub fn rt_generate_rays(ctx: *mut RtCtx, frame: u32, origins: *mut f32, directions: *mut f32, seeds: *mut u32) -> i32;
    pub fn rt_eval_bsdf(
        ctx: *mut RtCtx,
        n: u32,
        mats: *const RtMaterial,
        eta: *const f32,
        V: *const f32,
        N: *const f32,
        L: *const f32,
        out_f: *mut f32,
        out_pdf: *mut f32,
    ) -> i32;
    pub fn rt_sample_bsdf(
        ctx: *mut RtCtx,
        n: u32,
        mats: *const RtMaterial,
        eta: *const f32,
        V: *const f32,
        N: *const f32,
        r1r2: *const f32,
        out_L: *mut f32,
        out_f: *mut f32,
        out_pdf: *mut f32,
    ) -> i32;
    pub fn rt_sample_texture(ctx: *mut RtCtx, tex: u32, n: u32, uvw: *const f32, rgba: *mut f32) -> i32;
    pub fn rt_shade_vertices(
        ctx: *mut RtCtx,
        n: u32,
        bounce: u32,
        vertices: *const RtShadeVertex,
        results: *mut RtShadeResult,
    ) -> i32;
    pub fn rt_render_radiance(ctx: *mut RtCtx, frame: u32, rgb: *mut f32) -> i32;
}
