//! `include/rttnw_hip.h` in Rust: every struct `#[repr(C)]` with the header's field order, every entry point.
//! Reference items each call replaces are cited in the header (file:line of luliic2/rttnw).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RTTNW_ABI_VERSION: c_int = 3;

/// Opaque scene handle.
#[repr(C)]
pub struct rttnw_scene {
    _private: [u8; 0],
}
pub type rttnw_id = i32;

// enum rttnw_status
pub const RTTNW_OK: c_int = 0;
pub const RTTNW_ERR_INVALID: c_int = -1;
pub const RTTNW_ERR_STATE: c_int = -2;
pub const RTTNW_ERR_UNSUPPORTED: c_int = -3;
pub const RTTNW_ERR_HIP: c_int = -4;
pub const RTTNW_ERR_NOMEM: c_int = -5;
// enum rttnw_plane
pub const RTTNW_XY: c_int = 0;
pub const RTTNW_XZ: c_int = 1;
pub const RTTNW_YZ: c_int = 2;
// enum rttnw_precision
pub const RTTNW_F64: u32 = 0;
pub const RTTNW_F32: u32 = 1;
pub const RTTNW_F64_STRICT: u32 = 2;
pub const RTTNW_QUIRK_YROTATE_BACKROT: u32 = 1;
pub const RTTNW_QUIRKS_REFERENCE: u32 = RTTNW_QUIRK_YROTATE_BACKROT;
pub const RTTNW_BVH_HOST_SAH: u32 = 0;
pub const RTTNW_BVH_DEVICE_LBVH: u32 = 1;
pub const RTTNW_BVH_DEVICE_SAH: u32 = 2;
pub const RTTNW_BVH_AUTO: u32 = 3;
pub const RTTNW_BVH_AUTO_DEVICE_LEAVES: u32 = 100000;

/// `CameraDescriptor` — src/math/camera.rs:5-15
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rttnw_camera_desc {
    pub lookfrom: [f64; 3],
    pub lookat: [f64; 3],
    pub view_up: [f64; 3],
    pub vertical_fov: f64,
    pub aspect_ratio: f64,
    pub aperture: f64,
    pub focus_distance: f64,
    pub open_time: f64,
    pub close_time: f64,
}

/// What `render()` hard-codes or takes as arguments — src/main.rs:58,184-197,216,33
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rttnw_params {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub max_depth: u32,
    pub t_min: f64,
    pub background: [f64; 3],
    pub seed: u64,
    pub precision: u32,
    pub quirks: u32,
    pub spp_chunk: u32,
    pub tile_rank: u32,
    pub tile_world: u32,
    pub collect_counters: u32,
    pub sample_begin: u32,
    pub reserved0: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rttnw_stats {
    pub samples: u64,
    pub rays: u64,
    pub nodes_visited: u64,
    pub prims_tested: u64,
    pub texel_fetches: u64,
    pub kernel_ms: f64,
    pub n_nodes: u32,
    pub n_prims: u32,
    pub scene_bytes: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rttnw_tile_layout {
    pub tiles_x: u32,
    pub tiles_y: u32,
    pub n_tiles: u32,
    pub tiles_per_rank: u32,
    pub pixels_per_rank: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rttnw_build_info {
    pub builder: u32,
    pub n_nodes: u32,
    pub n_prims: u32,
    pub stack_depth: u32,
    pub lower_ms: f64,
    pub device_ms: f64,
}

/// Scene-building entry points as a table (`rttnw_builder()`), in the header's order.
#[repr(C)]
pub struct rttnw_builder_api {
    pub scene_create: Option<unsafe extern "C" fn(u64, *mut *mut rttnw_scene) -> c_int>,
    pub scene_destroy: Option<unsafe extern "C" fn(*mut rttnw_scene)>,
    pub tex_solid: Option<unsafe extern "C" fn(*mut rttnw_scene, f64, f64, f64) -> rttnw_id>,
    pub tex_checker: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id, rttnw_id) -> rttnw_id>,
    pub tex_noise: Option<unsafe extern "C" fn(*mut rttnw_scene, f64) -> rttnw_id>,
    pub tex_image_rgba8: Option<unsafe extern "C" fn(*mut rttnw_scene, *const u8, u32, u32) -> rttnw_id>,
    pub mat_lambertian: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id) -> rttnw_id>,
    pub mat_metal: Option<unsafe extern "C" fn(*mut rttnw_scene, f64, f64, f64, f64) -> rttnw_id>,
    pub mat_dielectric: Option<unsafe extern "C" fn(*mut rttnw_scene, f64) -> rttnw_id>,
    pub mat_diffuse_light: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id) -> rttnw_id>,
    pub mat_isotropic: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id) -> rttnw_id>,
    pub sphere: Option<unsafe extern "C" fn(*mut rttnw_scene, *const f64, f64, rttnw_id) -> rttnw_id>,
    pub moving_sphere: Option<unsafe extern "C" fn(*mut rttnw_scene, *const f64, *const f64, f64, f64, f64, rttnw_id) -> rttnw_id>,
    pub rectangle: Option<unsafe extern "C" fn(*mut rttnw_scene, c_int, f64, f64, f64, f64, f64, rttnw_id) -> rttnw_id>,
    pub cube: Option<unsafe extern "C" fn(*mut rttnw_scene, *const f64, *const f64, rttnw_id) -> rttnw_id>,
    pub list: Option<unsafe extern "C" fn(*mut rttnw_scene) -> rttnw_id>,
    pub list_push: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id, rttnw_id) -> c_int>,
    pub bvh_tree: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id) -> rttnw_id>,
    pub translate: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id, *const f64) -> rttnw_id>,
    pub rotate_y: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id, f64) -> rttnw_id>,
    pub constant_medium: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id, f64, rttnw_id) -> rttnw_id>,
    pub scene_set_world: Option<unsafe extern "C" fn(*mut rttnw_scene, rttnw_id) -> c_int>,
    pub scene_commit: Option<unsafe extern "C" fn(*mut rttnw_scene) -> c_int>,
    pub last_error: Option<unsafe extern "C" fn() -> *const c_char>,
}

extern "C" {
    // ---- lifecycle
    pub fn rttnw_scene_create(scene_seed: u64, out: *mut *mut rttnw_scene) -> c_int;
    pub fn rttnw_scene_destroy(scene: *mut rttnw_scene);
    // ---- textures (texture.rs)
    pub fn rttnw_tex_solid(s: *mut rttnw_scene, r: f64, g: f64, b: f64) -> rttnw_id;
    pub fn rttnw_tex_checker(s: *mut rttnw_scene, odd: rttnw_id, even: rttnw_id) -> rttnw_id;
    pub fn rttnw_tex_noise(s: *mut rttnw_scene, scale: f64) -> rttnw_id;
    pub fn rttnw_tex_image_rgba8(s: *mut rttnw_scene, rgba: *const u8, w: u32, h: u32) -> rttnw_id;
    // ---- materials (material.rs)
    pub fn rttnw_mat_lambertian(s: *mut rttnw_scene, tex: rttnw_id) -> rttnw_id;
    pub fn rttnw_mat_metal(s: *mut rttnw_scene, r: f64, g: f64, b: f64, fuzz: f64) -> rttnw_id;
    pub fn rttnw_mat_dielectric(s: *mut rttnw_scene, refraction_index: f64) -> rttnw_id;
    pub fn rttnw_mat_diffuse_light(s: *mut rttnw_scene, tex: rttnw_id) -> rttnw_id;
    pub fn rttnw_mat_isotropic(s: *mut rttnw_scene, tex: rttnw_id) -> rttnw_id;
    // ---- hittables (hittable.rs)
    pub fn rttnw_sphere(s: *mut rttnw_scene, center: *const f64, radius: f64, mat: rttnw_id) -> rttnw_id;
    pub fn rttnw_moving_sphere(s: *mut rttnw_scene, center0: *const f64, center1: *const f64, time0: f64, time1: f64, radius: f64, mat: rttnw_id) -> rttnw_id;
    pub fn rttnw_rectangle(s: *mut rttnw_scene, plane: c_int, a0: f64, a1: f64, b0: f64, b1: f64, k: f64, mat: rttnw_id) -> rttnw_id;
    pub fn rttnw_cube(s: *mut rttnw_scene, box_min: *const f64, box_max: *const f64, mat: rttnw_id) -> rttnw_id;
    pub fn rttnw_list(s: *mut rttnw_scene) -> rttnw_id;
    pub fn rttnw_list_push(s: *mut rttnw_scene, list: rttnw_id, item: rttnw_id) -> c_int;
    pub fn rttnw_bvh_tree(s: *mut rttnw_scene, list: rttnw_id) -> rttnw_id;
    pub fn rttnw_translate(s: *mut rttnw_scene, item: rttnw_id, offset: *const f64) -> rttnw_id;
    pub fn rttnw_rotate_y(s: *mut rttnw_scene, item: rttnw_id, angle_degrees: f64) -> rttnw_id;
    pub fn rttnw_constant_medium(s: *mut rttnw_scene, boundary: rttnw_id, density: f64, tex: rttnw_id) -> rttnw_id;
    pub fn rttnw_hittable_bounds(s: *const rttnw_scene, hittable: rttnw_id, initial_time: f64, final_time: f64, out_min_max: *mut f64) -> c_int;
    pub fn rttnw_scene_set_world(s: *mut rttnw_scene, world_list: rttnw_id) -> c_int;
    pub fn rttnw_scene_set_bvh_builder(s: *mut rttnw_scene, builder: u32) -> c_int;
    pub fn rttnw_scene_commit(s: *mut rttnw_scene) -> c_int;
    // ---- render (main.rs, camera.rs)
    pub fn rttnw_tile_layout_get(width: u32, height: u32, world: u32, out: *mut rttnw_tile_layout) -> c_int;
    pub fn rttnw_render(s: *mut rttnw_scene, cam: *const rttnw_camera_desc, p: *const rttnw_params, out_linear_rgb: *mut f64, out_rgba8: *mut u8, stats: *mut rttnw_stats) -> c_int;
    pub fn rttnw_render_multi(s: *mut rttnw_scene, cam: *const rttnw_camera_desc, p: *const rttnw_params, ngpu: u32, device_ids: *const i32, out_linear_rgb: *mut f64, out_rgba8: *mut u8, stats: *mut rttnw_stats) -> c_int;
    pub fn rttnw_render_tiles_device(s: *mut rttnw_scene, cam: *const rttnw_camera_desc, p: *const rttnw_params, d_packed: *mut c_void, hip_stream: *mut c_void, stats: *mut rttnw_stats) -> c_int;
    pub fn rttnw_untile_device(width: u32, height: u32, world: u32, precision: u32, d_gathered: *const c_void, d_linear_rgb: *mut c_void, d_rgba8: *mut u8, hip_stream: *mut c_void) -> c_int;
    // ---- introspection
    pub fn rttnw_abi_version() -> c_int;
    pub fn rttnw_shutdown();
    pub fn rttnw_device_count() -> c_int;
    pub fn rttnw_last_error() -> *const c_char;
    pub fn rttnw_scene_info(s: *mut rttnw_scene, out: *mut rttnw_stats) -> c_int;
    pub fn rttnw_scene_build_info(s: *const rttnw_scene, out: *mut rttnw_build_info) -> c_int;
    pub fn rttnw_debug_scene_nodes(s: *const rttnw_scene, out_nodes: *mut c_void, max_nodes: u32, top_root: *mut i32) -> c_int;
    pub fn rttnw_debug_scene_nodes4(s: *const rttnw_scene, out_nodes: *mut c_void, max_nodes: u32, top_root: *mut i32) -> c_int;
    pub fn rttnw_debug_probe_path(s: *mut rttnw_scene, cam: *const rttnw_camera_desc, p: *const rttnw_params, px: u32, row: u32, sample: u32, out: *mut f64, max_out: u32) -> c_int;
    pub fn rttnw_builder() -> *const rttnw_builder_api;
}
