//! The vocabulary of the reference's `src/scenes.rs` over handles: a `SceneBuilder` whose methods are named after the
//! constructors they replace, and `render()` (src/main.rs:58-233) as one call.  Every method is one FFI call.
use crate::ffi;
use std::ffi::CStr;
use std::ptr;

#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
pub type Result<T> = std::result::Result<T, Error>;

fn last_error(code: i32) -> Error {
    let message = unsafe {
        let p = ffi::rttnw_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    Error { code, message }
}
fn id(rc: i32) -> Result<i32> {
    if rc < 0 { Err(last_error(rc)) } else { Ok(rc) }
}
fn ok(rc: i32) -> Result<()> {
    if rc < 0 { Err(last_error(rc)) } else { Ok(()) }
}

/// `Arc<dyn Texture>` — texture.rs
#[derive(Clone, Copy, Debug)]
pub struct Tex(pub i32);
/// `Arc<dyn Material>` — material.rs
#[derive(Clone, Copy, Debug)]
pub struct Mat(pub i32);
/// `Box<dyn Hittable>` — hittable.rs
#[derive(Clone, Copy, Debug)]
pub struct Hit(pub i32);

#[derive(Clone, Copy, Debug)]
pub enum Plane {
    XY = 0,
    XZ = 1,
    YZ = 2,
}

pub struct SceneBuilder {
    raw: *mut ffi::rttnw_scene,
}

impl SceneBuilder {
    /// `scene_seed` feeds the Perlin tables the reference draws from `thread_rng()` (noise.rs:15-29,40-47).
    pub fn new(scene_seed: u64) -> Result<Self> {
        let mut raw = ptr::null_mut();
        ok(unsafe { ffi::rttnw_scene_create(scene_seed, &mut raw) })?;
        Ok(SceneBuilder { raw })
    }
    // ---- textures
    pub fn solid(&mut self, rgb: [f64; 3]) -> Result<Tex> { id(unsafe { ffi::rttnw_tex_solid(self.raw, rgb[0], rgb[1], rgb[2]) }).map(Tex) }
    pub fn checker(&mut self, odd: Tex, even: Tex) -> Result<Tex> { id(unsafe { ffi::rttnw_tex_checker(self.raw, odd.0, even.0) }).map(Tex) }
    pub fn noise(&mut self, scale: f64) -> Result<Tex> { id(unsafe { ffi::rttnw_tex_noise(self.raw, scale) }).map(Tex) }
    /// `ImageTexture::new(path)`: the host decodes (`image::open(path)?.to_rgba8()`); `None` = load failure -> cyan (texture.rs:102-105).
    pub fn image(&mut self, rgba8: Option<(&[u8], u32, u32)>) -> Result<Tex> {
        let rc = match rgba8 {
            Some((px, w, h)) => {
                assert!(px.len() as u64 == w as u64 * h as u64 * 4);
                unsafe { ffi::rttnw_tex_image_rgba8(self.raw, px.as_ptr(), w, h) }
            }
            None => unsafe { ffi::rttnw_tex_image_rgba8(self.raw, ptr::null(), 0, 0) },
        };
        id(rc).map(Tex)
    }
    // ---- materials
    pub fn lambertian(&mut self, t: Tex) -> Result<Mat> { id(unsafe { ffi::rttnw_mat_lambertian(self.raw, t.0) }).map(Mat) }
    pub fn metal(&mut self, albedo: [f64; 3], fuzz: f64) -> Result<Mat> { id(unsafe { ffi::rttnw_mat_metal(self.raw, albedo[0], albedo[1], albedo[2], fuzz) }).map(Mat) }
    pub fn dielectric(&mut self, refraction_index: f64) -> Result<Mat> { id(unsafe { ffi::rttnw_mat_dielectric(self.raw, refraction_index) }).map(Mat) }
    pub fn diffuse_light(&mut self, t: Tex) -> Result<Mat> { id(unsafe { ffi::rttnw_mat_diffuse_light(self.raw, t.0) }).map(Mat) }
    pub fn isotropic(&mut self, t: Tex) -> Result<Mat> { id(unsafe { ffi::rttnw_mat_isotropic(self.raw, t.0) }).map(Mat) }
    // ---- hittables
    pub fn sphere(&mut self, center: [f64; 3], radius: f64, m: Mat) -> Result<Hit> { id(unsafe { ffi::rttnw_sphere(self.raw, center.as_ptr(), radius, m.0) }).map(Hit) }
    pub fn moving_sphere(&mut self, center: std::ops::Range<[f64; 3]>, time: std::ops::Range<f64>, radius: f64, m: Mat) -> Result<Hit> {
        id(unsafe { ffi::rttnw_moving_sphere(self.raw, center.start.as_ptr(), center.end.as_ptr(), time.start, time.end, radius, m.0) }).map(Hit)
    }
    /// `XY|XZ|YZ::rectangle(material, a0..a1, b0..b1, k)`
    pub fn rectangle(&mut self, plane: Plane, m: Mat, a: std::ops::Range<f64>, b: std::ops::Range<f64>, k: f64) -> Result<Hit> {
        id(unsafe { ffi::rttnw_rectangle(self.raw, plane as i32, a.start, a.end, b.start, b.end, k, m.0) }).map(Hit)
    }
    pub fn cube(&mut self, min: [f64; 3], max: [f64; 3], m: Mat) -> Result<Hit> { id(unsafe { ffi::rttnw_cube(self.raw, min.as_ptr(), max.as_ptr(), m.0) }).map(Hit) }
    pub fn list(&mut self) -> Result<Hit> { id(unsafe { ffi::rttnw_list(self.raw) }).map(Hit) }
    pub fn push(&mut self, list: Hit, item: Hit) -> Result<()> { ok(unsafe { ffi::rttnw_list_push(self.raw, list.0, item.0) }) }
    /// `BvhTree::from(list)` (the list is consumed, as in the reference)
    pub fn bvh_tree(&mut self, list: Hit) -> Result<Hit> { id(unsafe { ffi::rttnw_bvh_tree(self.raw, list.0) }).map(Hit) }
    pub fn translate(&mut self, item: Hit, offset: [f64; 3]) -> Result<Hit> { id(unsafe { ffi::rttnw_translate(self.raw, item.0, offset.as_ptr()) }).map(Hit) }
    pub fn rotate_y(&mut self, item: Hit, angle_degrees: f64) -> Result<Hit> { id(unsafe { ffi::rttnw_rotate_y(self.raw, item.0, angle_degrees) }).map(Hit) }
    pub fn constant_medium(&mut self, boundary: Hit, density: f64, phase: Tex) -> Result<Hit> { id(unsafe { ffi::rttnw_constant_medium(self.raw, boundary.0, density, phase.0) }).map(Hit) }
    /// `Hittable::bounding_box(initial_time, final_time) -> Option<Bound>` (hittable.rs:50) of any hittable built so far: `Some((min, max))`, or `None`
    /// where the reference returns `None` (an empty `List`).  `YRotate`'s box is the correct rotation of the item's corners (not quirk Q2's).
    pub fn bounding_box(&self, item: Hit, time: std::ops::Range<f64>) -> Result<Option<([f64; 3], [f64; 3])>> {
        let mut b = [0f64; 6];
        let rc = unsafe { ffi::rttnw_hittable_bounds(self.raw, item.0, time.start, time.end, b.as_mut_ptr()) };
        if rc < 0 { return ok(rc).map(|_| None); }
        Ok(if rc == 1 { Some(([b[0], b[1], b[2]], [b[3], b[4], b[5]])) } else { None })
    }
    /// Force where the BVHs are built; before `commit`.  Without this call the library decides per tree (`RTTNW_BVH_AUTO`: host below
    /// 100 000 leaves, the device's binned-SAH build above).
    pub fn device_bvh(&mut self, on: bool) -> Result<()> {
        ok(unsafe { ffi::rttnw_scene_set_bvh_builder(self.raw, if on { ffi::RTTNW_BVH_DEVICE_SAH } else { ffi::RTTNW_BVH_HOST_SAH }) })
    }
    /// `Scene { world, .. }` is complete: lower, build, upload.
    pub fn commit(mut self, world: Hit) -> Result<Scene> {
        ok(unsafe { ffi::rttnw_scene_set_world(self.raw, world.0) })?;
        ok(unsafe { ffi::rttnw_scene_commit(self.raw) })?;
        let raw = std::mem::replace(&mut self.raw, ptr::null_mut());
        Ok(Scene { raw })
    }
}
impl Drop for SceneBuilder {
    fn drop(&mut self) {
        if !self.raw.is_null() { unsafe { ffi::rttnw_scene_destroy(self.raw) } }
    }
}

/// A committed, immutable scene resident on the device(s).
pub struct Scene {
    raw: *mut ffi::rttnw_scene,
}
unsafe impl Send for Scene {}

/// The constants of `render()` (main.rs:184-197,216,33) with the reference's values.
pub fn reference_params(width: u32, height: u32, samples: u32, background: [f64; 3]) -> ffi::rttnw_params {
    ffi::rttnw_params {
        width, height, spp: samples, max_depth: 50, t_min: 0.001, background, seed: 1,
        precision: ffi::RTTNW_F64, quirks: ffi::RTTNW_QUIRKS_REFERENCE, spp_chunk: 0, tile_rank: 0, tile_world: 1,
        collect_counters: 0, sample_begin: 0, reserved0: 0,
    }
}
pub fn reference_camera(lookfrom: [f64; 3], lookat: [f64; 3], vertical_fov: f64, aspect_ratio: f64, aperture: f64) -> ffi::rttnw_camera_desc {
    ffi::rttnw_camera_desc {
        lookfrom, lookat, view_up: [0.0, 1.0, 0.0], vertical_fov, aspect_ratio, aperture,
        focus_distance: 10.0, open_time: 0.0, close_time: 1.0, // main.rs:185-196
    }
}

impl Scene {
    /// `render()` on one GPU: RGBA8, row-major, top row first (main.rs:202-229).
    pub fn render(&self, cam: &ffi::rttnw_camera_desc, p: &ffi::rttnw_params) -> Result<(Vec<u8>, ffi::rttnw_stats)> {
        let mut rgba = vec![0u8; p.width as usize * p.height as usize * 4];
        let mut stats = ffi::rttnw_stats::default();
        ok(unsafe { ffi::rttnw_render(self.raw, cam, p, ptr::null_mut(), rgba.as_mut_ptr(), &mut stats) })?;
        Ok((rgba, stats))
    }
    /// The same image from the GPUs `devices` of this node (tile partition + RCCL gather inside the library).
    pub fn render_multi(&self, cam: &ffi::rttnw_camera_desc, p: &ffi::rttnw_params, devices: &[i32]) -> Result<(Vec<u8>, Vec<ffi::rttnw_stats>)> {
        let mut rgba = vec![0u8; p.width as usize * p.height as usize * 4];
        let mut stats = vec![ffi::rttnw_stats::default(); devices.len()];
        ok(unsafe { ffi::rttnw_render_multi(self.raw, cam, p, devices.len() as u32, devices.as_ptr(), ptr::null_mut(), rgba.as_mut_ptr(), stats.as_mut_ptr()) })?;
        Ok((rgba, stats))
    }
    pub fn build_info(&self) -> Result<ffi::rttnw_build_info> {
        let mut bi = ffi::rttnw_build_info::default();
        ok(unsafe { ffi::rttnw_scene_build_info(self.raw, &mut bi) })?;
        Ok(bi)
    }
}
impl Drop for Scene {
    fn drop(&mut self) {
        if !self.raw.is_null() { unsafe { ffi::rttnw_scene_destroy(self.raw) } }
    }
}
