//! `HipRenderer`: the headless counterpart of `VulkanRenderer` (src/vulkan/renderer.rs:62-507) over librt_hip.so.
//! Method names, argument meaning and the frame-counter rules are the facade's, so `render_loop`
//! (src/app/render_controller.rs:104-333) can hold either back end. Not compiled in the build image (no Rust toolchain);
//! the same call sequences run from C++ (include/hip_renderer.hpp) and Python (ray-tracer_amd/renderer.py) in the tests.
use std::ffi::CStr;
use std::path::Path;

use anyhow::{anyhow, Result};
use glam::{Mat4, Vec3};

use super::ffi::*;
use crate::scene::Scene;

fn check(code: i32, ctx: *const RtCtx) -> Result<()> {
    if code == RT_OK {
        return Ok(());
    }
    let msg = unsafe {
        let p = rt_last_error(ctx);
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    Err(anyhow!("rt status {code}: {msg}"))
}

pub struct HipRenderer {
    ctx: *mut RtCtx,
    scene: Scene,
    width: u32,
    height: u32,
}

// The context is confined to the render thread, like VulkanRenderer (render_controller.rs:24-34).
unsafe impl Send for HipRenderer {}

impl HipRenderer {
    /// VulkanRenderer::create (renderer.rs:62-120) without a window: resolution comes from the scene's camera block.
    pub fn create(scene: Scene, device: i32, max_bounces: u32) -> Result<Self> {
        let cam = scene.get_camera_controls();
        let (width, height) = (cam.resolution.0.x, cam.resolution.0.y);
        let cfg = RtConfig { device, width, height, max_bounces, flags: 0, frames_per_pass: 0 };
        let mut ctx: *mut RtCtx = std::ptr::null_mut();
        check(unsafe { rt_create(&cfg, &mut ctx) }, std::ptr::null())?;
        let mut r = Self { ctx, scene, width, height };
        r.upload_scene()?;
        Ok(r)
    }

    /// upload_scene (renderer.rs:122-131 -> path_tracer/mod.rs:275-288): blob, textures, cube map, camera.
    pub fn upload_scene(&mut self) -> Result<()> {
        let s = self.scene.get_buffer_sizes(); // scene/mod.rs:117-157
        let sizes = [s.0 as u64, s.1 as u64, s.2 as u64, s.3 as u64, s.4 as u64, s.5 as u64];
        let mut blob = vec![0u8; sizes.iter().sum::<u64>() as usize];
        self.scene.write_buffers(blob.as_mut_ptr() as *mut _); // scene/mod.rs:166-195
        let c = &self.scene.components;
        check(
            unsafe {
                rt_upload_scene(
                    self.ctx,
                    blob.as_ptr() as *const _,
                    sizes.as_ptr(),
                    c.bvh.len() as u32,
                    c.materials.len() as u32,
                    c.triangles.len() as u32,
                    c.lights.len() as u32,
                    c.mesh_light_sampler.tris.len() as u32, // the TRUE count: the shader's padded CDF length is a defect
                )
            },
            self.ctx,
        )?;
        // textures are RGBA8 by the time they reach the renderer (path_tracer/mod.rs:99-123)
        let texs: Vec<RtTexture> =
            c.textures.iter().map(|t| RtTexture { width: t.width, height: t.height, pixels: t.pixels.as_ptr() }).collect();
        let sky = RtTexture { width: c.skybox.width, height: c.skybox.height, pixels: c.skybox.pixels.as_ptr() };
        check(unsafe { rt_upload_textures(self.ctx, texs.as_ptr(), texs.len() as u32, &sky) }, self.ctx)?;
        let cam = c.camera;
        check(unsafe { rt_set_camera(self.ctx, &cam as *const _ as *const RtCamera) }, self.ctx)
    }

    /// A cube map of linear fp32 radiance (six faces of height x width x rgba, +x -x +y -y +z -z) in place of the 8-bit one;
    /// no counterpart in VulkanRenderer, whose sky is an RGBA8 image like every other texture.
    pub fn upload_skybox_f32(&mut self, width: u32, height: u32, rgba: &[f32]) -> Result<()> {
        anyhow::ensure!(rgba.len() == (width as usize) * (height as usize) * 24, "skybox needs 6 * width * height * 4 floats");
        check(unsafe { rt_upload_skybox_f32(self.ctx, width, height, rgba.as_ptr()) }, self.ctx)
    }

    /// reload_scene (renderer.rs:193-231): the viewer stays where it is.
    pub fn reload_scene(&mut self, mut new_scene: Scene) -> Result<()> {
        let old = &self.scene.components.camera;
        new_scene.components.camera.location = old.location;
        new_scene.components.camera.rotation = old.rotation;
        self.scene = new_scene;
        self.upload_scene()
    }

    /// set_camera_pose (renderer.rs:165-168): restarts accumulation.
    pub fn set_camera_pose(&mut self, location: Vec3, rotation: Mat4) -> Result<()> {
        self.scene.set_camera_pose(location, rotation);
        let cam = self.scene.components.camera;
        check(unsafe { rt_set_camera(self.ctx, &cam as *const _ as *const RtCamera) }, self.ctx)
    }

    /// reset_accumulator (renderer.rs:170-177)
    pub fn reset_accumulator(&mut self) {
        unsafe { rt_reset_accumulation(self.ctx) };
    }

    /// sample_count (renderer.rs:180-182)
    pub fn sample_count(&self) -> u32 {
        unsafe { rt_sample_count(self.ctx) }
    }

    /// render_resolution (renderer.rs:234-237)
    pub fn render_resolution(&self) -> (u32, u32) {
        (self.width, self.height)
    }

    /// handle_resize (renderer.rs:133-162): restarts accumulation; a failed resize keeps the old resolution.
    pub fn handle_resize(&mut self, width: u32, height: u32) {
        if width == 0 || height == 0 {
            return;
        }
        match check(unsafe { rt_resize(self.ctx, width, height) }, self.ctx) {
            Ok(()) => {
                self.width = width;
                self.height = height;
            }
            Err(e) => log::error!("path tracer resize failed: {e}"),
        }
    }

    /// dispatch_compute (renderer.rs:269-321): one sample per call, asynchronous. The slot index of the Vulkan back end has
    /// no meaning here (the library orders its own passes); `dispatch_samples` queues several at once, which is faster.
    pub fn dispatch_compute(&mut self, _slot: usize) -> Result<()> {
        check(unsafe { rt_render(self.ctx, 1) }, self.ctx)
    }
    pub fn dispatch_samples(&mut self, n: u32) -> Result<()> {
        check(unsafe { rt_render(self.ctx, n) }, self.ctx)
    }

    /// frame_complete (renderer.rs:251-267): the fence status the render loop polls while it keeps dispatches in flight
    /// (render_controller.rs:233-244). The library has one timeline, so every slot is done when all queued work is.
    pub fn frame_complete(&mut self, _slot: usize) -> Result<bool> {
        let mut done = 0i32;
        check(unsafe { rt_poll(self.ctx, &mut done) }, self.ctx)?;
        Ok(done != 0)
    }

    /// Blocks until everything queued has finished (headless use).
    pub fn wait_idle(&mut self) -> Result<()> {
        check(unsafe { rt_sync(self.ctx) }, self.ctx)
    }

    /// SaveFrame (renderer.rs:336-444; utils/save_frame.rs:27-65): the tone-mapped RGBA8 frame as a PNG.
    pub fn save_frame(&mut self, path: &Path) -> Result<()> {
        let mut rgba = vec![0u8; (self.width * self.height * 4) as usize];
        check(unsafe { rt_read_framebuffer_rgba8(self.ctx, rgba.as_mut_ptr(), rgba.len()) }, self.ctx)?;
        image::save_buffer(path, &rgba, self.width, self.height, image::ColorType::Rgba8)?;
        Ok(())
    }

    /// HDR mean image (new capability: the Vulkan back end accumulates in 8 bits, path_tracer/mod.rs:206-217).
    pub fn read_accum(&mut self) -> Result<Vec<f32>> {
        let mut rgba = vec![0f32; (self.width * self.height * 4) as usize];
        check(unsafe { rt_read_accum_f32(self.ctx, rgba.as_mut_ptr(), rgba.len()) }, self.ctx)?;
        Ok(rgba)
    }

    /// last_timer_perf (renderer.rs:239-248): per-kernel device times of the last call (needs RT_KERNEL_TIMING).
    pub fn last_timings(&self) -> Result<RtTimings> {
        let mut t: RtTimings = unsafe { std::mem::zeroed() };
        check(unsafe { rt_last_timings(self.ctx, &mut t) }, self.ctx)?;
        Ok(t)
    }

    /// Orders the node records by measured use for the scene and camera now installed (no counterpart upstream; output
    /// bits do not change). Worth calling once after `create` / `reload_scene` on scenes whose trees exceed the caches.
    pub fn tune_resident_set(&mut self) -> Result<RtTuneReport> {
        let mut rep: RtTuneReport = unsafe { std::mem::zeroed() };
        check(unsafe { rt_tune_resident_set(self.ctx, &mut rep) }, self.ctx)?;
        Ok(rep)
    }

    /// Multi-GPU: this renderer takes global frame indices first, first + stride, ... (one renderer per GPU).
    pub fn set_frame_slice(&mut self, first: u32, stride: u32) -> Result<()> {
        check(unsafe { rt_set_frame_slice(self.ctx, first, stride) }, self.ctx)
    }

    /// Multi-GPU, the other partition: this renderer takes the 8x8 tiles first, first + stride, ... of every frame.
    pub fn set_tile_slice(&mut self, first: u32, stride: u32) -> Result<()> {
        check(unsafe { rt_set_tile_slice(self.ctx, first, stride) }, self.ctx)
    }
}

/// destroy (renderer.rs:493-507)
impl Drop for HipRenderer {
    fn drop(&mut self) {
        unsafe { rt_destroy(self.ctx) }
    }
}
