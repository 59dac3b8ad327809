// C++ host-side mirror of the reference's renderer facade for the path-tracing step.
//
// The reference application talks to its GPU back end only through `VulkanRenderer`
// (src/vulkan/mod.rs:1-15; src/vulkan/renderer.rs:62-507) and feeds it a `Scene` (src/scene/mod.rs:67-205).
// `rt::Scene` and `rt::HipRenderer` below keep those names, argument meanings and frame-counter rules over the
// C ABI of librt_hip.so, so host code written against the facade ports line by line. Header-only; link -lrt_hip.
//
// Errors: the reference returns anyhow::Result and logs; here every failing call throws rt::Error carrying the
// status code and the library's message (nothing is swallowed, nothing aborts).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rt_api.h"

namespace rt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("rt status " + std::to_string(c) + ": " + m), code(c) {}
};

inline void check(int rc, const rt_ctx* ctx = nullptr) {
    if (rc != RT_OK) {
        const char* m = rt_last_error(ctx);
        throw Error(rc, m ? m : "");
    }
}

// Scene components in wire layout (`SceneComponents` upstream).
struct Scene {
    std::vector<rt_triangle> triangles;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_bvh_node> bvh;
    std::vector<rt_emissive_tri> emissive_tris;
    std::vector<float> cdf;
    float total_power = 0.0f;
    rt_camera camera{};
    rt_bvh_stats bvh_stats{};

    // Scene::build_bvh (src/scene/mod.rs:84-115): reorders `triangles` into leaf order.
    void build_bvh(const rt_bvh_params* params = nullptr) {
        bvh.assign(triangles.empty() ? 1 : 2 * triangles.size(), rt_bvh_node{});
        uint32_t n = 0;
        check(rt_bvh_build(triangles.data(), uint32_t(triangles.size()), params, bvh.data(), &n, &bvh_stats));
        bvh.resize(n);
    }
    // Scene::build_lights (src/scene/mod.rs:207-270); call after build_bvh.
    void build_lights() {
        emissive_tris.assign(triangles.size() + 1, rt_emissive_tri{});
        cdf.assign(triangles.size() + 1, 0.0f);
        uint32_t n = 0;
        check(rt_build_lights(triangles.data(), uint32_t(triangles.size()), materials.data(), uint32_t(materials.size()),
                              emissive_tris.data(), cdf.data(), &n, &total_power));
        emissive_tris.resize(n);
        cdf.resize(n);
    }
    // Scene::get_buffer_sizes (src/scene/mod.rs:117-157)
    void get_buffer_sizes(uint64_t sizes[6]) const {
        check(rt_scene_buffer_sizes(uint32_t(bvh.size()), uint32_t(materials.size()), uint32_t(triangles.size()),
                                    uint32_t(lights.size()), uint32_t(emissive_tris.size()), sizes));
    }
    // Scene::write_buffers (src/scene/mod.rs:166-195)
    std::vector<uint8_t> write_buffers() const {
        uint64_t sz[6];
        get_buffer_sizes(sz);
        std::vector<uint8_t> blob(sz[0] + sz[1] + sz[2] + sz[3] + sz[4] + sz[5]);
        check(rt_scene_write_buffers(bvh.data(), uint32_t(bvh.size()), materials.data(), uint32_t(materials.size()), triangles.data(),
                                     uint32_t(triangles.size()), lights.data(), uint32_t(lights.size()), emissive_tris.data(), cdf.data(),
                                     uint32_t(emissive_tris.size()), blob.data(), blob.size()));
        return blob;
    }
};

class HipRenderer {
public:
    // VulkanRenderer::create (renderer.rs:62-120), headless: resolution comes from the scene's camera block.
    static HipRenderer create(const Scene& scene, uint32_t max_bounces = 24, uint32_t flags = 0, int device = 0) {
        rt_config cfg{device, scene.camera.resolution[0], scene.camera.resolution[1], max_bounces, flags, 0};
        HipRenderer r;
        check(rt_create(&cfg, &r.ctx_));
        r.upload_scene(scene);
        return r;
    }
    HipRenderer(HipRenderer&& o) noexcept : ctx_(std::exchange(o.ctx_, nullptr)), camera_in_use_(o.camera_in_use_), have_camera_(o.have_camera_) {}
    HipRenderer& operator=(HipRenderer&& o) noexcept {
        destroy(); ctx_ = std::exchange(o.ctx_, nullptr); camera_in_use_ = o.camera_in_use_; have_camera_ = o.have_camera_; return *this;
    }
    HipRenderer(const HipRenderer&) = delete;
    HipRenderer& operator=(const HipRenderer&) = delete;
    ~HipRenderer() { destroy(); }

    // upload_scene / reload_scene (renderer.rs:122-131,193-231): also installs the scene's camera and restarts accumulation.
    void upload_scene(const Scene& scene) {
        uint64_t sz[6];
        scene.get_buffer_sizes(sz);
        std::vector<uint8_t> blob = scene.write_buffers();
        check(rt_upload_scene(ctx_, blob.data(), sz, uint32_t(scene.bvh.size()), uint32_t(scene.materials.size()),
                              uint32_t(scene.triangles.size()), uint32_t(scene.lights.size()), uint32_t(scene.emissive_tris.size())), ctx_);
        check(rt_set_camera(ctx_, &scene.camera), ctx_);
        camera_in_use_ = scene.camera; have_camera_ = true;
    }
    // reload_scene (renderer.rs:193-231): the new scene is viewed from the current camera pose (location and rotation
    // are carried over, renderer.rs:194-196).
    void reload_scene(Scene scene) {
        if (have_camera_) {
            for (int k = 0; k < 4; ++k) scene.camera.location[k] = camera_in_use_.location[k];
            for (int k = 0; k < 16; ++k) scene.camera.rotation[k] = camera_in_use_.rotation[k];
        }
        upload_scene(scene);
    }
    // the same with the pose given by the caller
    void reload_scene(Scene scene, const rt_camera& current) {
        for (int k = 0; k < 4; ++k) scene.camera.location[k] = current.location[k];
        for (int k = 0; k < 16; ++k) scene.camera.rotation[k] = current.rotation[k];
        upload_scene(scene);
    }
    void upload_textures(const std::vector<rt_texture>& textures, const rt_texture* skybox) {
        check(rt_upload_textures(ctx_, textures.data(), uint32_t(textures.size()), skybox), ctx_);
    }
    // fp32 cube map of linear radiance, 6 x height x width x rgba, in place of the 8-bit one (no counterpart upstream)
    void upload_skybox_f32(uint32_t width, uint32_t height, const float* rgba) { check(rt_upload_skybox_f32(ctx_, width, height, rgba), ctx_); }
    // set_camera_pose (renderer.rs:165-168)
    void set_camera(const rt_camera& cam) { check(rt_set_camera(ctx_, &cam), ctx_); camera_in_use_ = cam; have_camera_ = true; }
    void reset_accumulator() { check(rt_reset_accumulation(ctx_), ctx_); }         // renderer.rs:170-177
    uint32_t sample_count() const { return rt_sample_count(ctx_); }                  // renderer.rs:180-182
    void handle_resize(uint32_t w, uint32_t h) { check(rt_resize(ctx_, w, h), ctx_); }   // renderer.rs:133-162
    // No counterpart upstream: orders the node records by measured use for the scene and camera now installed, so the
    // traversal kernels keep the busiest ones in LDS. Output bits do not change.
    rt_tune_report tune_resident_set() { rt_tune_report rep{}; check(rt_tune_resident_set(ctx_, &rep), ctx_); return rep; }
    void set_frame_slice(uint32_t first, uint32_t stride) { check(rt_set_frame_slice(ctx_, first, stride), ctx_); }
    void set_tile_slice(uint32_t first, uint32_t stride) { check(rt_set_tile_slice(ctx_, first, stride), ctx_); }
    // dispatch_compute (renderer.rs:269-321): queues n samples, asynchronous.
    void dispatch_compute(uint32_t n_samples = 1) { check(rt_render(ctx_, n_samples), ctx_); }
    void frame_complete() { check(rt_sync(ctx_), ctx_); }                              // waits (headless use)
    bool frame_done() { int d = 0; check(rt_poll(ctx_, &d), ctx_); return d != 0; }      // frame_complete as a fence status, renderer.rs:251-267
    void read_accum(std::vector<float>& rgba) { check(rt_read_accum_f32(ctx_, rgba.data(), rgba.size()), ctx_); }
    void read_framebuffer(std::vector<uint8_t>& rgba) { check(rt_read_framebuffer_rgba8(ctx_, rgba.data(), rgba.size()), ctx_); }
    rt_timings last_timer_perf() const { rt_timings t{}; check(rt_last_timings(ctx_, &t), ctx_); return t; }   // renderer.rs:239-248
    rt_counters counters() { rt_counters c{}; check(rt_get_counters(ctx_, &c), ctx_); return c; }
    rt_ctx* raw() const { return ctx_; }
    void destroy() { if (ctx_) { rt_destroy(ctx_); ctx_ = nullptr; } }              // renderer.rs:493-507

private:
    HipRenderer() = default;
    rt_ctx* ctx_ = nullptr;
    rt_camera camera_in_use_{};      // what upload_scene / set_camera installed last: the pose reload_scene carries over
    bool have_camera_ = false;
};

}  // namespace rt
