"""`HipRenderer`: the headless counterpart of the reference's `VulkanRenderer` facade
(src/vulkan/renderer.rs:62-507), with the same method names and frame-counter rules, bound to librt_hip.so.

    create / upload_scene / reload_scene / set_camera_pose / reset_accumulator / sample_count /
    render_resolution / handle_resize / dispatch_compute / frame_complete / save_frame / last_timer_perf / destroy

`dispatch_compute(n)` queues n samples (the reference dispatches one per call); everything else keeps the
reference's meaning. Stage-level methods (trace_closest, eval_bsdf, ...) expose single kernels for parity tests.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api
from .api import CAMERA, HIT, MATERIAL, RtConfig, RtCounters, RtTexture, RtTimings, check, load_library, ptr
from .scene import Scene


def _f32(a, shape_last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None:
        a = a.reshape(-1, shape_last)
    return a


class HipRenderer:
    def __init__(self, width: int, height: int, max_bounces: int = 24, flags: int = 0, device: int = 0, frames_per_pass: int = 0):
        self.lib = load_library()
        self._ctx = C.c_void_p(None)
        cfg = RtConfig(device, width, height, max_bounces, flags, frames_per_pass)
        check(self.lib.rt_create(C.byref(cfg), C.byref(self._ctx)))
        self.width, self.height = width, height
        self.flags = flags
        self.max_bounces = max_bounces
        self._keepalive = []

    # VulkanRenderer::create(window, scene): here without a window
    @classmethod
    def create(cls, scene: Scene, max_bounces: int = 24, flags: int = 0, device: int = 0, frames_per_pass: int = 0,
               width: int | None = None, height: int | None = None):
        w = int(scene.camera["resolution"][0][0]) if width is None else width
        h = int(scene.camera["resolution"][0][1]) if height is None else height
        r = cls(w, h, max_bounces, flags, device, frames_per_pass)
        r.upload_scene(scene)
        return r

    def _check(self, code):
        return check(code, self._ctx)

    def upload_skybox_hdr(self, faces):
        """A cube map of linear fp32 radiance, (6, h, w, 4) or (6, h, w, 3), in place of the 8-bit one (no counterpart upstream)."""
        f = np.asarray(faces, dtype=np.float32)
        if f.ndim != 4 or f.shape[0] != 6 or f.shape[3] not in (3, 4):
            raise ValueError("skybox_hdr must have shape (6, h, w, 3 or 4)")
        if f.shape[3] == 3:
            f = np.concatenate([f, np.ones(f.shape[:3] + (1,), np.float32)], axis=-1)
        f = np.ascontiguousarray(f)
        self._check(self.lib.rt_upload_skybox_f32(self._ctx, f.shape[2], f.shape[1], f.ctypes.data))

    # upload_scene (renderer.rs:122-131) / reload_scene (renderer.rs:193-231)
    def upload_scene(self, scene: Scene):
        blob = scene.write_buffers()
        sizes = scene.get_buffer_sizes()
        self._check(self.lib.rt_upload_scene(self._ctx, ptr(blob), ptr(sizes), *scene.counts()))
        texs = (RtTexture * max(1, len(scene.textures)))()
        keep = []
        for i, t in enumerate(scene.textures):
            t = np.ascontiguousarray(t, dtype=np.uint8)
            assert t.ndim == 3 and t.shape[2] == 4
            keep.append(t)
            texs[i] = RtTexture(t.shape[1], t.shape[0], t.ctypes.data)
        sky = None
        if scene.skybox is not None:
            s = np.ascontiguousarray(scene.skybox, dtype=np.uint8)
            assert s.ndim == 4 and s.shape[0] == 6 and s.shape[3] == 4
            keep.append(s)
            sky = RtTexture(s.shape[2], s.shape[1], s.ctypes.data)
        self._check(self.lib.rt_upload_textures(self._ctx, texs, len(scene.textures), C.byref(sky) if sky else None))
        if getattr(scene, "skybox_hdr", None) is not None:
            self.upload_skybox_hdr(scene.skybox_hdr)
        cam = np.ascontiguousarray(scene.camera)
        self._check(self.lib.rt_set_camera(self._ctx, ptr(cam)))
        self._camera_in_use = cam.copy()
        self.scene = scene

    # reload_scene (renderer.rs:193-231): the new scene is viewed from where the camera currently is
    # (location and rotation are carried over, renderer.rs:194-196); everything else comes from the new scene.
    def reload_scene(self, scene: Scene):
        import copy
        current = getattr(self, "_camera_in_use", None)          # what set_camera / set_camera_pose / upload_scene installed last
        if current is not None:
            cam = np.ascontiguousarray(scene.camera).copy()
            cam["location"][0] = current["location"][0]
            cam["rotation"][0] = current["rotation"][0]
            scene = copy.copy(scene)                              # the caller's Scene keeps its own camera
            scene.camera = cam
        self.upload_scene(scene)

    # No counterpart upstream: orders the node records by how often this scene and camera visit them, so the traversal
    # kernels keep the busiest ones in LDS. Rendered values do not change. Returns what was measured.
    def tune_resident_set(self) -> dict:
        rep = np.zeros(1, api.TUNE_REPORT)
        self._check(self.lib.rt_tune_resident_set(self._ctx, ptr(rep)))
        return {k: rep[k][0].tolist() for k in rep.dtype.names}

    # set_camera_pose (renderer.rs:165-168): location + camera->world rotation (column-major 4x4), restarts accumulation
    def set_camera_pose(self, location, rotation):
        cam = np.ascontiguousarray(self.scene.camera).copy()
        cam["location"][0][:3] = np.asarray(location, dtype=np.float32)
        cam["rotation"][0] = np.asarray(rotation, dtype=np.float32).reshape(16)
        self.scene.camera = cam
        self._check(self.lib.rt_set_camera(self._ctx, ptr(cam)))
        self._camera_in_use = cam.copy()

    def set_camera(self, cam: np.ndarray):
        cam = np.ascontiguousarray(cam)
        self._check(self.lib.rt_set_camera(self._ctx, ptr(cam)))
        self._camera_in_use = cam.copy()

    def reset_accumulator(self):
        self._check(self.lib.rt_reset_accumulation(self._ctx))

    def sample_count(self) -> int:
        return int(self.lib.rt_sample_count(self._ctx))

    def render_resolution(self):
        return (self.width, self.height)

    def handle_resize(self, width: int, height: int):
        self._check(self.lib.rt_resize(self._ctx, width, height))
        self.width, self.height = width, height

    def set_frame_slice(self, first: int, stride: int):
        self._check(self.lib.rt_set_frame_slice(self._ctx, first, stride))

    def set_tile_slice(self, first: int, stride: int):
        """This renderer takes the 8x8 tiles first, first + stride, ... of every frame; the rest of its accumulator stays zero."""
        self._check(self.lib.rt_set_tile_slice(self._ctx, first, stride))

    # dispatch_compute (renderer.rs:269-321): asynchronous
    def dispatch_compute(self, n_samples: int = 1):
        self._check(self.lib.rt_render(self._ctx, n_samples))

    # frame_complete (renderer.rs:251-267)
    def frame_complete(self):
        self._check(self.lib.rt_sync(self._ctx))

    # frame_complete as the reference's render loop uses it (renderer.rs:251-267, render_controller.rs:233-244): a fence
    # status, not a wait
    def frame_done(self) -> bool:
        done = C.c_int(0)
        self._check(self.lib.rt_poll(self._ctx, C.byref(done)))
        return bool(done.value)

    def render(self, n_samples: int):
        self.dispatch_compute(n_samples)
        self.frame_complete()

    def read_accum(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._check(self.lib.rt_read_accum_f32(self._ctx, ptr(out), out.size))
        return out

    def read_sum(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._check(self.lib.rt_read_sum_f32(self._ctx, ptr(out), out.size))
        return out

    def add_sum(self, rgba_sum: np.ndarray):
        a = _f32(rgba_sum)
        self._check(self.lib.rt_add_sum_f32(self._ctx, ptr(a), a.size))

    def accum_device_buffer(self):
        p = C.c_void_p(None)
        n = C.c_size_t(0)
        self._check(self.lib.rt_accum_device_buffer(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_framebuffer(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._check(self.lib.rt_read_framebuffer_rgba8(self._ctx, ptr(out), out.size))
        return out

    # SaveFrame (renderer.rs:336-444, utils/save_frame.rs:27-65): tone-mapped RGBA PNG
    def save_frame(self, path):
        from PIL import Image
        Image.fromarray(self.read_framebuffer(), "RGBA").save(path)

    # last_timer_perf (renderer.rs:239-248)
    def last_timer_perf(self) -> dict:
        t = RtTimings()
        self._check(self.lib.rt_last_timings(self._ctx, C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_}

    def counters(self) -> dict:
        c = RtCounters()
        self._check(self.lib.rt_get_counters(self._ctx, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._check(self.lib.rt_reset_counters(self._ctx))

    # ---- stage-level entry points ----
    def trace_closest(self, origins, directions) -> np.ndarray:
        o, d = _f32(origins, 3), _f32(directions, 3)
        hits = np.zeros(len(o), dtype=HIT)
        self._check(self.lib.rt_trace_closest(self._ctx, len(o), ptr(o), ptr(d), ptr(hits)))
        return hits

    def trace_closest_static(self, origins, directions) -> np.ndarray:
        """rt_trace_closest without any device allocation (at most api.RT_STATIC_MAX_RAYS rays per call)."""
        o, d = _f32(origins, 3), _f32(directions, 3)
        hits = np.zeros(len(o), dtype=HIT)
        self._check(self.lib.rt_trace_closest_static(self._ctx, len(o), ptr(o), ptr(d), ptr(hits)))
        return hits

    def shade_vertices(self, bounce: int, vertices) -> np.ndarray:
        """The shade stage on caller-supplied vertices (api.SHADE_VERTEX records) -> api.SHADE_RESULT records."""
        v = np.ascontiguousarray(vertices, dtype=api.SHADE_VERTEX)
        out = np.zeros(len(v), dtype=api.SHADE_RESULT)
        self._check(self.lib.rt_shade_vertices(self._ctx, len(v), bounce, ptr(v), ptr(out)))
        return out

    def trace_shadow(self, origins, directions, tmax) -> np.ndarray:
        o, d, t = _f32(origins, 3), _f32(directions, 3), _f32(tmax)
        vis = np.zeros(len(o), dtype=np.uint8)
        self._check(self.lib.rt_trace_shadow(self._ctx, len(o), ptr(o), ptr(d), ptr(t), ptr(vis)))
        return vis

    def bench_trace(self, origins, directions, shadow=False, repeat=5) -> float:
        o, d = _f32(origins, 3), _f32(directions, 3)
        ms = C.c_float(0)
        self._check(self.lib.rt_bench_trace(self._ctx, len(o), ptr(o), ptr(d), int(shadow), repeat, C.byref(ms)))
        return float(ms.value)

    def generate_rays(self, frame: int):
        n = self.width * self.height
        o = np.zeros((n, 3), dtype=np.float32)
        d = np.zeros((n, 3), dtype=np.float32)
        s = np.zeros(n, dtype=np.uint32)
        self._check(self.lib.rt_generate_rays(self._ctx, frame, ptr(o), ptr(d), ptr(s)))
        return o, d, s

    def eval_bsdf(self, mats, eta, V, N, L):
        mats = np.ascontiguousarray(mats, dtype=MATERIAL)
        n = len(mats)
        f = np.zeros((n, 3), dtype=np.float32)
        pdf = np.zeros(n, dtype=np.float32)
        self._check(self.lib.rt_eval_bsdf(self._ctx, n, ptr(mats), ptr(_f32(eta)), ptr(_f32(V, 3)), ptr(_f32(N, 3)), ptr(_f32(L, 3)),
                                          ptr(f), ptr(pdf)))
        return f, pdf

    def sample_bsdf(self, mats, eta, V, N, r1r2):
        mats = np.ascontiguousarray(mats, dtype=MATERIAL)
        n = len(mats)
        L = np.zeros((n, 3), dtype=np.float32)
        f = np.zeros((n, 3), dtype=np.float32)
        pdf = np.zeros(n, dtype=np.float32)
        self._check(self.lib.rt_sample_bsdf(self._ctx, n, ptr(mats), ptr(_f32(eta)), ptr(_f32(V, 3)), ptr(_f32(N, 3)),
                                            ptr(_f32(r1r2, 2)), ptr(L), ptr(f), ptr(pdf)))
        return L, f, pdf

    def sample_texture(self, tex: int, uvw):
        u = _f32(uvw, 3)
        out = np.zeros((len(u), 4), dtype=np.float32)
        self._check(self.lib.rt_sample_texture(self._ctx, tex, len(u), ptr(u), ptr(out)))
        return out

    def render_radiance(self, frame: int) -> np.ndarray:
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self._check(self.lib.rt_render_radiance(self._ctx, frame, ptr(out)))
        return out

    # destroy (renderer.rs:493-507)
    def destroy(self):
        if self._ctx:
            self.lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
