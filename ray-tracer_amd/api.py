"""ctypes binding of librt_hip.so (the C ABI declared in include/rt_api.h) and numpy views of the wire records.

Nothing here computes: every call goes through the shared library, and loading fails loudly when the
library has not been built (there is no Python or CPU fallback for the render path).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
# RT_HIP_LIB selects another build of the same library (kernel-variant experiments); there is still no non-HIP path.
LIB_PATH = Path(os.environ.get("RT_HIP_LIB", PKG_DIR / "librt_hip.so"))

# ---- wire records (include/rt_types.h) as numpy dtypes ---------------------------------------------------
BVH_NODE = np.dtype([("min", "<f4", 3), ("idx", "<u4"), ("max", "<f4", 3), ("amt", "<u4")])
TRIANGLE = np.dtype([("material_idx", "<u4"), ("is_sphere", "<u4"), ("_pad0", "<u4", 2), ("v", "<f4", (3, 4)),
                     ("n", "<f4", (3, 4)), ("uv", "<f4", (3, 2)), ("_pad1", "<u4", 2)])
MATERIAL = np.dtype([("base_colour", "<f4", 4), ("emission", "<f4", 4), ("metallic", "<f4"), ("roughness", "<f4"),
                     ("subsurface", "<f4"), ("anisotropic", "<f4"), ("specular_tint", "<f4"), ("sheen", "<f4"),
                     ("sheen_tint", "<f4"), ("clearcoat", "<f4"), ("clearcoat_roughness", "<f4"), ("spec_trans", "<f4"),
                     ("ior", "<f4"), ("shade_smooth", "<u4"), ("base_color_tex", "<u4"), ("metallic_roughness_tex", "<u4"),
                     ("normal_tex", "<u4"), ("emission_tex", "<u4")])
LIGHT = np.dtype([("position", "<f4", 4), ("emission", "<f4", 4), ("type", "<u4"), ("_pad", "<u4", 3)])
EMISSIVE_TRI = np.dtype([("tri_index", "<u4"), ("mesh_index", "<u4"), ("area", "<f4"), ("power", "<f4")])
CAMERA = np.dtype([("resolution", "<u4", 2), ("view_port_uv", "<f4", 2), ("focal_length", "<f4"), ("focus_distance", "<f4"),
                   ("aperture_radius", "<f4"), ("frame", "<u4"), ("location", "<f4", 4), ("rotation", "<f4", 16),
                   ("_pad", "<u4", 4)])
HIT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])
assert BVH_NODE.itemsize == 32 and TRIANGLE.itemsize == 144 and MATERIAL.itemsize == 96
SHADE_VERTEX = np.dtype([("ray_o", "<f4", 3), ("prev_pdf", "<f4"), ("ray_d", "<f4", 3), ("seed", "<u4"), ("throughput", "<f4", 3),
                         ("reserved", "<u4"), ("hit", HIT)])
SHADE_RESULT = np.dtype([("emitted", "<f4", 3), ("seed", "<u4"), ("throughput", "<f4", 3), ("pdf", "<f4"), ("next_o", "<f4", 3),
                         ("alive", "<u4"), ("next_d", "<f4", 3), ("has_shadow", "<u4"), ("shadow_o", "<f4", 3), ("shadow_tmax", "<f4"),
                         ("shadow_d", "<f4", 3), ("light_index", "<i4"), ("shadow_contrib", "<f4", 3), ("lobe", "<i4"),
                         ("light_tri", "<i4"), ("light_pdf", "<f4"), ("bsdf_pdf_light", "<f4"), ("roughness", "<f4"),
                         ("medium_throughput", "<f4", 3), ("reserved", "<u4")])
assert SHADE_VERTEX.itemsize == 64 and SHADE_RESULT.itemsize == 144
RT_STATIC_MAX_RAYS = 1 << 18
TUNE_REPORT = np.dtype([("sampled_visits", "<u8", 3), ("resident_records", "<u4", 3), ("resident_share_before", "<f4", 3),
                        ("resident_share_after", "<f4", 3), ("records_moved", "<u4", 3)])
assert TUNE_REPORT.itemsize == 72
assert LIGHT.itemsize == 48 and EMISSIVE_TRI.itemsize == 16 and CAMERA.itemsize == 128 and HIT.itemsize == 16

NO_TEXTURE = 0xFFFFFFFF

# flags (rt_api.h)
RT_REF_RGBA8_ACCUM = 1 << 0
RT_REF_CDF_PAD = 1 << 1
RT_FIX_ABSORPTION = 1 << 2
RT_COUNT_TRAVERSAL = 1 << 3
RT_KERNEL_TIMING = 1 << 4
RT_SINGLE_STREAM = 1 << 5
RT_EXACT_ORDER = 1 << 6
RT_EXACT_ARITH = 1 << 7


class RtConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("max_bounces", C.c_uint32),
                ("flags", C.c_uint32), ("frames_per_pass", C.c_uint32)]


class RtTimings(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("raygen_ms", C.c_float), ("extend_ms", C.c_float), ("shade_ms", C.c_float),
                ("connect_ms", C.c_float), ("accumulate_ms", C.c_float), ("raygen_launches", C.c_uint32),
                ("extend_launches", C.c_uint32), ("shade_launches", C.c_uint32), ("connect_launches", C.c_uint32),
                ("accumulate_launches", C.c_uint32)]


class RtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("camera_rays", "extend_rays", "shadow_rays", "extend_node_pops", "extend_tri_tests",
                                          "shadow_node_pops", "shadow_tri_tests", "extend_leaf_pops", "shadow_leaf_pops", "tail_paths")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RtBvhParams(C.Structure):
    _fields_ = [("traversal_cost", C.c_float), ("intersect_cost", C.c_float), ("min_leaf", C.c_uint32),
                ("max_leaf", C.c_uint32), ("threads", C.c_uint32)]


class RtBvhStats(C.Structure):
    _fields_ = [("sah_cost", C.c_float), ("node_count", C.c_uint32), ("leaf_count", C.c_uint32), ("max_depth", C.c_uint32),
                ("max_leaf_size", C.c_uint32), ("avg_leaf_size", C.c_float), ("build_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RtTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixels", C.c_void_p)]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt status {code}: {msg}")
        self.code = code


# Every symbol include/rt_api.h declares, with its signature. tests check that the library exports all of them.
_P, _U32, _U64, _F, _I = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int
API = {
    "rt_version": (C.c_char_p, []),
    "rt_last_error": (C.c_char_p, [_P]),
    "rt_bvh_build": (_I, [_P, _U32, _P, _P, _P, _P]),
    "rt_bvh_build_sweep": (_I, [_P, _U32, _P, _P, _P]),
    "rt_bvh_build_spatial": (_I, [_P, _U32, _P, C.c_float, C.c_float, _P, _U32, _P, _P, _U32, _P, _P]),
    "rt_build_lights": (_I, [_P, _U32, _P, _U32, _P, _P, _P, _P]),
    "rt_scene_buffer_sizes": (_I, [_U32, _U32, _U32, _U32, _U32, _P]),
    "rt_scene_write_buffers": (_I, [_P, _U32, _P, _U32, _P, _U32, _P, _U32, _P, _P, _U32, _P, _U64]),
    "rt_camera_look_at": (_I, [_P, _U32, _U32, _P, _P, _F, _F, _F]),
    "rt_create": (_I, [_P, _P]),
    "rt_destroy": (None, [_P]),
    "rt_upload_scene": (_I, [_P, _P, _P, _U32, _U32, _U32, _U32, _U32]),
    "rt_upload_textures": (_I, [_P, _P, _U32, _P]),
    "rt_upload_skybox_f32": (_I, [_P, _U32, _U32, _P]),
    "rt_set_camera": (_I, [_P, _P]),
    "rt_resize": (_I, [_P, _U32, _U32]),
    "rt_reset_accumulation": (_I, [_P]),
    "rt_set_frame_slice": (_I, [_P, _U32, _U32]),
    "rt_set_tile_slice": (_I, [_P, _U32, _U32]),
    "rt_render": (_I, [_P, _U32]),
    "rt_sync": (_I, [_P]),
    "rt_sample_count": (_U32, [_P]),
    "rt_read_accum_f32": (_I, [_P, _P, C.c_size_t]),
    "rt_read_sum_f32": (_I, [_P, _P, C.c_size_t]),
    "rt_add_sum_f32": (_I, [_P, _P, C.c_size_t]),
    "rt_accum_device_buffer": (_I, [_P, _P, _P]),
    "rt_read_framebuffer_rgba8": (_I, [_P, _P, C.c_size_t]),
    "rt_last_timings": (_I, [_P, _P]),
    "rt_get_counters": (_I, [_P, _P]),
    "rt_reset_counters": (_I, [_P]),
    "rt_poll": (_I, [_P, _P]),
    "rt_tune_resident_set": (_I, [_P, _P]),
    "rt_trace_closest": (_I, [_P, _U32, _P, _P, _P]),
    "rt_trace_closest_static": (_I, [_P, _U32, _P, _P, _P]),
    "rt_trace_shadow": (_I, [_P, _U32, _P, _P, _P, _P]),
    "rt_shade_vertices": (_I, [_P, _U32, _U32, _P, _P]),
    "rt_bench_trace": (_I, [_P, _U32, _P, _P, _I, _U32, _P]),
    "rt_generate_rays": (_I, [_P, _U32, _P, _P, _P]),
    "rt_eval_bsdf": (_I, [_P, _U32, _P, _P, _P, _P, _P, _P, _P]),
    "rt_sample_bsdf": (_I, [_P, _U32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rt_sample_texture": (_I, [_P, _U32, _U32, _P, _P]),
    "rt_render_radiance": (_I, [_P, _U32, _P]),
}

_lib = None


def load_library(path: os.PathLike | None = None):
    """Loads librt_hip.so and types every entry point. Raises if the library is missing: the render path
    has no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise FileNotFoundError(f"{p} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
    # HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams sharing a queue
    # run one after the other. A context keeps to four streams by default; with more queues, more lanes pay (RT_HIP_LANES).
    # The variable has to be in the environment before the HIP runtime initialises; setting it here works when nothing in
    # the process has touched HIP yet. Where something has (a preloaded library, an earlier import), put
    # GPU_MAX_HW_QUEUES=16 into the process environment instead.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    lib = C.CDLL(str(p))
    for name, (res, args) in API.items():
        fn = getattr(lib, name)  # AttributeError here means the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def ptr(a: np.ndarray | None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def check(code: int, ctx=None):
    if code != 0:
        lib = load_library()
        msg = lib.rt_last_error(ctx)
        raise RtError(code, msg.decode("utf-8", "replace") if msg else "")
    return code
