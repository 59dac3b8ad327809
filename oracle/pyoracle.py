"""TEST INFRASTRUCTURE: ctypes wrapper of the CPU parity oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
never does."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"

F_REF_RGBA8, F_REF_CDF_PAD, F_FIX_ABSORPTION = 1, 2, 4


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "shadow_rays", "node_pops", "leaf_pops", "tri_tests", "shadow_node_pops",
                                          "shadow_leaf_pops", "shadow_tri_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Tex(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixels", C.c_void_p)]


def build(force=False):
    src = HERE / "oracle.cpp"
    if force or not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s"], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            raise FileNotFoundError(f"{LIB} missing: run `make -C oracle`")
        _lib = C.CDLL(str(LIB))
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_pcg.restype = C.c_float
        _lib.orc_pcg_word.restype = C.c_uint32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(a, last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if last is None else a.reshape(-1, last)


HIT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])
# stage-probe records of include/rt_api.h (rt_shade_vertex, rt_shade_result)
SHADE_VERTEX = np.dtype([("ray_o", "<f4", 3), ("prev_pdf", "<f4"), ("ray_d", "<f4", 3), ("seed", "<u4"), ("throughput", "<f4", 3),
                         ("reserved", "<u4"), ("hit", HIT)])
SHADE_RESULT = np.dtype([("emitted", "<f4", 3), ("seed", "<u4"), ("throughput", "<f4", 3), ("pdf", "<f4"), ("next_o", "<f4", 3),
                         ("alive", "<u4"), ("next_d", "<f4", 3), ("has_shadow", "<u4"), ("shadow_o", "<f4", 3), ("shadow_tmax", "<f4"),
                         ("shadow_d", "<f4", 3), ("light_index", "<i4"), ("shadow_contrib", "<f4", 3), ("lobe", "<i4"),
                         ("light_tri", "<i4"), ("light_pdf", "<f4"), ("bsdf_pdf_light", "<f4"), ("roughness", "<f4"),
                         ("medium_throughput", "<f4", 3), ("reserved", "<u4")])
assert SHADE_VERTEX.itemsize == 64 and SHADE_RESULT.itemsize == 144


class OracleScene:
    """Wraps a scene in reference wire layout (the same blob the renderer uploads)."""

    def __init__(self, scene):
        L = lib()
        blob = scene.write_buffers()
        sizes = scene.get_buffer_sizes()
        nb, nm, nt, nl, ne = scene.counts()
        self.h = C.c_void_p(L.orc_scene_create(_p(blob), _p(sizes), C.c_uint32(nb), C.c_uint32(nm), C.c_uint32(nt), C.c_uint32(nl), C.c_uint32(ne)))
        self.cam = np.ascontiguousarray(scene.camera).copy()
        self.keep = []
        texs = (Tex * max(1, len(scene.textures)))()
        for i, t in enumerate(scene.textures):
            t = np.ascontiguousarray(t, dtype=np.uint8)
            self.keep.append(t)
            texs[i] = Tex(t.shape[1], t.shape[0], t.ctypes.data)
        sky = None
        if scene.skybox is not None:
            s = np.ascontiguousarray(scene.skybox, dtype=np.uint8)
            self.keep.append(s)
            sky = Tex(s.shape[2], s.shape[1], s.ctypes.data)
        L.orc_scene_set_textures(self.h, texs, C.c_uint32(len(scene.textures)), C.byref(sky) if sky else None)
        if getattr(scene, "skybox_hdr", None) is not None:
            f = np.asarray(scene.skybox_hdr, dtype=np.float32)
            if f.shape[3] == 3:
                f = np.concatenate([f, np.ones(f.shape[:3] + (1,), np.float32)], axis=-1)
            f = np.ascontiguousarray(f)
            L.orc_scene_set_skybox_f32(self.h, C.c_uint32(f.shape[2]), C.c_uint32(f.shape[1]), _p(f))

    def __del__(self):
        try:
            lib().orc_scene_destroy(self.h)
        except Exception:
            pass

    def set_resolution(self, w, h):
        self.cam["resolution"][0] = (w, h)
        r = np.float32(w) / np.float32(h)
        self.cam["view_port_uv"][0] = (r, 1.0) if r > 1.0 else (1.0, np.float32(1.0) / r)

    @property
    def npix(self):
        return int(self.cam["resolution"][0][0]) * int(self.cam["resolution"][0][1])

    def generate_rays(self, frame):
        n = self.npix
        o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32); s = np.zeros(n, np.uint32)
        lib().orc_generate_rays(_p(self.cam), C.c_uint32(frame), _p(o), _p(d), _p(s))
        return o, d, s

    def trace_closest(self, o, d, threads=0):
        o, d = _f(o, 3), _f(d, 3)
        hits = np.zeros(len(o), dtype=HIT)
        cnt = Counters()
        lib().orc_trace_closest(self.h, C.c_uint32(len(o)), _p(o), _p(d), _p(hits), C.byref(cnt), C.c_uint32(threads))
        return hits, cnt.as_dict()

    def trace_brute(self, o, d):
        o, d = _f(o, 3), _f(d, 3)
        hits = np.zeros(len(o), dtype=HIT)
        lib().orc_trace_brute(self.h, C.c_uint32(len(o)), _p(o), _p(d), _p(hits))
        return hits

    def test_triangles(self, o, d, prim):
        """hit_triangle of ray i against triangle prim[i] alone."""
        o, d = _f(o, 3), _f(d, 3)
        prim = np.ascontiguousarray(prim, dtype=np.uint32)
        hits = np.zeros(len(o), dtype=HIT)
        lib().orc_test_triangles(self.h, C.c_uint32(len(o)), _p(o), _p(d), _p(prim), _p(hits))
        return hits

    def trace_shadow(self, o, d, tmax):
        o, d, t = _f(o, 3), _f(d, 3), _f(tmax)
        vis = np.zeros(len(o), dtype=np.uint8)
        lib().orc_trace_shadow(self.h, C.c_uint32(len(o)), _p(o), _p(d), _p(t), _p(vis))
        return vis

    def sample_texture(self, tex, uvw):
        u = _f(uvw, 3)
        out = np.zeros((len(u), 4), np.float32)
        lib().orc_sample_texture(self.h, C.c_uint32(tex), C.c_uint32(len(u)), _p(u), _p(out))
        return out

    def render_radiance(self, frame, max_bounces, flags=0, threads=0):
        W, H = int(self.cam["resolution"][0][0]), int(self.cam["resolution"][0][1])
        rgb = np.zeros((H, W, 3), np.float32)
        cnt = Counters()
        lib().orc_render_radiance(self.h, _p(self.cam), C.c_uint32(frame), C.c_uint32(max_bounces), C.c_uint32(flags), _p(rgb), C.byref(cnt),
                                  C.c_uint32(threads))
        return rgb, cnt.as_dict()

    def path_states(self, frame, max_bounces, bounce, flags=0, threads=0):
        """Every pixel's path of `frame` on arrival at vertex `bounce`: (SHADE_VERTEX records, reached mask)."""
        n = self.npix
        out = np.zeros(n, SHADE_VERTEX)
        reached = np.zeros(n, np.uint8)
        lib().orc_path_states(self.h, _p(self.cam), C.c_uint32(frame), C.c_uint32(max_bounces), C.c_uint32(flags), C.c_uint32(bounce),
                              _p(out), _p(reached), C.c_uint32(threads))
        return out, reached.astype(bool)

    def shade_vertices(self, bounce, vertices, flags=0):
        v = np.ascontiguousarray(vertices, dtype=SHADE_VERTEX)
        out = np.zeros(len(v), SHADE_RESULT)
        lib().orc_shade_vertices(self.h, C.c_uint32(len(v)), C.c_uint32(bounce), C.c_uint32(flags), _p(v), _p(out))
        return out

    def render(self, n_frames, max_bounces, flags=0, first=0, stride=1, threads=0, sum_rgba=None, accum8=None):
        W, H = int(self.cam["resolution"][0][0]), int(self.cam["resolution"][0][1])
        if sum_rgba is None:
            sum_rgba = np.zeros((H, W, 4), np.float32)
        if accum8 is None:
            accum8 = np.zeros((H, W, 4), np.uint8)
        cnt = Counters()
        lib().orc_render(self.h, _p(self.cam), C.c_uint32(first), C.c_uint32(stride), C.c_uint32(n_frames), C.c_uint32(max_bounces), C.c_uint32(flags),
                         _p(sum_rgba), _p(accum8), C.byref(cnt), C.c_uint32(threads))
        return sum_rgba, accum8, cnt.as_dict()


def set_libm_bias(sin_ulp=0, cos_ulp=0, other_ulp=0):
    """Nudges sin / cos / {log, exp, pow} inside the oracle's shading code by that many ulp (conditioning probe)."""
    lib().orc_set_libm_bias(C.c_int(sin_ulp), C.c_int(cos_ulp), C.c_int(other_ulp))


def pcg_floats(seed, n):
    s = C.c_uint32(seed)
    return np.array([lib().orc_pcg(C.byref(s)) for _ in range(n)], dtype=np.float32), s.value


def pcg_words(seed, n):
    s = C.c_uint32(seed)
    words, states = [], []
    for _ in range(n):
        words.append(lib().orc_pcg_word(C.byref(s)))
        states.append(s.value)
    return words, states


def eval_bsdf(mats, eta, V, N, L):
    n = len(mats)
    f = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
    mats = np.ascontiguousarray(mats)
    lib().orc_eval_bsdf(C.c_uint32(n), _p(mats), _p(_f(eta)), _p(_f(V, 3)), _p(_f(N, 3)), _p(_f(L, 3)), _p(f), _p(pdf))
    return f, pdf


def sample_bsdf(mats, eta, V, N, r):
    n = len(mats)
    L = np.zeros((n, 3), np.float32); f = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
    mats = np.ascontiguousarray(mats)
    lib().orc_sample_bsdf(C.c_uint32(n), _p(mats), _p(_f(eta)), _p(_f(V, 3)), _p(_f(N, 3)), _p(_f(r, 2)), _p(L), _p(f), _p(pdf))
    return L, f, pdf


def tonemap(mean_rgba):
    m = _f(mean_rgba).reshape(-1, 4)
    out = np.zeros((len(m), 4), np.uint8)
    lib().orc_tonemap(C.c_uint32(len(m)), _p(m), _p(out))
    return out


def suppress_fireflies(rgb):
    a = _f(rgb).reshape(-1, 3).copy()
    lib().orc_suppress_fireflies(C.c_uint32(len(a)), _p(a))
    return a
