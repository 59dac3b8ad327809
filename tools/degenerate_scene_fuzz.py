"""Degenerate geometry through the whole path on the GPU box: triangle soups with zero-area, collinear, point-like, huge, tiny,
NaN and infinite triangles, built with each builder, uploaded, traced in both traversal modes and rendered. Prints what each
case does; any exception other than a documented rejection, or hits that differ beyond eps-ties between the modes, fails."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api
from ray_tracer_amd.scene import Scene, default_material, look_at_camera, make_triangles

rng = np.random.default_rng(5)
cam = look_at_camera(64, 64, (0, 0, 6), (0, 0, 0))

def soup(n, kind):
    v = rng.normal(size=(n, 3, 3)).astype(np.float32)
    if kind == "zero_area": v[::3, 1] = v[::3, 0]
    if kind == "collinear": v[::2, 2] = v[::2, 0] + 2 * (v[::2, 1] - v[::2, 0])
    if kind == "points": v[:, 1] = v[:, 0]; v[:, 2] = v[:, 0]
    if kind == "huge": v[::5] *= 1e30
    if kind == "tiny": v *= 1e-30
    if kind == "nan": v[::7, 0, 0] = np.nan
    if kind == "inf": v[::7, 1, 1] = np.inf
    if kind == "same": v[:] = v[0]
    if kind == "flat": v[..., 2] = 0
    if kind == "line": v[..., 1:] = 0
    return v

bad = 0
for kind in ("ok", "zero_area", "collinear", "points", "huge", "tiny", "nan", "inf", "same", "flat", "line"):
    for n in (2, 65, 3000):
        for mode in ("object", "spatial", "sweep"):
            v = soup(n, kind)
            mats = default_material(2)
            mats["emission"][1][:3] = (5, 5, 5)
            sc = Scene(triangles=make_triangles(v, None, None, (np.arange(n) % 17 == 0).astype(np.uint32)), materials=mats, camera=cam)
            with np.errstate(all="ignore"):
                sc.build_bvh(sweep=True) if mode == "sweep" else sc.build_bvh(spatial=(mode == "spatial"))
                sc.build_lights()
            tag = f"{kind:10s} n={n:5d} {mode:7s}"
            try:
                rd_ = rt.HipRenderer.create(sc, max_bounces=4)
                rx = rt.HipRenderer.create(sc, max_bounces=4, flags=api.RT_EXACT_ORDER)
            except rt.api.RtError as e:
                ok = "UNSUPPORTED" in str(e).upper() or "-4" in str(e) or "64 triangles" in str(e)
                print(tag, "rejected:", str(e)[:90], "" if ok else "  <-- UNEXPECTED", flush=True)
                bad += 0 if ok else 1
                continue
            o = rng.normal(size=(4000, 3)).astype(np.float32) * 3
            d = rng.normal(size=(4000, 3)).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            t0 = time.time()
            g, h = rd_.trace_closest(o, d), rx.trace_closest(o, d)
            both = (g["prim"] != 0xFFFFFFFF) & (h["prim"] != 0xFFFFFFFF)
            with np.errstate(all="ignore"):
                tie = both & (np.abs(g["t"].astype(np.float64) - h["t"]) <= 1e-6 * np.abs(h["t"].astype(np.float64)))
            differ = g["prim"] != h["prim"]
            other = differ & ~tie & ~((g["prim"] != 0xFFFFFFFF) & ((h["prim"] == 0xFFFFFFFF) | (g["t"] < h["t"])))
            rd_.render(3); rx.render(3)
            a, b = rd_.read_accum(), rx.read_accum()
            dt = time.time() - t0
            print(tag, f"hits {int((g['prim'] != 0xFFFFFFFF).sum()):4d}/{int((h['prim'] != 0xFFFFFFFF).sum()):4d} ids differ {int(differ.sum()):3d} (ties {int((differ & tie).sum())}, unexplained {int(other.sum())}); "
                  f"render finite {bool(np.isfinite(a).all())}/{bool(np.isfinite(b).all())}; {dt * 1e3:.0f} ms", flush=True)
            bad += int(other.sum() > 0)
            rd_.destroy(); rx.destroy()
print("RESULT", "ok" if bad == 0 else f"{bad} cases need a look")
