"""Share of a small pass spent between kernels: device time first launch to last against the sum of kernel times, and the
wall time per call, for one-sample passes. usage: launch_gaps.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

for name, kw, b in (("cornell", dict(width=256, height=256), 4), ("cornell", dict(width=1920, height=1080), 4), ("atrium", dict(width=1920, height=1080), 8), ("atrium", dict(width=480, height=270), 8)):
    sc = scenes.GENERATORS[name](**kw).finalize(spatial=True)
    r = rt.HipRenderer.create(sc, max_bounces=b, flags=api.RT_KERNEL_TIMING, frames_per_pass=1)
    r.render(4); r.frame_complete()
    r.render(1); r.frame_complete()
    tm = r.last_timer_perf()
    ks = sum(tm[f"{k}_ms"] for k in ("raygen", "extend", "shade", "connect", "accumulate"))
    nl = sum(tm[f"{k}_launches"] for k in ("raygen", "extend", "shade", "connect", "accumulate"))
    r.destroy()
    r = rt.HipRenderer.create(sc, max_bounces=b, frames_per_pass=1)
    r.render(8); r.frame_complete()
    t0 = time.perf_counter()
    for _ in range(50):
        r.render(1); r.frame_complete()
    sync_ms = (time.perf_counter() - t0) * 1e3 / 50
    t0 = time.perf_counter(); r.dispatch_compute(50); r.frame_complete(); async_ms = (time.perf_counter() - t0) * 1e3 / 50
    t0 = time.perf_counter()
    for _ in range(50):
        r.dispatch_compute(1)
    enqueue_ms = (time.perf_counter() - t0) * 1e3 / 50
    r.frame_complete()
    t0 = time.perf_counter()
    for _ in range(50):
        r.dispatch_compute(1)
    r.frame_complete()
    back_to_back_ms = (time.perf_counter() - t0) * 1e3 / 50
    r.destroy()
    print(f"{name} {kw['width']}x{kw['height']}: one pass = {nl} launches, {tm['total_ms']:.3f} ms first to last, kernels {ks:.3f} ms ({100 * (1 - ks / tm['total_ms']):.0f} % between kernels); "
          f"wall per sample: call+wait {sync_ms:.3f} ms, 50 in one call {async_ms:.3f} ms, 50 calls then one wait {back_to_back_ms:.3f} ms, host time to queue one pass {enqueue_ms:.3f} ms", flush=True)
