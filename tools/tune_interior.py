"""Interior-class scene before and after rt_tune_resident_set: ms per sample and the share of node visits served from LDS.
usage: tune_interior.py [scene] [width] [height] [samples]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "interior"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
t0 = time.time()
sc = scenes.GENERATORS[name](width=w, height=h).finalize(spatial=True)
print(f"{name}: {len(sc.triangles)} triangles, {len(sc.bvh)} nodes, built in {time.time() - t0:.1f} s", flush=True)
out = {"scene": name, "resolution": [w, h], "triangles": int(len(sc.triangles)), "samples_timed": n}
for mode, flags in (("default", 0), ("exact_order", api.RT_EXACT_ORDER)):
    r = rt.HipRenderer.create(sc, max_bounces=8, flags=flags)
    def timed():
        r.reset_accumulator(); r.reset_counters(); r.frame_complete()
        t = time.perf_counter(); r.render(n); r.frame_complete(); dt = time.perf_counter() - t
        c = r.counters()
        return dt * 1e3 / n, (c["extend_rays"] + c["shadow_rays"]) / dt / 1e6
    r.render(2); r.frame_complete()
    before = [timed() for _ in range(3)]
    img0 = r.read_accum()
    t = time.perf_counter(); rep = r.tune_resident_set(); tune_ms = (time.perf_counter() - t) * 1e3
    after = [timed() for _ in range(3)]
    same = bool(np.array_equal(img0, r.read_accum(), equal_nan=True))
    r.destroy()
    out[mode] = {"ms_per_sample_before": min(b[0] for b in before), "ms_per_sample_after": min(a[0] for a in after),
                 "mrays_per_s_before": max(b[1] for b in before), "mrays_per_s_after": max(a[1] for a in after),
                 "tune_call_ms": tune_ms, "image_identical": same, "report": rep}
    print(mode, json.dumps(out[mode]), flush=True)
print(json.dumps(out))
