"""Render a small scene on CU-masked streams (RT_HIP_CUS) and on the whole chip; the images must be identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ray_tracer_amd as rt
from ray_tracer_amd import scenes

def run(cus, scene, n):
    if cus:
        os.environ["RT_HIP_CUS"] = str(cus)
    else:
        os.environ.pop("RT_HIP_CUS", None)
    print(f"cus={cus}: create", flush=True)
    r = rt.HipRenderer.create(scene, max_bounces=4)
    os.environ.pop("RT_HIP_CUS", None)
    print("  render", flush=True)
    r.render(1); r.frame_complete(); r.reset_accumulator()
    t = time.perf_counter(); r.render(n); r.frame_complete(); dt = time.perf_counter() - t
    img = r.read_accum()
    r.destroy()
    print(f"  {n} samples in {dt * 1e3:.2f} ms", flush=True)
    return img

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
if name == "cornell":
    sc = scenes.cornell(width=256, height=256).finalize()
else:
    sc = scenes.atrium(width=960, height=540, scale=0.25).finalize()
a = run(0, sc, 4) if "--masked-only" not in sys.argv else None
b = run(32, sc, 4)
print("identical:", a is None or np.array_equal(a, b, equal_nan=True), flush=True)
