"""Fixed cost (run-down tail) of one closest-hit launch: time the kernel on n, 2n, 4n copies of a set of second-bounce rays
(T(n) = a n + tail, so tail = 2 T(n) - T(2n)). usage: tail_estimate.py [scene w h]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
sc = scenes.GENERATORS[name](width=w, height=h).finalize(spatial=True)
r = rt.HipRenderer.create(sc, max_bounces=8)
o, d, seeds = r.generate_rays(0)
hits = r.trace_closest(o, d)
v = np.zeros(len(o), api.SHADE_VERTEX)
v["ray_o"], v["ray_d"], v["seed"], v["throughput"] = o, d, seeds, 1.0
for f in ("t", "u", "v", "prim"):
    v["hit"][f] = hits[f]
res = r.shade_vertices(0, v)
a = res["alive"] != 0
so, sd = np.ascontiguousarray(res["next_o"][a]), np.ascontiguousarray(res["next_d"][a])
print(f"{name}: {len(so)} second-bounce rays", flush=True)
rng = np.random.default_rng(3)
T = {}
for k in (1, 2, 4, 8):
    O, D = np.concatenate([so] * k), np.concatenate([sd] * k)
    T[k] = min(r.bench_trace(O, D, repeat=3) for _ in range(2))
    print(f"  {k} x: {len(O):9d} rays {T[k]:.3f} ms  ({T[k] * 1e6 / len(O):.0f} ps/ray)", flush=True)
for k in (1, 2, 4):
    print(f"  tail estimate from {k}x and {2 * k}x: {2 * T[k] - T[2 * k]:.3f} ms; slope {(T[2 * k] - T[k]) * 1e6 / (k * len(so)):.0f} ps/ray")
