"""How the default closest-hit kernel's time grows with the number of (cheap, coherent) camera rays. usage: cw_scaling.py scene w h [--scale=s]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kw = {}
for a in sys.argv:
    if a.startswith("--scale="):
        kw["scale"] = float(a.split("=")[1])
sc = scenes.GENERATORS[name](width=w, height=h, **kw).finalize(spatial=True)
rs = {"default": rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_COUNT_TRAVERSAL), "exact": rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_EXACT_ORDER | api.RT_COUNT_TRAVERSAL)}
o, d, _ = rs["exact"].generate_rays(0)
o1, d1, _ = rs["exact"].generate_rays(1)
cases = {"frame 0": (o, d), "frame 1": (o1, d1), "frame 0 twice": (np.concatenate([o, o]), np.concatenate([d, d])),
         "frame 0 x4": (np.concatenate([o] * 4), np.concatenate([d] * 4)), "frame 0 x4 shuffled": None, "half of frame 0 x8": (np.concatenate([o[:len(o) // 2]] * 8), np.concatenate([d[:len(o) // 2]] * 8))}
rng = np.random.default_rng(1)
O4, D4 = cases["frame 0 x4"]
perm = rng.permutation(len(O4))
cases["frame 0 x4 shuffled"] = (O4[perm], D4[perm])
for label, (O, D) in cases.items():
    row = []
    for tag, r in rs.items():
        ms = r.bench_trace(O, D, repeat=3)
        r.reset_counters(); r.trace_closest(O[:1 << 20], D[:1 << 20]); c = r.counters()
        row.append(f"{tag} {ms:.3f} ms ({ms * 1e6 / len(O):.0f} ps/ray, {c['extend_node_pops'] / (1 << 20):.1f} nodes/ray)")
    print(f"{label:22s} {len(O):9d} rays: " + "   ".join(row), flush=True)
