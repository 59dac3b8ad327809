"""Camera rays of frames 0..3 that are non-finite or have zero direction components, and what each traversal mode does with them."""
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
for f in range(4):
    o, d, _ = rs["exact"].generate_rays(f)
    bad = ~(np.isfinite(o).all(axis=1) & np.isfinite(d).all(axis=1))
    zero = (d == 0).any(axis=1) & ~bad
    print(f"frame {f}: non-finite rays {int(bad.sum())}, rays with a zero direction component {int(zero.sum())}", flush=True)
    for label, m in (("non-finite", bad), ("zero component", zero)):
        idx = np.flatnonzero(m)[:3]
        for i in idx:
            row = []
            for tag, r in rs.items():
                r.reset_counters(); hh = r.trace_closest(o[i:i + 1], d[i:i + 1]); c = r.counters()
                row.append(f"{tag}: {c['extend_node_pops']} nodes, {c['extend_tri_tests']} tris, prim {hh['prim'][0]}")
            print(f"   {label} ray {i} (pixel {i % w},{i // w}) o={o[i]} d={d[i]}: " + "; ".join(row), flush=True)
