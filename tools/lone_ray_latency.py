"""Per-visit latency of a ray that has a wavefront to itself (what the run-down of every traversal launch is made of): picks
second-bounce rays with many node visits and times each alone. usage: lone_ray_latency.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
sc = scenes.GENERATORS[name](width=480, height=270).finalize(spatial=True)
r = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_COUNT_TRAVERSAL)
o, d, seeds = r.generate_rays(0)
hits = r.trace_closest(o, d)
v = np.zeros(len(o), api.SHADE_VERTEX)
v["ray_o"], v["ray_d"], v["seed"], v["throughput"] = o, d, seeds, 1.0
for f in ("t", "u", "v", "prim"):
    v["hit"][f] = hits[f]
res = r.shade_vertices(0, v)
a = res["alive"] != 0
so, sd = np.ascontiguousarray(res["next_o"][a]), np.ascontiguousarray(res["next_d"][a])
# node visits per ray, in groups of 64 first, then singly inside the costliest groups
groups = []
for k in range(0, min(len(so), 64 * 400), 64):
    r.reset_counters(); r.trace_closest(so[k:k + 64], sd[k:k + 64]); groups.append(r.counters()["extend_node_pops"])
top = np.argsort(-np.array(groups))[:3]
rows = []
for g in top:
    for k in range(g * 64, g * 64 + 64):
        r.reset_counters(); r.trace_closest(so[k:k + 1], sd[k:k + 1]); c = r.counters()
        rows.append((c["extend_node_pops"], c["extend_leaf_pops"], c["extend_tri_tests"], k))
rows.sort(reverse=True)
empty_ms = r.bench_trace(so[:1] + 1e6, sd[:1], repeat=20)          # a ray that misses the scene bounds: launch cost only
print(f"{name}: launch with one ray that misses everything: {empty_ms * 1e3:.1f} us")
for nodes, leaves, tris, k in rows[:6] + rows[len(rows) // 2: len(rows) // 2 + 2]:
    ms = r.bench_trace(so[k:k + 1], sd[k:k + 1], repeat=20)
    print(f"  ray {k}: {nodes} node visits, {leaves} leaves, {tris} triangles: {ms * 1e3:.1f} us alone = {(ms - empty_ms) * 1e3 / max(1, nodes + leaves):.2f} us per visit", flush=True)
