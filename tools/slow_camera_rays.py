"""Camera rays that cost the default traversal far more than the exact-order one. usage: slow_camera_rays.py scene w h [--scale=s]"""
import os, sys, time
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
print(f"{name}: {len(sc.triangles)} triangles; camera at {sc.camera['location'][0][:3]}", flush=True)
rd_ = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_COUNT_TRAVERSAL)
rx = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_COUNT_TRAVERSAL | api.RT_EXACT_ORDER)
o, d, _ = rx.generate_rays(0)
B = 1 << 16
per = {}
for r, tag in ((rd_, "default"), (rx, "exact")):
    rows = []
    for i in range(0, len(o), B):
        r.reset_counters()
        t0 = time.perf_counter()
        r.trace_closest(o[i:i + B], d[i:i + B])
        dt = time.perf_counter() - t0
        c = r.counters()
        rows.append((c["extend_node_pops"] / B, c["extend_tri_tests"] / B, dt * 1e3))
    per[tag] = np.array(rows)
    print(f"{tag}: nodes/ray mean {per[tag][:,0].mean():.1f} max batch {per[tag][:,0].max():.1f}; tris/ray mean {per[tag][:,1].mean():.1f} max {per[tag][:,1].max():.1f}; "
          f"call ms mean {per[tag][:,2].mean():.2f} max {per[tag][:,2].max():.2f}", flush=True)
ratio = per["default"][:, 0] / np.maximum(per["exact"][:, 0], 1e-3)
for b in np.argsort(-ratio)[:4]:
    print(f"batch {b} (rows {b * B // w}..): default {per['default'][b]} exact {per['exact'][b]}")
b = int(np.argmax(per["default"][:, 0]))
i0 = b * B
cnt = []
for k in range(i0, i0 + B, 64):
    rd_.reset_counters(); rd_.trace_closest(o[k:k + 64], d[k:k + 64]); cnt.append(rd_.counters()["extend_node_pops"] / 64)
cnt = np.array(cnt)
k = int(cnt.argmax())
print(f"worst 64-ray group in batch {b}: {cnt[k]:.0f} nodes/ray (median group {np.median(cnt):.0f})")
for j in range(i0 + k * 64, i0 + k * 64 + 64, 8):
    rd_.reset_counters(); hd = rd_.trace_closest(o[j:j + 1], d[j:j + 1]); nd = rd_.counters()["extend_node_pops"]
    rx.reset_counters(); hx = rx.trace_closest(o[j:j + 1], d[j:j + 1]); nx = rx.counters()["extend_node_pops"]
    print(f"  ray {j}: default {nd} nodes (t={hd['t'][0]:.3f} prim {hd['prim'][0]}), exact {nx} nodes (t={hx['t'][0]:.3f} prim {hx['prim'][0]})  d={d[j]}")
