"""Standalone traversal timing on a fixed ray set (camera rays of the 1080p atrium and the diffuse bounce rays they spawn),
for library variants (RT_HIP_LIB): ms per launch of 2 M rays, closest-hit and shadow queries. usage: trace_ablation.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import scenes

sc = scenes.atrium(width=1920, height=1080).finalize(spatial=True)
r = rt.HipRenderer.create(sc, max_bounces=8)
o, d, _ = r.generate_rays(0)
h = r.trace_closest(o, d)
hit = h["prim"] != 0xFFFFFFFF
rng = np.random.default_rng(5)
p = (o + d * h["t"][:, None])[hit]
nd = rng.normal(size=p.shape).astype(np.float32)
nd /= np.linalg.norm(nd, axis=1, keepdims=True)
p = (p + nd * 1e-3).astype(np.float32)
name = os.path.basename(os.environ.get("RT_HIP_LIB", "librt_hip.so"))
for label, oo, dd in (("camera", o, d), ("bounce", p, nd)):
    ms = [r.bench_trace(oo, dd, shadow=False, repeat=5) for _ in range(3)]
    ms_s = [r.bench_trace(oo, dd, shadow=True, repeat=5) for _ in range(3)]
    print(f"{name} {label} rays ({len(oo)}): closest {min(ms):.3f} ms ({len(oo) / min(ms) / 1e3:.0f} Mrays/s), shadow {min(ms_s):.3f} ms ({len(oo) / min(ms_s) / 1e3:.0f} Mrays/s)", flush=True)
r.destroy()
