"""Special rays against both traversal modes on the GPU box: per class, node visits per ray in the default (compressed) and the
exact-order traversal, and how many hits differ beyond eps-ties. A class whose default cost is a multiple of its exact cost is
a pathology of the kind a zero direction component was. usage: ray_fuzz.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

names = sys.argv[1:] or ["atrium", "interior", "glass_blob", "car", "cornell"]
rng = np.random.default_rng(99)
N = 20000
worst = 0.0
for name in names:
    kw = {} if name == "cornell" else {"scale": 0.25}
    sc = scenes.GENERATORS[name](width=64, height=36, **kw).finalize(spatial=True)
    v = sc.triangles["v"][..., :3]
    lo, hi = v.min(axis=(0, 1)), v.max(axis=(0, 1))
    size = float(np.abs(v).max())
    rd_ = rt.HipRenderer.create(sc, max_bounces=4, flags=api.RT_COUNT_TRAVERSAL)
    rx = rt.HipRenderer.create(sc, max_bounces=4, flags=api.RT_COUNT_TRAVERSAL | api.RT_EXACT_ORDER)

    def unit(d):
        n = np.linalg.norm(d, axis=1, keepdims=True)
        return (d / np.where(n > 0, n, 1)).astype(np.float32)

    inside = lambda: rng.uniform(lo, hi, size=(N, 3)).astype(np.float32)
    tri = rng.integers(0, len(v), N)
    bary = rng.dirichlet([1, 1, 1], N).astype(np.float32)
    on_surface = (v[tri] * bary[:, :, None]).sum(axis=1).astype(np.float32)
    classes = {}
    classes["random"] = (inside(), unit(rng.normal(size=(N, 3))))
    d = rng.normal(size=(N, 3)); d[:, rng.integers(0, 3)] = 0.0
    classes["one zero component"] = (inside(), unit(d))
    d = rng.normal(size=(N, 3)); d[np.arange(N), rng.integers(0, 3, N)] = -0.0
    classes["one negative-zero component"] = (inside(), unit(d))
    d = np.zeros((N, 3)); d[np.arange(N), rng.integers(0, 3, N)] = rng.choice([-1.0, 1.0], N)
    classes["along an axis"] = (inside(), d.astype(np.float32))
    d = rng.normal(size=(N, 3)); d[np.arange(N), rng.integers(0, 3, N)] *= 1e-30
    classes["one component 1e-30"] = (inside(), unit(d))
    d = rng.normal(size=(N, 3)); d[np.arange(N), rng.integers(0, 3, N)] = 1e-42          # denormal after normalisation
    classes["one denormal component"] = (inside(), unit(d))
    classes["starting on a surface"] = (on_surface, unit(rng.normal(size=(N, 3))))
    classes["starting on a vertex"] = (v[tri, rng.integers(0, 3, N)].astype(np.float32), unit(rng.normal(size=(N, 3))))
    e0, e1 = v[tri, 0], v[tri, 1]
    classes["along a triangle edge"] = ((e0 - 0.5 * (e1 - e0)).astype(np.float32), unit(e1 - e0))
    classes["towards a vertex"] = (inside(), None)
    o_ = classes["towards a vertex"][0]
    classes["towards a vertex"] = (o_, unit(v[tri, 0] - o_))
    far = unit(rng.normal(size=(N, 3))) * np.float32(25.0 * size)
    classes["from 25 scene sizes away"] = (far.astype(np.float32), unit(rng.uniform(lo, hi, size=(N, 3)) - far))
    snap = inside(); k = rng.integers(0, 3, N); snap[np.arange(N), k] = np.where(rng.integers(0, 2, N) == 0, lo[k], hi[k])
    classes["starting on the scene bounds"] = (snap, unit(rng.normal(size=(N, 3))))
    print(f"== {name}: {len(v)} triangles", flush=True)
    for label, (o, d) in classes.items():
        rd_.reset_counters(); g = rd_.trace_closest(o, d); cd = rd_.counters()
        rx.reset_counters(); h = rx.trace_closest(o, d); cx = rx.counters()
        both = (g["prim"] != 0xFFFFFFFF) & (h["prim"] != 0xFFFFFFFF)
        tie = both & (np.abs(g["t"].astype(np.float64) - h["t"]) <= 1e-6 * np.abs(h["t"].astype(np.float64)))
        differ = (g["prim"] != h["prim"])
        beyond = differ & ~tie
        closer = beyond & (g["t"] < h["t"])                 # hits the exact order's box test lost
        ratio = cd["extend_node_pops"] / max(1, cx["extend_node_pops"])
        tri_ratio = cd["extend_tri_tests"] / max(1, cx["extend_tri_tests"])
        worst = max(worst, ratio)
        print(f"  {label:32s} nodes/ray default {cd['extend_node_pops'] / N:7.1f} exact {cx['extend_node_pops'] / N:7.1f} (x{ratio:.2f})  tris x{tri_ratio:.2f}  "
              f"ids differ {int(differ.sum()):5d}: ties {int((differ & tie).sum()):5d}, default closer {int(closer.sum()):4d}, other {int((beyond & ~closer).sum()):4d}", flush=True)
    rd_.destroy(); rx.destroy()
print(f"worst default/exact node ratio {worst:.2f}")
