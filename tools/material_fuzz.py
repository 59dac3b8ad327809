"""Extreme and broken material parameters through the renderer on the GPU box: zeros, ones, negatives, huge values, NaN and
infinities in every float field, out-of-range texture ids. Each scene must render (finite or not) and return; prints the
share of finite pixels per case."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

rng = np.random.default_rng(11)
base = scenes.cornell(width=96, height=96).finalize()
floats = [n for n in base.materials.dtype.names if base.materials.dtype[n].kind == "f"]
texids = [n for n in base.materials.dtype.names if n.endswith("_tex")]
values = [0.0, 1.0, -1.0, 0.5, 1e-30, 1e30, -1e30, np.nan, np.inf, -np.inf, 2.0, 1e-8]
t0 = time.time()
n_cases = 0
worst = 1.0
for it in range(60):
    import dataclasses
    m = base.materials.copy()
    for name in floats:
        col = m[name]
        mask = rng.random(col.shape) < 0.35
        col[mask] = rng.choice(values, size=int(mask.sum())).astype(np.float32)
    for name in texids:
        if rng.random() < 0.3:
            m[name][rng.integers(0, len(m))] = rng.choice([0, 1, 7, 1000, 0xFFFFFFFE])
    if "shade_smooth" in m.dtype.names:
        m["shade_smooth"] = rng.integers(0, 2, len(m))
    sc = dataclasses.replace(base, materials=m)
    with np.errstate(all="ignore"):
        sc.build_lights()
    for flags in (0, api.RT_EXACT_ORDER, api.RT_FIX_ABSORPTION):
        r = rt.HipRenderer.create(sc, max_bounces=6, flags=flags)
        r.render(4)
        img = r.read_accum()
        fin = float(np.isfinite(img).mean())
        worst = min(worst, fin)
        n_cases += 1
        r.destroy()
    if it % 10 == 9:
        print(f"{it + 1} material sets, {n_cases} renders, lowest share of finite pixels {worst:.3f}, {time.time() - t0:.1f} s", flush=True)
print("RESULT ok")
