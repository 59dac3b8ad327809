"""Census of the shade-stage differences between the HIP kernel and the CPU oracle, vertex by vertex (GPU box). Not a test
(pytest does not collect it): run it by name, `python tests/census_shade_outliers.py`."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ray_tracer_amd as rt
from ray_tracer_amd import scenes
from oracle import pyoracle as po
from util import big_tables_scene

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)

for name, kw in [("cornell", {"width": 64, "height": 64}), ("glass_blob", {"scale": 0.05, "width": 96, "height": 54}),
                 ("interior", {"scale": 0.01, "width": 96, "height": 54}), ("car", {"scale": 0.03, "width": 96, "height": 54})]:
    sc = scenes.GENERATORS[name](**kw).finalize()
    r = rt.HipRenderer.create(sc, max_bounces=8); o = po.OracleScene(sc)
    for b in (0, 1, 4):
        v, reached = o.path_states(1, 8, b); v = v[reached]
        if len(v) == 0: continue
        g = r.shade_vertices(b, v); h = o.shade_vertices(b, v)
        ok = (v["hit"]["prim"] != 0xFFFFFFFF) & np.isfinite(h["pdf"]) & (g["alive"] == h["alive"])
        e_pdf = rel(g["pdf"], h["pdf"]); e_thr = rel(g["throughput"], h["throughput"]).max(axis=1)
        mat = sc.materials[sc.triangles["material_idx"][np.where(ok, v["hit"]["prim"], 0)]]
        for what, e in (("pdf", e_pdf), ("thr", e_thr)):
            bad = np.flatnonzero(ok & (e > 1e-5))
            print(f"{name} b{b} {what}: {len(bad)} of {ok.sum()} beyond 1e-5; max {e[ok].max():.3g}")
            for i in bad[:8]:
                print(f"   lobe {h['lobe'][i]} rough {mat['roughness'][i]:.4f} ccr {mat['clearcoat_roughness'][i]:.3f} mrtex {mat['metallic_roughness_tex'][i]!r} "
                      f"pdf g {g['pdf'][i]:.7g} h {h['pdf'][i]:.7g} rel {e[i]:.3g} thr h {h['throughput'][i]}")
    r.destroy()
