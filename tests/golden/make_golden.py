#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

The reference has no tests or vectors for this path and cannot be run in this image, so the fixtures are
(a) known answers derived by hand / from first principles (pcg_vectors.json, triangle_hits.json), and
(b) dumps of the CPU oracle on the synthetic Cornell scene at 64x64, frames 0-3, 4 bounces (cornell64.npz),
which freeze the oracle's behaviour so later edits to it are detected.
Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
HERE = Path(__file__).resolve().parent


def pcg_vectors():
    from util import pcg_reference
    out = {}
    for seed in (0, 1, 12345, 0xFFFFFFFF, 1920 * 1080 - 1):
        w, s, f = pcg_reference(seed, 8)
        out[str(seed)] = {"words": w, "states": s, "floats": [float(x) for x in f]}
    (HERE / "pcg_vectors.json").write_text(json.dumps(out, indent=1))


def triangle_hits():
    """Single triangle A=(0,0,0) B=(1,0,0) C=(0,1,0); rays chosen so t,u,v are exact in binary32."""
    cases = [
        # origin, direction, expect (hit, t, u, v)
        {"o": [0.25, 0.25, 1.0], "d": [0, 0, -1], "hit": True, "t": 1.0, "u": 0.25, "v": 0.25},
        {"o": [0.25, 0.25, -2.0], "d": [0, 0, 1], "hit": True, "t": 2.0, "u": 0.25, "v": 0.25},     # back face also hits
        {"o": [0.0, 0.0, 1.0], "d": [0, 0, -1], "hit": True, "t": 1.0, "u": 0.0, "v": 0.0},          # vertex: bounds inclusive
        {"o": [0.5, 0.5, 1.0], "d": [0, 0, -1], "hit": True, "t": 1.0, "u": 0.5, "v": 0.5},          # on the hypotenuse (w = 0)
        {"o": [0.75, 0.5, 1.0], "d": [0, 0, -1], "hit": False},                                        # outside (w < 0)
        {"o": [-0.125, 0.5, 1.0], "d": [0, 0, -1], "hit": False},                                      # u < 0
        {"o": [0.25, 0.25, 1.0], "d": [0, 0, 1], "hit": False},                                        # behind the origin (t < 0)
        {"o": [0.25, 0.25, 0.00005], "d": [0, 0, -1], "hit": False},                                   # t < EPSILON = 1e-4
        {"o": [0.25, 0.25, 0.0002], "d": [0, 0, -1], "hit": True, "t": 0.0002, "u": 0.25, "v": 0.25, "approx": True},
        {"o": [0.25, 0.25, 1.0], "d": [1, 0, 0], "hit": False},                                        # parallel to the plane
        {"o": [0.25, 0.25, 1.0], "d": [1, 0, -0.00005], "hit": False, "normalize": True},             # |n.d| < 1e-4: grazing reject
    ]
    (HERE / "triangle_hits.json").write_text(json.dumps(cases, indent=1))


def cornell_dump():
    from ray_tracer_amd import scenes
    from oracle import pyoracle as po
    po.build()
    sc = scenes.cornell(64, 64).finalize()
    o = po.OracleScene(sc)
    data = {}
    for frame in range(4):
        ro, rd, seeds = o.generate_rays(frame)
        hits, _ = o.trace_closest(ro, rd)
        rad, cnt = o.render_radiance(frame, 4)
        data[f"rays_o_{frame}"], data[f"rays_d_{frame}"], data[f"seeds_{frame}"] = ro, rd, seeds
        data[f"hit_t_{frame}"], data[f"hit_prim_{frame}"] = hits["t"], hits["prim"]
        data[f"hit_u_{frame}"], data[f"hit_v_{frame}"] = hits["u"], hits["v"]
        data[f"radiance_{frame}"] = rad.astype(np.float32)
        data[f"counts_{frame}"] = np.array([cnt["rays"], cnt["shadow_rays"]], dtype=np.int64)
    s, a8, _ = o.render(4, 4)
    data["sum4"] = s
    _, a8, _ = o.render(4, 4, flags=po.F_REF_RGBA8)
    data["accum8_4"] = a8
    np.savez_compressed(HERE / "cornell64.npz", **data)


if __name__ == "__main__":
    pcg_vectors()
    triangle_hits()
    cornell_dump()
    print("golden fixtures written to", HERE)
