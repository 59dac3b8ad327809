#!/usr/bin/env python3
"""How much does the order of the ray list matter to the closest-hit kernel? Builds second-bounce-like rays on the
atrium (primary hit points + cosine-distributed directions about the geometric normal), then times rt_bench_trace on the
same rays in (a) pixel order, the order the renderer's queues have, (b) a random shuffle, (c) sorted by direction octant
and a Morton code of the origin. Runs on the GPU box."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import scenes  # noqa: E402


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    W, H = 1920, 1080
    sc = scenes.atrium(W, H).finalize(spatial=True)
    r = rt.HipRenderer.create(sc, max_bounces=2)
    rng = np.random.default_rng(1)
    os_, ds_ = [], []
    for frame in range(2):
        o, d, _ = r.generate_rays(frame)
        h = r.trace_closest(o, d)
        ok = h["prim"] != 0xFFFFFFFF
        tri = sc.triangles["v"][h["prim"][ok]][..., :3]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        n /= np.linalg.norm(n, axis=1, keepdims=True) + 1e-30
        n[np.einsum("ij,ij->i", n, d[ok]) > 0] *= -1
        p = o[ok] + d[ok] * h["t"][ok, None] + n * 1e-3
        # cosine-weighted direction about n
        u1, u2 = rng.random(len(p)), rng.random(len(p))
        rr, phi = np.sqrt(u1), 2 * np.pi * u2
        a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
        t1 = np.cross(n, a); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
        t2 = np.cross(n, t1)
        dd = t1 * (rr * np.cos(phi))[:, None] + t2 * (rr * np.sin(phi))[:, None] + n * np.sqrt(1 - u1)[:, None]
        os_.append(p.astype(np.float32)); ds_.append(dd.astype(np.float32))
    o2, d2 = np.concatenate(os_), np.concatenate(ds_)
    n = len(o2)
    lo, hi = o2.min(0), o2.max(0)
    q = ((o2 - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint32)
    octant = ((d2[:, 0] > 0).astype(np.uint64) | ((d2[:, 1] > 0).astype(np.uint64) << 1) | ((d2[:, 2] > 0).astype(np.uint64) << 2))
    orders = {
        "pixel order (as rendered)": np.arange(n),
        "random shuffle": rng.permutation(n),
        "sorted: octant, Morton(origin)": np.argsort((octant << np.uint64(30)) | morton3(q), kind="stable"),
        "sorted: Morton(origin) only": np.argsort(morton3(q), kind="stable"),
        "sorted: coarse cell (5 bits/axis), octant": np.argsort((morton3(q >> 5) << np.uint64(3)) | octant, kind="stable"),
    }
    print(f"{n} secondary rays")
    for name, idx in orders.items():
        ms = r.bench_trace(o2[idx], d2[idx], shadow=False, repeat=5)
        print(f"{name:45s} {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrays/s", flush=True)
    r.destroy()


if __name__ == "__main__":
    main()
