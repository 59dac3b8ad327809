#!/usr/bin/env python3
"""Does running independent frame groups on separate streams overlap the VALU-bound traversal of one group with
the HBM-bound shading of the other? Renders the same 64 samples with 1, 2, 3 and 4 concurrent contexts."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import scenes  # noqa: E402


def run(sc, n_ctx, fpp, spp=64, bounces=8):
    ctxs = []
    for i in range(n_ctx):
        r = rt.HipRenderer.create(sc, max_bounces=bounces, frames_per_pass=fpp)
        r.set_frame_slice(i, n_ctx)
        ctxs.append(r)
    per = spp // n_ctx
    for r in ctxs:
        r.dispatch_compute(fpp)
    for r in ctxs:
        r.frame_complete()
        r.reset_accumulator()
    t0 = time.perf_counter()
    for r in ctxs:
        r.dispatch_compute(per)
    for r in ctxs:
        r.frame_complete()
    dt = time.perf_counter() - t0
    for r in ctxs:
        r.destroy()
    return {"contexts": n_ctx, "frames_per_pass": fpp, "spp": per * n_ctx, "ms_per_sample": dt * 1e3 / (per * n_ctx)}


def main():
    sc = scenes.atrium(width=1920, height=1080).finalize()
    for n_ctx, fpp in [(1, 8), (1, 4), (2, 8), (2, 4), (3, 4), (4, 4), (4, 2), (2, 2)]:
        print(json.dumps(run(sc, n_ctx, fpp)), flush=True)


if __name__ == "__main__":
    main()
