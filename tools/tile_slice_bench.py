"""Throughput of one context rendering a tile slice (its 8x8 tiles of every frame) against the whole frame, at the same
number of paths per call (GPU box). usage: tile_slice_bench.py [scene] [width] [height]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
from ray_tracer_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
sc = scenes.GENERATORS[name](width=w, height=h).finalize(spatial=True)
for stride in (1, 2, 4, 8):
    r = rt.HipRenderer.create(sc, max_bounces=8)
    r.set_tile_slice(0, stride)
    frames = 20 * stride                       # the same number of paths per call
    r.render(frames); r.frame_complete()
    r.reset_accumulator(); r.reset_counters(); r.frame_complete()
    t0 = time.perf_counter()
    r.dispatch_compute(frames); r.frame_complete()
    dt = time.perf_counter() - t0
    c = r.counters()
    print(f"{sc.name} {w}x{h} tiles 0 of {stride}: {frames} frames in {dt * 1e3:.1f} ms, {(c['extend_rays'] + c['shadow_rays']) / dt / 1e6:.0f} Mrays/s, "
          f"{c['camera_rays'] / dt / 1e6:.1f} Mpaths/s", flush=True)
    r.destroy()
