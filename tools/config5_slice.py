"""One GPU's share of BASELINE configuration 5 (interior, 3840x2160, 4096 spp over 8 GPUs, 8 bounces): rank r of 8 renders
global frames r, r+8, ... (512 of them) and would then take part in one all-reduce of the 133 MB accumulation buffer.
Prints time, rate, NaN count and the mean. With `tiles` the rank takes the other partition instead: its 8x8 tiles (one in
eight) of all 4096 frames. usage: config5_slice.py [rank] [tiles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = scenes.interior(width=3840, height=2160).finalize(spatial=True)
r = rt.HipRenderer.create(sc, max_bounces=8)
tiles = len(sys.argv) > 2 and sys.argv[2] == "tiles"
if tiles:
    r.set_tile_slice(rank, 8)
else:
    r.set_frame_slice(rank, 8)
per_call, frames = (512, 4096) if tiles else (64, 512)
rep = r.tune_resident_set()
r.render(7); r.reset_accumulator(); r.reset_counters()
t0 = time.perf_counter()
for _ in range(8):
    r.dispatch_compute(per_call)
r.frame_complete()
dt = time.perf_counter() - t0
c = r.counters()
s = r.read_sum()
owned = float((s[..., 3] > 0).mean())
print(f"rank {rank} of 8 ({'tile' if tiles else 'sample'} slice): {frames} samples of {owned:.3f} of 3840x2160 in {dt:.2f} s = {frames * owned / dt:.1f} whole-frame samples/s, {(c['extend_rays'] + c['shadow_rays']) / dt / 1e6:.0f} Mrays/s; "
      f"sample count {r.sample_count()}, non-finite sums {int((~np.isfinite(s)).sum())}, mean radiance {s[..., :3].sum(axis=(0, 1)) / s[..., 3].sum()}", flush=True)
print(f"resident share {rep['resident_share_before'][1]:.2f} -> {rep['resident_share_after'][1]:.2f}; 8 such ranks: {8 * frames * owned / dt:.0f} samples/s before the reduce "
      f"(133 MB all-reduce per reporting interval)", flush=True)
