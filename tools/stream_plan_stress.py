"""Stress the stream plan on the GPU box: the default plan (side stream for shadow rays, two lanes for short or cheap passes,
adaptive lane choice) against RT_SINGLE_STREAM, bit for bit, over many call patterns: call sizes from 1 to 40 samples, with
and without synchronisation between calls, camera moves and resets in between. Any race between the streams would show up as
a differing bit. usage: stream_plan_stress.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(7)
cases = [("atrium", dict(width=960, height=540, scale=0.25), 8), ("glass_blob", dict(width=640, height=360, scale=0.25), 12),
         ("car", dict(width=640, height=360, scale=0.1), 8), ("cornell", dict(width=256, height=256), 4), ("atrium", dict(width=1920, height=1080), 8),
         # long passes of one call side by side on two lanes (what scenes with cheap paths get), forced for a scene with lights
         ("cornell", dict(width=1920, height=1080), 4), ("atrium", dict(width=1920, height=1080, side_by_side=True), 8)]
bad = 0
t0 = time.time()
for name, kw, bounces in cases:
    kw = dict(kw)
    forced = kw.pop("side_by_side", False)
    sc = scenes.GENERATORS[name](**kw).finalize(spatial=True)
    if forced:
        os.environ["RT_HIP_CHEAP_NS"] = "10"              # read once, when the context is created
    a = rt.HipRenderer.create(sc, max_bounces=bounces)
    os.environ.pop("RT_HIP_CHEAP_NS", None)
    b = rt.HipRenderer.create(sc, max_bounces=bounces, flags=api.RT_SINGLE_STREAM)
    n_rounds = rounds if kw.get("width", 0) < 1920 else max(3, rounds // 6)
    for it in range(n_rounds):
        sizes = [1, 1, 2, 3, 5, 8, 13, 20, 40] if kw.get("width", 0) < 1920 else [1, 2, 5, 13, 20, 40, 64, 70]
        pattern = [int(x) for x in rng.choice(sizes, size=rng.integers(1, 6))]
        sync_between = bool(rng.integers(0, 2))
        if it % 5 == 4:                                   # move the camera: accumulation restarts, the lane plan is re-measured
            cam = sc.camera.copy()
            cam["location"][0][:3] += rng.normal(scale=0.05, size=3).astype(np.float32)
            a.set_camera(cam); b.set_camera(cam)
        else:
            a.reset_accumulator(); b.reset_accumulator()
        for n in pattern:
            a.render(n); b.render(n)
            if sync_between:
                a.frame_complete(); b.frame_complete()
        ia, ib = a.read_accum(), b.read_accum()
        same = np.array_equal(ia, ib, equal_nan=True) and a.sample_count() == b.sample_count() == sum(pattern)
        if not same:
            bad += 1
            print(f"MISMATCH {name} round {it} pattern {pattern} sync {sync_between}: {int((ia != ib).sum())} values differ", flush=True)
    print(f"{name} {kw}{' (long passes side by side)' if forced else ''}: {n_rounds} rounds done, mismatches so far {bad}, {time.time() - t0:.0f} s", flush=True)
    a.destroy(); b.destroy()
print("RESULT", "ok" if bad == 0 else f"{bad} mismatching rounds")
sys.exit(0 if bad == 0 else 1)
