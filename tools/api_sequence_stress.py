"""Random sequences of entry points without synchronisation in between (render, poll, camera moves, resets, resizes, scene
reloads, tile and frame slices, resident-set tuning, counter and image reads), on the GPU box. After every sequence the accumulated image must equal
what a fresh context produces for the final scene, camera, resolution and sample count: any ordering mistake between the
main stream, the lanes and their side streams shows up as a differing bit. usage: api_sequence_stress.py [rounds]"""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(23)
pool = [scenes.atrium(width=160, height=90, scale=0.05).finalize(spatial=True), scenes.car(width=160, height=90, scale=0.03).finalize(spatial=True),
        scenes.cornell(width=160, height=90).finalize(), scenes.glass_blob(width=160, height=90, scale=0.05).finalize(spatial=True)]
bad = 0
t0 = time.time()
for it in range(rounds):
    sc = pool[rng.integers(0, len(pool))]
    r = rt.HipRenderer.create(sc, max_bounces=5)
    cam = sc.camera.copy()
    w, h = 160, 90
    samples = 0
    tiles, frames = (0, 1), (0, 1)
    log = []
    for step in range(int(rng.integers(3, 14))):
        op = rng.choice(["render", "render", "render", "poll", "camera", "reset", "resize", "reload", "tune", "read", "counters", "tiles", "frames"])
        log.append(op)
        if op == "render":
            n = int(rng.choice([1, 1, 2, 3, 7, 20])); r.dispatch_compute(n); samples += n
        elif op == "poll":
            r.frame_done()
        elif op == "camera":
            cam = cam.copy(); cam["location"][0][:3] += rng.normal(scale=0.05, size=3).astype(np.float32); r.set_camera(cam); samples = 0
        elif op == "reset":
            r.reset_accumulator(); samples = 0
        elif op == "resize":
            w, h = int(rng.choice([96, 160, 200])), int(rng.choice([54, 90, 120])); r.handle_resize(w, h); samples = 0
        elif op == "reload":
            sc = pool[rng.integers(0, len(pool))]; r.upload_scene(sc); cam = sc.camera.copy(); samples = 0
        elif op == "tiles":
            st = int(rng.integers(1, 5)); tiles = (int(rng.integers(0, st)), st); r.set_tile_slice(*tiles); samples = 0
        elif op == "frames":
            st = int(rng.integers(1, 4)); frames = (int(rng.integers(0, 5)), st); r.set_frame_slice(*frames); samples = 0
        elif op == "tune":
            r.tune_resident_set()
        elif op == "read":
            r.read_accum()
        elif op == "counters":
            r.counters()
    got = r.read_sum()
    count = r.sample_count()
    r.destroy()
    ref = rt.HipRenderer.create(sc, max_bounces=5, flags=api.RT_SINGLE_STREAM, width=w, height=h)
    ref.set_camera(cam)
    ref.set_tile_slice(*tiles)
    ref.set_frame_slice(*frames)
    if samples:
        ref.render(samples)
    want = ref.read_sum()
    ref.destroy()
    same = count == samples and got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
    if not same:
        bad += 1
        print(f"MISMATCH round {it}: {log} samples {samples} count {count} shapes {got.shape} {want.shape}", flush=True)
    if it % 10 == 9:
        print(f"{it + 1} sequences, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("RESULT", "ok" if bad == 0 else f"{bad} mismatching sequences")
sys.exit(0 if bad == 0 else 1)
