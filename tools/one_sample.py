"""One sample per call (what an interactive host does) with the per-path kernel off and at several settings: wall time per
sample when the host waits for each call and when calls are queued back to back, and how many paths the kernel took over.
usage: one_sample.py [scene] [width height] [bounces]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
from ray_tracer_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
bounces = int(sys.argv[4]) if len(sys.argv) > 4 else 8
settings = os.environ.get("ONE_SAMPLE_SETTINGS", "0:3 65536:3 131072:3 196608:3 393216:3 786432:3 196608:2 196608:4 393216:2 1572864:1")
sc = scenes.GENERATORS[name](width=W, height=H).finalize(spatial=True)
N = 60
for s in settings.split():
    tmax, tfrom = s.split(":")
    os.environ["RT_HIP_TAIL_MAX"], os.environ["RT_HIP_TAIL_FROM"] = tmax, tfrom
    os.environ["RT_HIP_TAIL_SLOTS"] = str(1 << 30)          # offered whatever the size of the pass
    r = rt.HipRenderer.create(sc, max_bounces=bounces, frames_per_pass=1)
    r.render(8); r.frame_complete()
    r.reset_counters()
    t0 = time.perf_counter()
    for _ in range(N):
        r.render(1); r.frame_complete()
    sync_ms = (time.perf_counter() - t0) * 1e3 / N
    t0 = time.perf_counter()
    for _ in range(N):
        r.dispatch_compute(1)
    r.frame_complete()
    b2b_ms = (time.perf_counter() - t0) * 1e3 / N
    c = r.counters()
    r.destroy()
    rays = (c["extend_rays"] + c["shadow_rays"]) / (2 * N)
    print(f"{name} {W}x{H} {bounces} bounces, per-path kernel max {int(tmax):>8} from bounce {tfrom}: call+wait {sync_ms:.3f} ms, back to back {b2b_ms:.3f} ms "
          f"({rays / b2b_ms / 1e3:.0f} Mrays/s), {c['tail_paths'] / (2 * N):.0f} paths per sample finished by it", flush=True)
