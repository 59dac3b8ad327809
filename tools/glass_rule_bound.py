"""What the shadow rule 'closest hit is glass => unoccluded' costs: renders the scenes that hold glass-like materials as they are (closest-hit shadow rays) and with spec_trans capped at 0.5 (shadow rays may stop at the first hit), and prints visits per shadow ray and per-kernel times (GPU box)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import scenes, api
for name, kw, spp in (("interior", dict(width=3840, height=2160), 8), ("car", dict(width=1920, height=1080), 32), ("glass_blob", dict(width=1920, height=1080), 32)):
    for strip in (False, True):
        sc = scenes.GENERATORS[name](**kw)
        if strip:
            sc.materials["spec_trans"][:] = np.minimum(sc.materials["spec_trans"], 0.5)   # still transmissive in shading, not "glass" for shadows
        sc = sc.finalize(spatial=True)
        r = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_COUNT_TRAVERSAL | api.RT_KERNEL_TIMING)
        r.render(2); r.reset_accumulator(); r.reset_counters()
        t0 = time.perf_counter(); r.render(spp); dt = time.perf_counter() - t0
        c = r.counters(); t = r.last_timer_perf(); print(t) if False else None
        print(name, "no-glass-rule" if strip else "glass rule", "ms/sample %.3f" % (dt * 1e3 / spp), "shadow pops/ray %.2f tri/ray %.2f" % (c["shadow_node_pops"] / max(1, c["shadow_rays"]), c["shadow_tri_tests"] / max(1, c["shadow_rays"])),
              "timings", {k: round(v, 3) for k, v in t.items()}, flush=True)
        r.destroy()
