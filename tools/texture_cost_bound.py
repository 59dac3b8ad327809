"""What the 37 textures of the interior cost its shade kernel: the scene as it is and with every texture binding removed (GPU box)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import scenes, api
for strip in (False, True):
    sc = scenes.interior(width=3840, height=2160)
    if strip:
        for f in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emission_tex"):
            sc.materials[f][:] = 0xFFFFFFFF
    sc = sc.finalize(spatial=True)
    r = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_KERNEL_TIMING)
    r.render(2); r.reset_accumulator(); r.reset_counters()
    t0 = time.perf_counter(); r.render(8); dt = time.perf_counter() - t0
    c = r.counters(); t = r.last_timer_perf()
    print("no textures" if strip else "textures", "ms/sample %.3f" % (dt * 1e3 / 8), {k: round(v / 8, 3) for k, v in t.items() if k.endswith("_ms")}, "vertices/sample %.2fM" % (c["extend_rays"] / 8e6), flush=True)
    r.destroy()
