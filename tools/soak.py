"""Soak run on the GPU box: 1024 spp on three scenes through repeated rt_render calls; prints time, sample count, NaN count and mean."""
import sys, time, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import ray_tracer_amd as rt
from ray_tracer_amd import scenes
for name, kw, b in (("atrium", dict(width=1920, height=1080), 8), ("glass_blob", dict(width=1920, height=1080), 24), ("car", dict(width=1280, height=720), 8)):
    sc = scenes.GENERATORS[name](**kw).finalize(spatial=True)
    r = rt.HipRenderer.create(sc, max_bounces=b)
    t0 = time.perf_counter()
    for _ in range(8):
        r.render(128)
    m = r.read_accum()
    dt = time.perf_counter() - t0
    print(name, "1024 spp in %.2f s" % dt, "sample_count", r.sample_count(), "nan", int(np.isnan(m).sum()), "mean", m[..., :3].mean(axis=(0, 1)))
    r.destroy()
