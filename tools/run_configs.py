#!/usr/bin/env python3
"""Renders the five BASELINE.json configurations (synthetic stand-ins) on GPU 0 and prints one JSON line each:
ms per sample, Mrays/s, paths/s. Optionally writes small tone-mapped previews."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import scenes  # noqa: E402

CONFIGS = [
    ("cornell", dict(width=256, height=256), 4, 16),
    ("atrium", dict(width=1920, height=1080), 8, 64),
    ("glass_blob", dict(width=1920, height=1080), 8, 64),
    ("glass_blob", dict(width=1920, height=1080), 24, 64),
    ("car", dict(width=1920, height=1080), 8, 64),
    ("interior", dict(width=3840, height=2160), 8, 16),
]


def main(out_dir=None):
    for name, kw, bounces, spp in CONFIGS:
        t0 = time.time()
        sc = scenes.GENERATORS[name](**kw).finalize(spatial=True)      # the policy Scene::build_bvh uses upstream
        t_prep = time.time() - t0
        r = rt.HipRenderer.create(sc, max_bounces=bounces)
        r.render(min(4, spp))
        r.reset_accumulator()
        r.reset_counters()
        t0 = time.perf_counter()
        r.render(spp)
        dt = time.perf_counter() - t0
        c = r.counters()
        mean = r.read_accum()
        rays = c["extend_rays"] + c["shadow_rays"]
        rec = {"scene": sc.name, "res": f"{kw['width']}x{kw['height']}", "bounces": bounces, "spp": spp, "triangles": len(sc.triangles),
               "ms_per_sample": dt * 1e3 / spp, "samples_per_s": spp / dt, "mrays_per_s": rays / dt / 1e6,
               "extend_rays_per_path": c["extend_rays"] / c["camera_rays"], "shadow_rays_per_path": c["shadow_rays"] / c["camera_rays"],
               "mean_rgb": [float(x) for x in mean[..., :3].mean(axis=(0, 1))], "nan_pixels": int(np.isnan(mean).any(axis=-1).sum()),
               "scene_prep_s": round(t_prep, 2), "bvh_build_ms": sc.bvh_stats["build_ms"]}
        print(json.dumps(rec), flush=True)
        if out_dir:
            from PIL import Image
            fb = r.read_framebuffer()
            im = Image.fromarray(fb, "RGBA").convert("RGB")
            im.thumbnail((480, 270))
            im.save(Path(out_dir) / f"preview_{sc.name}_{bounces}b.jpg", quality=80)
        r.destroy()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
