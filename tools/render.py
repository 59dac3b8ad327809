#!/usr/bin/env python3
"""Headless render: scene file (YAML + OBJ, glTF/GLB) or a named synthetic scene -> PNG, the job the reference's window
does interactively (`main.rs:166-168` takes the scene path; there is no headless mode upstream).

    python tools/render.py scenes/my_scene.yaml --spp 256 --out frame.png
    python tools/render.py synthetic:atrium --spp 64 --bounces 8 --out atrium.png
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import loaders, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene", help="path to .yaml/.yml, .gltf/.glb, or synthetic:<cornell|atrium|glass_blob|car|interior>")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=24, help="BOUNCES in main.comp:4")
    ap.add_argument("--out", default="frame.png")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--object-splits", action="store_true", help="build the BVH without spatial splits")
    a = ap.parse_args()

    t0 = time.perf_counter()
    spatial = not a.object_splits
    if a.scene.startswith("synthetic:"):
        kw = {k: v for k, v in (("width", a.width), ("height", a.height)) if v}
        sc = scenes.GENERATORS[a.scene.split(":", 1)[1]](**kw).finalize(spatial=spatial)
    elif a.scene.lower().endswith((".gltf", ".glb")):
        sc = loaders.from_gltf(a.scene, spatial=spatial, build_lights=True)
    else:
        sc = loaders.from_new(a.scene, spatial=spatial)
    t1 = time.perf_counter()
    r = rt.HipRenderer.create(sc, max_bounces=a.bounces, device=a.device, width=a.width, height=a.height)
    r.render(a.spp)
    t2 = time.perf_counter()
    out = loaders.save_frame(r, a.out)
    c = r.counters()
    print(f"{sc.name}: {len(sc.triangles)} triangles, {r.width}x{r.height}, {a.spp} spp in {t2 - t1:.3f} s "
          f"({(c['extend_rays'] + c['shadow_rays']) / (t2 - t1) / 1e6:.0f} Mrays/s incl. upload), scene prep {t1 - t0:.2f} s -> {out}")
    r.destroy()


if __name__ == "__main__":
    main()
