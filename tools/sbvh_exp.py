#!/usr/bin/env python3
"""Object-split vs spatial-split BVH on the benchmark scenes: build time, SAH cost, per-stage kernel time and
traversal work per ray. One JSON line per (scene, policy)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import api, scenes  # noqa: E402

CONFIGS = [("atrium", dict(width=1920, height=1080), 8, 16), ("car", dict(width=1920, height=1080), 8, 8),
           ("glass_blob", dict(width=1920, height=1080), 8, 8), ("interior", dict(width=3840, height=2160), 8, 4)]


def main(names):
    for name, kw, bounces, spp in CONFIGS:
        if names and name not in names:
            continue
        for spatial in (False, True):
            sc = scenes.GENERATORS[name](**kw).finalize(spatial=spatial)
            rec = {"scene": name, "spatial": spatial, "tris": len(sc.triangles), "sah": sc.bvh_stats["sah_cost"],
                   "build_ms": sc.bvh_stats["build_ms"], "dup": sc.bvh_stats["duplicated_refs"]}
            for flags, tag in ((api.RT_KERNEL_TIMING, "t"), (api.RT_COUNT_TRAVERSAL, "c")):
                r = rt.HipRenderer.create(sc, max_bounces=bounces, flags=flags)
                r.render(2)
                r.reset_accumulator()
                r.reset_counters()
                r.render(spp)
                r.frame_complete()
                if tag == "t":
                    t = r.last_timer_perf()
                    n = max(1, spp)
                    rec.update({k: round(t[k] / n, 4) for k in ("total_ms", "extend_ms", "shade_ms", "connect_ms")})
                else:
                    c = r.counters()
                    rec["ext_nodes_per_ray"] = round(c["extend_node_pops"] / max(1, c["extend_rays"]), 2)
                    rec["ext_tris_per_ray"] = round(c["extend_tri_tests"] / max(1, c["extend_rays"]), 2)
                    rec["sh_nodes_per_ray"] = round(c["shadow_node_pops"] / max(1, c["shadow_rays"]), 2)
                    rec["sh_tris_per_ray"] = round(c["shadow_tri_tests"] / max(1, c["shadow_rays"]), 2)
                r.destroy()
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
