"""Leaf size against traversal cost on the atrium: builds the tree with min_leaf 2, 3, 4 and 6 and prints nodes, SAH cost, node visits and triangle tests per ray and per-kernel time (GPU box)."""
import sys, time
sys.path.insert(0,'/root/repo')
import ray_tracer_amd as rt
from ray_tracer_amd import scenes, api
for ml in (2,3,4,6):
    sc = scenes.atrium().build_bvh(min_leaf=ml).build_lights()
    r = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_COUNT_TRAVERSAL)
    r.render(8); c = r.counters(); r.destroy()
    r = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_KERNEL_TIMING)
    r.render(8); r.reset_accumulator(); r.render(16); t = r.last_timer_perf(); r.destroy()
    print("min_leaf", ml, "nodes", len(sc.bvh), "sah %.1f" % sc.bvh_stats["sah_cost"], "depth", sc.bvh_stats["max_depth"], "pops/ray %.1f tris/ray %.2f" % (c["extend_node_pops"]/c["extend_rays"], c["extend_tri_tests"]/c["extend_rays"]),
          "extend %.3f connect %.3f shade %.3f total %.3f ms/sample" % (t["extend_ms"]/16, t["connect_ms"]/16, t["shade_ms"]/16, t["total_ms"]/16), flush=True)
