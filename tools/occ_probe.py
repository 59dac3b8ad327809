"""Lane utilisation of the closest-hit kernel from its counters, for a library built with -DRT_OCCUPANCY_PROBE (the leaf and triangle counters then count wave-level node steps and leaf phases): lanes per node step, node steps per leaf phase (GPU box)."""
import sys; sys.path.insert(0,'.')
import ray_tracer_amd as rt
from ray_tracer_amd import scenes, api
sc = scenes.atrium().finalize(spatial=True)
r = rt.HipRenderer.create(sc, max_bounces=8, flags=api.RT_COUNT_TRAVERSAL)
r.render(20); c = r.counters()
print(c)
print("extend: lanes per wave node-step %.1f ; node steps per ray %.2f; wave-steps %d; leaf phases %d; node steps per leaf phase %.2f" % (c["extend_node_pops"]/c["extend_leaf_pops"], c["extend_node_pops"]/c["extend_rays"], c["extend_leaf_pops"], c["extend_tri_tests"], c["extend_leaf_pops"]/c["extend_tri_tests"]))
