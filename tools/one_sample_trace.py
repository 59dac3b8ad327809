"""Six one-frame passes of the 1080p atrium on one lane; run under `rocprofv3 --kernel-trace` with RT_HIP_OVERLAP_CONNECT=0 to
read the launch time of every bounce of a one-sample pass. (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes
sc = scenes.GENERATORS["atrium"](width=1920, height=1080).finalize(spatial=True)
if os.environ.get("ONE_SAMPLE_TRACE_DEFAULT", "0") == "0":
    os.environ["RT_HIP_LANES"] = "1"          # one lane; ONE_SAMPLE_TRACE_DEFAULT=1 traces the default stream plan instead
r = rt.HipRenderer.create(sc, max_bounces=8, frames_per_pass=1)
for _ in range(6):
    r.render(1); r.frame_complete()
r.destroy()
