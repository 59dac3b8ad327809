#!/bin/bash
# Runs on the GPU box: GPU parity tests, then a short bench. Usage: tools/gpu_check.sh <tag> [bench args]
TAG=${1:-x}; shift || true
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests_$TAG.log 2>&1
rc=$?
tail -5 gpurun_out/gputests_$TAG.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; grep -n "Error\|assert\|FAILED" gpurun_out/gputests_$TAG.log | head -30; exit $rc; fi
timeout -k 10 400 python -u bench.py --steps 64 --warmup 8 --no-cpu-baseline "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { echo bench failed; tail -5 gpurun_out/bench_$TAG.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print("Mrays/s %.1f  samples/s %.2f  ms/step %.3f" % (d["value"], d["samples_per_sec"], d["ms_per_step"]))
print("kernel ms/sample", {k: round(v,3) for k,v in d.get("kernel_ms_per_sample",{}).items()})
r=d.get("roofline",{}); print("extend: %.0f Mrays/s, %.0f GB/s algorithmic (frac %.3f), pops/ray %.1f tris/ray %.2f" % (r.get("extend_mrays_per_s",0), r.get("achieved",0), r.get("frac",0), r.get("node_pops_per_ray",0), r.get("tri_tests_per_ray",0)))
print("connect", d.get("connect"))
PY
