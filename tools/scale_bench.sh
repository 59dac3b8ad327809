#!/bin/bash
# Usage: tools/scale_bench.sh "<scales>"  -- node-step throughput of extend as the scene (and its working set) shrinks.
mkdir -p gpurun_out
for S in $1; do
  timeout -k 10 300 python -u bench.py --steps 16 --warmup 2 --no-cpu-baseline --scale $S > gpurun_out/scale_$S.json 2> gpurun_out/scale_$S.err || { echo "scale $S failed"; tail -3 gpurun_out/scale_$S.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/scale_$S.json')); r=d['roofline']; k=d['kernel_ms_per_sample']
steps=r['extend_mrays_per_s']*r['node_pops_per_ray']/1e3
print('scale $S tris %d: extend %.3f ms, %.0f Mrays/s, %.1f pops/ray -> %.1f G node steps/s; total %.1f Mrays/s' % (d['config']['triangles'], k['extend'], r['extend_mrays_per_s'], r['node_pops_per_ray'], steps, d['value']))"
done
