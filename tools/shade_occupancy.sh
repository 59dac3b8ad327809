#!/bin/bash
# GPU box: shade kernel time for library variants built with RT_SHADE_WAVES=w (register budget for w waves per SIMD),
# each at several shade grid sizes (256-thread workgroups per CU).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for w in base sw4 sw5 sw6 sw8; do
  for sb in 2 3 4 6 8; do
    RT_HIP_LIB=$PWD/variants/librt_hip_$w.so RT_HIP_SHADE_BLOCKS=$sb timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eighth 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_sample']; print('$w shade blocks $sb: %.0f Mrays/s  %.3f ms/step  shade %.3f extend %.3f connect %.3f' % (d['value'], d['ms_per_step'], k['shade'], k['extend'], k['connect']), flush=True)" || exit 1
  done
done
