#!/bin/bash
# GPU box: shade grid size (256-thread workgroups per CU) vs per-kernel time, for each library given (default: the shipped one).
cd ${GRAFT_REPO_ROOT:-.}
LIBS=${@:-ray-tracer_amd/librt_hip.so}
for lib in $LIBS; do
for sb in ${SHADE_BLOCKS:-2 3 4 6 8}; do
  RT_HIP_LIB=$PWD/$lib RT_HIP_SHADE_BLOCKS=$sb timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eighth 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_sample']; print('$(basename $lib .so) shade blocks $sb: %.0f Mrays/s  %.3f ms/step  shade %.3f extend %.3f connect %.3f' % (d['value'], d['ms_per_step'], k['shade'], k['extend'], k['connect']))" || exit 1
done
done
