#!/bin/bash
# GPU box: throughput by pass size for each library variant given (names under variants/).
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  for fpp in 1 4 0; do
    RT_HIP_LIB=$PWD/variants/librt_hip_$v.so timeout -k 10 120 python bench.py --steps 24 --warmup 6 --frames-per-pass $fpp --no-cpu-baseline --no-eighth --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v frames per pass $fpp: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)" || exit 1
  done
done
