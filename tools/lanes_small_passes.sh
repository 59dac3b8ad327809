#!/bin/bash
# GPU box: concurrent passes (RT_HIP_LANES, every pass alternates) by pass size. Usage: lanes_small_passes.sh "<fpp list>" "<lanes list>" [steps]
cd ${GRAFT_REPO_ROOT:-.}
STEPS=${3:-24}
for fpp in ${1:-1 2 4 8}; do
  for ln in ${2:-1 2 3}; do
    RT_HIP_LANES=$ln timeout -k 10 120 python bench.py --steps $STEPS --warmup 6 --frames-per-pass $fpp --no-cpu-baseline --no-eighth --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frames per pass $fpp lanes $ln: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)" || exit 1
  done
done
