#!/bin/bash
# GPU box: shadow rays on a side stream (RT_HIP_OVERLAP_CONNECT=1) against everything on one stream (=0), at several pass sizes.
cd ${GRAFT_REPO_ROOT:-.}
for fpp in 0 4 1; do
  for ov in 0 1; do
    RT_HIP_OVERLAP_CONNECT=$ov timeout -k 10 120 python bench.py --steps 20 --warmup 5 --frames-per-pass $fpp --no-cpu-baseline --no-eighth --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frames per pass $fpp overlap $ov: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)" || exit 1
  done
done
