#!/bin/bash
# GPU box: size of the persistent traversal grids (RT_HIP_TRACE_BLOCKS, 256-thread units per compute unit: 8 = two 1024-thread
# workgroups per CU = every wave slot) by call size. Usage: tools/trace_grid_steps.sh "<steps list>" "<blocks list>" [bench args]
STEPS=${1:-"1 2 4 8 20"}; TBS=${2:-"4 6 8"}; shift 2 || true
for S in $STEPS; do for rep in 1 2; do for TB in $TBS; do
  RT_HIP_TRACE_BLOCKS=$TB timeout -k 10 300 python -u bench.py --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $S trace blocks $TB: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done; done; done
