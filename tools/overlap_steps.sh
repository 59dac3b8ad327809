#!/bin/bash
# GPU box: shadow rays on the side stream (1) or on the pass's own stream (0), by call size. Usage: tools/overlap_steps.sh "<steps list>" [bench args]
STEPS=${1:-"1 4 8 20 64"}; shift || true
cd ${GRAFT_REPO_ROOT:-.}
for S in $STEPS; do for O in 1 0; do for rep in 1 2; do
  RT_HIP_OVERLAP_CONNECT=$O timeout -k 10 300 python -u bench.py --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $S overlap $O: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done; done; done
