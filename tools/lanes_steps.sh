#!/bin/bash
# GPU box: lanes (concurrent passes) x call size. Usage: tools/lanes_steps.sh "<lanes list>" "<steps list>" [bench args]
LANES=${1:-"1 2"}; STEPS=${2:-"20 64"}; shift 2 || true
cd ${GRAFT_REPO_ROOT:-.}
for S in $STEPS; do for L in $LANES; do for rep in 1 2; do
  RT_HIP_LANES=$L timeout -k 10 300 python -u bench.py --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $S lanes $L: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done; done; done
