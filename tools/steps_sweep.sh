#!/bin/bash
# GPU box: the bench line by call size at the library's defaults. Usage: tools/steps_sweep.sh "<steps list>" [bench args]
STEPS=${1:-"1 2 3 4 6 8 12 20 64"}; shift || true
for S in $STEPS; do for rep in 1 2; do
  timeout -k 10 300 python -u bench.py --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $S: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done; done
