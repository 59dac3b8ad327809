#!/bin/bash
# GPU box: largest pass whose shadow rays go to the side stream (RT_HIP_OVERLAP_SLOTS, paths), by call size.
# Usage: tools/overlap_slots.sh "<steps list>" "<slots list>" [bench args]
STEPS=${1:-"8 12 20"}; SLOTS=${2:-"12582912 25165824 50331648 1073741824"}; shift 2 || true
for S in $STEPS; do for rep in 1 2; do for O in $SLOTS; do
  RT_HIP_OVERLAP_SLOTS=$O timeout -k 10 300 python -u bench.py --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $S side stream up to $O paths: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done; done; done
