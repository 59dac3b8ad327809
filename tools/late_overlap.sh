#!/bin/bash
# GPU box: in long passes, from which bounce the shadow rays go to the side stream (RT_HIP_OVERLAP_FROM) and with what grid the
# overlapping launches run (RT_HIP_TRACE_BLOCKS_LATE). Usage: tools/late_overlap.sh "<steps>" "<from list>" "<late blocks list>" [bench args]
STEPS=${1:-"20 64"}; FROM=${2:-"8 3 4 5"}; LATE=${3:-"8 4"}; shift 3 || true
for S in $STEPS; do for rep in 1 2; do for F in $FROM; do for L in $LATE; do
  RT_HIP_OVERLAP_FROM=$F RT_HIP_TRACE_BLOCKS_LATE=$L timeout -k 10 300 python -u bench.py --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $S side stream from bounce $F, grid $L: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done; done; done; done
