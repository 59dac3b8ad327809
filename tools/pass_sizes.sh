#!/bin/bash
# GPU box: throughput by pass size with the default stream plan.
cd ${GRAFT_REPO_ROOT:-.}
for fpp in 1 2 4 8 16 0; do
  timeout -k 10 120 python bench.py --steps 24 --warmup 6 --frames-per-pass $fpp --no-cpu-baseline --no-eighth --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frames per pass $fpp: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)" || exit 1
done
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eighth --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('driver run (20 steps): %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)"
