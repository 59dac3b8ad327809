#!/bin/bash
# GPU box: whole-job rate for 1 and 2 lanes at several call lengths.
cd ${GRAFT_REPO_ROOT:-.}
for lanes in 1 2; do for steps in 20 64 128; do
  RT_HIP_LANES=$lanes timeout -k 10 300 python bench.py --steps $steps --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes $lanes steps $steps: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done; done
