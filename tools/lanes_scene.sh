#!/bin/bash
# GPU box: one lane against two for a scene at the driver's call size. Usage: lanes_scene.sh <scene> [steps]
cd ${GRAFT_REPO_ROOT:-.}
S=${2:-20}
for ln in 1 2; do
  RT_HIP_LANES=$ln timeout -k 10 200 python bench.py --scene $1 --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 steps $S lanes $ln: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)" || exit 1
done
timeout -k 10 200 python bench.py --scene $1 --steps $S --warmup 5 --no-cpu-baseline --no-eighth --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 steps $S default plan: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)"
