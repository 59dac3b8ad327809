#!/bin/bash
# GPU box: the driver's bench run, N times alternating between one and two lanes. Usage: lanes_repeat.sh N [bench args]
cd ${GRAFT_REPO_ROOT:-.}
N=$1; shift
for i in $(seq 1 $N); do
  for ln in 1 2; do
    RT_HIP_LANES=$ln timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eighth --no-roofline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes $ln run $i: %.0f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)" || exit 1
  done
done
