#!/bin/bash
# Usage: tools/lanes_bench.sh "<lanes list>" [bench args] -- sweeps the number of concurrent passes.
LIST=${1:-"1 2 3 4"}; shift || true
mkdir -p gpurun_out
for L in $LIST; do
  RT_HIP_LANES=$L timeout -k 10 300 python -u bench.py --steps 32 --warmup 4 --no-cpu-baseline "$@" > gpurun_out/lanes_$L.json 2> gpurun_out/lanes_$L.err || { echo "lanes $L failed"; tail -3 gpurun_out/lanes_$L.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/lanes_$L.json')); print('lanes $L: Mrays/s %.1f samples/s %.2f ms/step %.3f' % (d['value'], d['samples_per_sec'], d['ms_per_step']))"
done
