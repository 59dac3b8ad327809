#!/bin/bash
# Usage: tools/split_bench.sh <steps> <frames per pass>...  -- default bench run with passes capped at that many frames
S=$1; shift
mkdir -p gpurun_out
for F in "$@"; do
  timeout -k 10 300 python -u bench.py --steps $S --warmup 4 --no-cpu-baseline --no-roofline --frames-per-pass $F > gpurun_out/split_$F.json 2> gpurun_out/split_$F.err || { echo "fpp $F failed"; tail -3 gpurun_out/split_$F.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/split_$F.json')); print('steps $S, passes of <= $F frames: Mrays/s %.1f samples/s %.2f ms/step %.3f' % (d['value'], d['samples_per_sec'], d['ms_per_step']))"
done
