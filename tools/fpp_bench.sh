#!/bin/bash
# Usage: tools/fpp_bench.sh <frames per pass>...   -- bench.py with passes of that many frames (steps = 8 passes)
mkdir -p gpurun_out
for F in "$@"; do
  S=$((F * 8))
  timeout -k 10 300 python -u bench.py --steps $S --warmup $F --no-cpu-baseline --no-roofline --frames-per-pass $F > gpurun_out/fpp_$F.json 2> gpurun_out/fpp_$F.err || { echo "fpp $F failed"; tail -3 gpurun_out/fpp_$F.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/fpp_$F.json')); print('frames/pass $F: Mrays/s %.1f samples/s %.2f ms/step %.3f' % (d['value'], d['samples_per_sec'], d['ms_per_step']))"
done
