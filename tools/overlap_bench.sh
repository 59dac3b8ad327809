#!/bin/bash
# Usage: tools/overlap_bench.sh <lib.so> "ENV=a,ENV2=b" ...  -- bench.py (two overlapped lanes) per environment setting for one library variant
LIB=$1; shift
mkdir -p gpurun_out
for SPEC in "$@"; do
  TAG=$(basename $LIB .so)_$(echo "$SPEC" | tr -c 'A-Za-z0-9=\n' '_')
  ( IFS=','; for kv in $SPEC; do export "$kv"; done
    RT_HIP_LIB=$PWD/$LIB timeout -k 10 300 python -u bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/ov_$TAG.json 2> gpurun_out/ov_$TAG.err ) || { echo "$SPEC failed"; tail -3 gpurun_out/ov_$TAG.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ov_$TAG.json')); print('$(basename $LIB .so) $SPEC: Mrays/s %.1f samples/s %.2f ms/step %.3f' % (d['value'], d['samples_per_sec'], d['ms_per_step']))"
done
