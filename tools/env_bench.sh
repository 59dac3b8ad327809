#!/bin/bash
# Usage: tools/env_bench.sh "VAR=a,VAR2=b" "VAR=c" ...  -- one short bench per environment setting (comma-separated assignments).
mkdir -p gpurun_out
for SPEC in "$@"; do
  TAG=$(echo "$SPEC" | tr -c 'A-Za-z0-9=\n' '_')
  ( IFS=','; for kv in $SPEC; do export "$kv"; done
    timeout -k 10 300 python -u bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/env_$TAG.json 2> gpurun_out/env_$TAG.err ) || { echo "$SPEC failed"; tail -3 gpurun_out/env_$TAG.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/env_$TAG.json')); print('$SPEC: Mrays/s %.1f samples/s %.2f ms/step %.3f' % (d['value'], d['samples_per_sec'], d['ms_per_step']))"
done
