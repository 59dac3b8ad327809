#!/bin/bash
# A/B on the GPU box: long passes cut in two and run on two lanes (RT_HIP_CHEAP_NS=10: always) or not (0: never).
# Usage: tools/cheap_sweep.sh <scene> [<scene> ...]
mkdir -p gpurun_out
for sc in "$@"; do for ns in 0 10 0 10; do for st in 20 64; do
RT_HIP_CHEAP_NS=$ns timeout -k 10 200 python -u bench.py --scene $sc --steps $st --warmup 8 --no-cpu-baseline --no-eighth --no-roofline > gpurun_out/cheap_${sc}_${ns}_$st.json 2>gpurun_out/cheap.err || { tail -3 gpurun_out/cheap.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/cheap_${sc}_${ns}_$st.json')); print('$sc cheap_ns $ns steps $st', round(d['value'],1), round(d['ms_per_step'],4))"
done; done; done
