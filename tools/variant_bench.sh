#!/bin/bash
# Runs bench.py once per library variant given on the command line (paths relative to the repo root).
mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename $lib .so)
  RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --steps ${VB_STEPS:-20} --warmup ${VB_WARMUP:-5} --no-cpu-baseline --no-eighth > gpurun_out/var_$name.json 2> gpurun_out/var_$name.err || { echo "$name failed"; tail -3 gpurun_out/var_$name.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/var_$name.json"))
k=d["kernel_ms_per_sample"]
print("%-28s Mrays/s %7.1f  ms/step %6.3f | extend %.3f shade %.3f connect %.3f | ext %.0f Mr/s" % ("$name", d["value"], d["ms_per_step"], k["extend"], k["shade"], k["connect"], d["roofline"]["extend_mrays_per_s"]))
PY
done
