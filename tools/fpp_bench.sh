#!/bin/bash
for f in "$@"; do
  timeout -k 10 300 python -u bench.py --steps 16 --warmup 4 --no-cpu-baseline --frames-per-pass $f > gpurun_out/fpp_$f.json 2> gpurun_out/fpp_$f.err || { echo "fpp $f failed"; tail -3 gpurun_out/fpp_$f.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/fpp_$f.json"))
k=d["kernel_ms_per_sample"]
print("fpp %-3s Mrays/s %7.1f  ms/step %6.3f | extend %.3f shade %.3f connect %.3f | ext %.0f Mr/s" % ("$f", d["value"], d["ms_per_step"], k["extend"], k["shade"], k["connect"], d["roofline"]["extend_mrays_per_s"]))
PY
done
