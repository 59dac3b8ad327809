#!/bin/bash
# GPU box: per-launch durations of the three hot kernels for the last pass of a short bench run. Usage: trace_launches.sh <tag> [bench args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tr_$TAG
mkdir -p $OUT
cd $R
RT_HIP_OVERLAP_CONNECT=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python3 $R/bench.py --steps ${PMC_STEPS:-4} --warmup ${PMC_STEPS:-4} --no-cpu-baseline --no-roofline --no-eighth "$@" > $OUT/bench.json 2> $OUT/err.txt || { echo "trace failed"; tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("k_extend", "k_shade", "k_connect"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-24:]
line = {"k_extend": [], "k_shade": [], "k_connect": []}
for r in last:
    k = [k for k in line if k in r["Kernel_Name"]][0]
    line[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in line.items():
    print("$TAG %-10s ms per bounce: %s" % (k, " ".join("%7.2f" % x for x in v)))
PY
