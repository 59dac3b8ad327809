#!/bin/bash
# One rocprofv3 counter pass over a short bench.py run; prints the per-kernel sums. Usage: tools/pmc_once.sh <tag> "<counters>" [bench args...]
set -u
TAG=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $R
timeout -s KILL 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o bench -- python3 $R/bench.py --steps ${PMC_STEPS:-8} --warmup ${PMC_STEPS:-8} --no-cpu-baseline --no-roofline --no-eighth "$@" > $OUT/bench.json 2> $OUT/err.txt || { echo "pass failed"; tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv, collections, glob
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("rtdev::", "").replace("void ", "")
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k in ("k_extend", "k_connect", "k_shade"):
    if k in acc: print("$TAG", k, {a: "%.4g" % b for a, b in sorted(acc[k].items())})
PY
