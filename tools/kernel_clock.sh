#!/bin/bash
# GPU box: effective shader clock per kernel = GRBM_GUI_ACTIVE cycles / traced duration, from one rocprofv3 run that collects the
# counter together with the kernel trace. Usage: tools/kernel_clock.sh <tag> [bench args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/clk_$TAG
mkdir -p $OUT
cd $R
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -o bench -- python3 $R/bench.py --steps ${PMC_STEPS:-4} --warmup ${PMC_STEPS:-4} --no-cpu-baseline --no-roofline --no-eighth "$@" > $OUT/bench.json 2> $OUT/err.txt || { echo "pass failed"; tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv, collections, glob
tr = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
cc = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(tr[0])):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
    name, ns = dur[r["Dispatch_Id"]]
    k = name.split("(")[0].split("<")[0].replace("rtdev::", "").replace("void ", "")
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += ns
for k in ("k_extend", "k_shade", "k_connect", "k_raygen"):
    if k in acc:
        c, ns = acc[k]
        print("$TAG %-10s %8.2f ms traced   %.3e cycles (sum over XCDs)   %.2f GHz if the counter sums 8 XCDs" % (k, ns / 1e6, c, c / 8 / ns))
PY
