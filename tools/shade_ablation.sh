#!/bin/bash
# GPU box: duration of the bounce-0 shade launch of every pass, for library variants built with RT_SHADE_ABLATE
# (tools/build_variant.sh). Only bounce 0 is comparable across variants: later bounces depend on what the ablated kernel
# wrote. One lane, one stream, 8-frame passes of the 1080p atrium. Usage: shade_ablation.sh <lib>...
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for lib in "$@"; do
  name=$(basename $lib .so)
  OUT=$R/gpurun_out/abl_$name
  mkdir -p $OUT
  RT_HIP_LIB=$R/$lib RT_HIP_LANES=1 RT_HIP_OVERLAP_CONNECT=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/bench.py --steps 24 --warmup 8 --frames-per-pass 8 --no-cpu-baseline --no-roofline --no-eighth > $OUT/bench.json 2> $OUT/err.txt || { echo "$name failed"; tail -5 $OUT/err.txt; exit 1; }
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first, after_raygen = [], False
for r in rows:
    k = r["Kernel_Name"]
    if "k_raygen" in k: after_raygen = True
    elif "k_shade" in k and after_raygen:
        first.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6); after_raygen = False
print("%-24s bounce-0 shade, ms per pass of 8 frames: %s" % ("$name", " ".join("%.3f" % x for x in first)))
PY
done
