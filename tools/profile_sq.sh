#!/bin/bash
# One counter pass over bench.py for the SQ-side picture of the traversal kernels: vector-ALU occupancy of the SIMDs,
# lane utilisation, LDS activity. Usage: tools/profile_sq.sh <tag> [library.so]
set -u
TAG=${1:-sq}; LIB=${2:-}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
[ -n "$LIB" ] && export RT_HIP_LIB=$R/$LIB
export RT_HIP_LANES=1
PM="--steps 8 --warmup 8 --no-cpu-baseline --no-roofline"
timeout -s KILL 240 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || { echo "pmc_sq failed"; tail -5 $OUT/pmc_sq.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_sq2.json 2> $OUT/pmc_sq2.err || { echo "pmc_sq2 failed"; tail -5 $OUT/pmc_sq2.err; exit 1; }
python3 - <<PY
import csv, collections, glob
for d in ("pmc_sq", "pmc_sq2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print("no csv for", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("rtdev::", "").replace("void ", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k in ("k_extend", "k_connect", "k_shade"):
        c = acc.get(k)
        if not c: continue
        print(k, {a: "%.4g" % b for a, b in sorted(c.items())})
        if "SQ_ACTIVE_INST_VALU" in c:
            simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024     # GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs
            print("   VALU busy (4 cycles per wave-instruction): %.3f   lane utilisation %.3f   wait_inst_any/wave_cycles %.3f   waves resident/SIMD %.2f" % (
                c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64), c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                c["SQ_WAVE_CYCLES"] * 4 / simd_cycles))
PY
