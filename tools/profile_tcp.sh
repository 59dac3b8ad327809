#!/bin/bash
# Vector-memory-pipeline counters of the traversal kernels (single-stream so kernels do not overlap). Usage: tools/profile_tcp.sh <tag>
set -u
TAG=${1:-tcp}; shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
export RT_HIP_LANES=1
PM="--steps 8 --warmup 8 --no-cpu-baseline --no-roofline $*"
timeout -s KILL 240 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_d -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_d.json 2> $OUT/pmc_d.err || { echo "pmc_d failed"; tail -5 $OUT/pmc_d.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_e -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_e.json 2> $OUT/pmc_e.err || { echo "pmc_e failed"; tail -5 $OUT/pmc_e.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_f -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_f.json 2> $OUT/pmc_f.err || { echo "pmc_f failed"; tail -5 $OUT/pmc_f.err; exit 1; }
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A12 "^k_extend:\|^k_connect:" $OUT/summary.txt
