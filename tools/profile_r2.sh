#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + separate counter passes of bench.py at the launch size the driver
# times (--steps 20 --warmup 5: one pass of 20 frames per call). Usage: tools/profile_r2.sh <tag> [bench args...]
set -u
TAG=${1:-r2}; shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-eighth $*"
PM="--steps 20 --warmup 20 --no-cpu-baseline --no-roofline --no-eighth $*"      # warm-up of the same size: every counted launch is a 20-frame launch
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
echo trace done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_a -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_a.json 2> $OUT/pmc_a.err || { echo "pmc_a failed"; tail -5 $OUT/pmc_a.err; exit 1; }
echo pmc_a done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc_b -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_b.json 2> $OUT/pmc_b.err || { echo "pmc_b failed"; tail -5 $OUT/pmc_b.err; exit 1; }
echo pmc_b done
timeout -s KILL 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_c -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_c.json 2> $OUT/pmc_c.err || { echo "pmc_c failed"; tail -5 $OUT/pmc_c.err; exit 1; }
echo pmc_c done
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -80 $OUT/summary.txt
