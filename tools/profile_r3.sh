#!/bin/bash
# Runs on the GPU box (through gpurun). Separate rocprofv3 runs, each of ONE kind of context, at the launch size the driver
# times (--steps 20: one pass of 20 frames per call):
#   trace_timing      kernel trace of the single-stream context that bench.py times its kernels in (kernels run alone)
#   trace_overlapped  kernel trace of the context the timed region uses (shadow rays beside the next closest-hit search)
#   pmc_a, pmc_b, pmc_c   three counter passes of the single-stream context
#   clock             GRBM_GUI_ACTIVE beside the kernel trace: effective shader clock per kernel
# Usage: tools/profile_r3.sh <tag> [bench args, e.g. --scene interior --width 3840 --height 2160]
set -u
TAG=${1:-r3}; shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
STEPS=${PROF_STEPS:-20}
T="--profile-context timing --steps $STEPS $*"
O="--profile-context overlapped --steps $STEPS --warmup $STEPS --no-cpu-baseline --no-eighth $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_timing -o bench -- python3 $R/bench.py $T > $OUT/bench_trace_timing.json 2> $OUT/trace_timing.err || { echo "trace_timing failed"; tail -5 $OUT/trace_timing.err; exit 1; }
echo trace_timing done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlapped -o bench -- python3 $R/bench.py $O > $OUT/bench_trace_overlapped.json 2> $OUT/trace_overlapped.err || { echo "trace_overlapped failed"; tail -5 $OUT/trace_overlapped.err; exit 1; }
echo trace_overlapped done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_a -o bench -- python3 $R/bench.py $T > $OUT/bench_pmc_a.json 2> $OUT/pmc_a.err || { echo "pmc_a failed"; tail -5 $OUT/pmc_a.err; exit 1; }
echo pmc_a done
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc_b -o bench -- python3 $R/bench.py $T > $OUT/bench_pmc_b.json 2> $OUT/pmc_b.err || { echo "pmc_b failed"; tail -5 $OUT/pmc_b.err; exit 1; }
echo pmc_b done
timeout -s KILL 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_c -o bench -- python3 $R/bench.py $T > $OUT/bench_pmc_c.json 2> $OUT/pmc_c.err || { echo "pmc_c failed"; tail -5 $OUT/pmc_c.err; exit 1; }
echo pmc_c done
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/clock -o bench -- python3 $R/bench.py $T > $OUT/bench_clock.json 2> $OUT/clock.err || { echo "clock failed"; tail -5 $OUT/clock.err; exit 1; }
echo clock done
python3 $R/tools/roofline_counters.py $OUT $OUT/roofline_counters.json && cat $OUT/roofline_counters.json
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
# keep what travels back small: the per-dispatch CSVs of the counter passes are large
find $OUT -name "*counter_collection.csv" -size +8M -delete
