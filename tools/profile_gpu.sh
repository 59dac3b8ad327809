#!/bin/bash
# Runs on the GPU box (through gpurun): kernel traces (default two-lane run, and single-stream) + separate counter passes of
# bench.py. Counter passes and the second trace run single-stream (RT_HIP_LANES=1) so per-kernel numbers are those of a
# kernel running alone, which is what bench.py's roofline block reports.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
ARGS="--steps 64 --warmup 8 --no-cpu-baseline $*"      # two overlapped passes of 32 frames (the bench default)
ARGS1="--steps 32 --warmup 32 --no-cpu-baseline $*"    # single stream: one pass of 32 frames
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlapped -o bench -- python3 $R/bench.py $ARGS > $OUT/bench_trace_overlapped.json 2> $OUT/trace_overlapped.err || { echo "trace failed"; exit 1; }
export RT_HIP_LANES=1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py $ARGS1 > $OUT/bench_trace.json 2> $OUT/trace.err || { echo "single-stream trace failed"; exit 1; }
PM="--steps 32 --warmup 32 --no-cpu-baseline --no-roofline $*"   # whole passes of 32 frames, like the bench
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_a -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_a.json 2> $OUT/pmc_a.err || { echo "pmc_a failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc_b -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_b.json 2> $OUT/pmc_b.err || { echo "pmc_b failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_c -o bench -- python3 $R/bench.py $PM > $OUT/bench_pmc_c.json 2> $OUT/pmc_c.err || { echo "pmc_c failed"; exit 1; }
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
