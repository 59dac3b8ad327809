#!/bin/bash
# Copies what tools/profile_r3.sh left under gpurun_out/prof_r3a (atrium) and gpurun_out/prof_r3i (interior) to the names
# bench.py reads under profiles/.
set -e
P=gpurun_out/prof_r3a
cp $P/roofline_counters.json profiles/r3_roofline_counters.json
cp $P/summary.txt profiles/r3_summary.txt
cp $P/trace_timing/bench_kernel_stats.csv profiles/r3_kernel_stats_timing_context.csv
cp $P/trace_overlapped/bench_kernel_stats.csv profiles/r3_kernel_stats_overlapped_context.csv
cp $P/bench_trace_timing.json profiles/r3_bench_under_trace_timing_context.json
P=gpurun_out/prof_r3i
cp $P/roofline_counters.json profiles/r3_roofline_counters_interior.json
cp $P/summary.txt profiles/r3_summary_interior.txt
cp $P/trace_timing/bench_kernel_stats.csv profiles/r3_kernel_stats_interior_timing_context.csv
cp $P/trace_overlapped/bench_kernel_stats.csv profiles/r3_kernel_stats_interior_overlapped_context.csv
