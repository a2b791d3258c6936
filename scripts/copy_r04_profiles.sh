#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): the round-4 evidence files produced by scripts/gpu_final_profiles_r04.sh
cd "$(dirname "$0")/.." || exit 1
G=gpurun_out; P=profiles
cp $G/r04final/r04_bench_line_default.json $P/ 2>/dev/null
cp $G/r04_default/r04_default_kernel_stats.csv $P/r04_default_kernel_stats.csv 2>/dev/null
cp $G/r04_default/r04_default_pmc_summary.json $P/r04_pmc_summary_default.json 2>/dev/null
cp $G/r04_default/bench_line_under_rocprof.json $P/r04_default_bench_line_under_rocprof.json 2>/dev/null
cp $G/r04_scan_sq/sq_summary.json $P/r04_scan_sq_counters.json 2>/dev/null
cp $G/r04final/r04_125M_stream_bench_line.json $P/ 2>/dev/null
cp $G/r04_125M_stream/r04_125M_stream_kernel_stats.csv $P/ 2>/dev/null
cp $G/r04_125M_stream/r04_125M_stream_pmc_summary.json $P/r04_pmc_summary_125M_stream.json 2>/dev/null
cp $G/r04_125M_stream/bench_line_under_rocprof.json $P/r04_125M_stream_bench_line_under_rocprof.json 2>/dev/null
cp $G/r04_sq_strand1/sq_summary.json $P/r04_sq_strand1_counters.json 2>/dev/null
cp $G/r04final/r04_m16_top1000_bench_line.json $G/r04final/r04_top1000_bench_line.json $G/r04final/r04_top2048_bench_line.json $P/ 2>/dev/null
cp $G/r04_m16_top1000/r04_m16_top1000_kernel_stats.csv $P/ 2>/dev/null
cp $G/r04_m16_top1000/r04_m16_top1000_pmc_summary.json $P/r04_pmc_summary_m16_top1000.json 2>/dev/null
cp $G/r04final/r04_pytest_gpu.log $G/r04final/r04_fuzz.log $G/r04final/r04_fuzz_big.log $P/ 2>/dev/null
cp $G/s1_skip.txt $P/r04_strand1_parts_left_out.txt 2>/dev/null
cp $G/scaling/inputs.txt $P/r04b_scaling_inputs.txt 2>/dev/null   # (r04_scaling_inputs.txt: the first half of the round)
cp $G/prio_ab.txt $P/r04_lane_gate_and_priority_ab.txt 2>/dev/null
ls -la $P | grep r04
