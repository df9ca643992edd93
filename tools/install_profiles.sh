#!/bin/bash
# Copies the summaries collect_profiles.sh left in gpurun_out/$ROUND into profiles/$ROUND (default r6) (tracked) and refreshes profiles/pmc_traffic.json.
set -e
R=${ROUND:-r6}; S=gpurun_out/$R D=profiles/$R
mkdir -p $D/t5000
cp $S/bench.json $S/bench_detail.json $S/bench_lanes_traced.json $S/bench_serial_traced.json $S/pmc_fetch_size_per_kernel.csv $S/pmc_write_size_per_kernel.csv $S/pmc_traffic.json $S/pmc_traffic.txt $D/
cp $S/serial/serial_kernel_stats.csv $D/kernel_stats_serial.csv
cp $S/lanes/lanes_kernel_stats.csv $D/kernel_stats_lanes.csv
cp $S/bench_t5000.json $D/t5000/bench_t5000.json
cp $S/bench_t5000_serial_traced.json $D/t5000/bench_t5000_serial_traced.json
cp $S/t5000/t5000_kernel_stats.csv $D/t5000/kernel_stats_serial.csv
cp $S/pmc_traffic.json profiles/pmc_traffic.json
mkdir -p $D/lm
cp $S/lm/bench_lm.json $S/lm/bench_lm_traced.json $D/lm/
cp $S/lm/trace/lm_kernel_stats.csv $D/lm/kernel_stats.csv
cp $S/gemm_x3_vs_fp32.txt $S/pmc_gemm_x3_ffn_fwd.txt $S/conv_random_data.txt $S/conv_zero_data.txt $S/clock_probe.txt $D/
for f in conv_stall_breakdown.txt conv_step_model.txt ldsdma_rate.txt; do [ -f $S/$f ] && cp $S/$f $D/; done
if [ -d $S/one_task ]; then mkdir -p $D/one_task; cp $S/one_task/bench_one_task.json $S/one_task/bench_one_task_traced.json $D/one_task/; cp $S/one_task/trace/one_kernel_stats.csv $D/one_task/kernel_stats.csv; fi
if [ -d $S/ragged ]; then mkdir -p $D/ragged; cp $S/ragged/*.json $D/ragged/; cp $S/ragged/trace/ragged_kernel_stats.csv $D/ragged/kernel_stats.csv; fi
if [ -d $S/eval ]; then mkdir -p $D/eval; cp $S/eval/bench_eval.json $S/eval/bench_eval_traced.json $D/eval/; cp $S/eval/trace/eval_kernel_stats.csv $D/eval/kernel_stats.csv; fi
