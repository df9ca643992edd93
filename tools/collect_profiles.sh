#!/bin/bash
# Runs on the GPU box (via gpurun): collects the round's rocprofv3 summaries into gpurun_out/$ROUND (default r6; copied to profiles/$ROUND afterwards).
# usage: bash tools/collect_profiles.sh
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${ROUND:-r6}
mkdir -p $O
# counter passes FIRST: bench.py reads profiles/pmc_traffic.json for `roofline.traffic` (per launch, so it must come from the same launch structure)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --serial > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --serial > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python tools/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv FETCH_SIZE $O/pmc_fetch_size_per_kernel.csv
python tools/pmc_summary.py $O/pmc_write/w_counter_collection.csv WRITE_SIZE $O/pmc_write_size_per_kernel.csv
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/bench.json 2> $O/bench.err; cp gpurun_out/bench_detail.json $O/bench_detail.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial > $O/bench_serial_traced.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lanes -o lanes -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_lanes_traced.json 2> /dev/null
# what ONE rank of the 8-GPU configuration runs (BASELINE.json configs[2]): one task per step
mkdir -p $O/one_task
rocprofv3 --kernel-trace --stats --output-format csv -d $O/one_task/trace -o one -- python bench.py --tasks 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/one_task/bench_one_task_traced.json 2> /dev/null
python bench.py --tasks 1 --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2> /dev/null | tail -1 > $O/one_task/bench_one_task.json
# manifest-like batches (every batch padded to its own longest utterance, new shapes every step): probe line + kernel stats
mkdir -p $O/ragged
python tools/probe/ragged_steps.py --mode ragged --steps 40 2> /dev/null | tail -1 > $O/ragged/ragged_stacked.json
python tools/probe/ragged_steps.py --mode ragged --steps 20 --lanes --own-widths 2> /dev/null | tail -1 > $O/ragged/ragged_lane_per_task_own_widths.json
python tools/probe/ragged_steps.py --mode ragged --steps 40 --lanes 2> /dev/null | tail -1 > $O/ragged/ragged_lane_per_task_rounded_widths.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ragged/trace -o ragged -- python tools/probe/ragged_steps.py --mode ragged --steps 6 > /dev/null 2>&1
# f2: the validation loop's forward and greedy decoding (bench.py's `eval` leg) under the profiler
mkdir -p $O/eval
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval/trace -o eval -- python bench.py --eval-only > $O/eval/bench_eval_traced.json 2> /dev/null
python bench.py --eval-only 2> /dev/null | tail -1 > $O/eval/bench_eval.json
# long-utterance configuration (BASELINE.json configs[3]): T = 5000
python bench.py --frames 5000 --tasks 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_t5000.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t5000 -o t5000 -- python bench.py --frames 5000 --tasks 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --serial > $O/bench_t5000_serial_traced.json 2> /dev/null
# LM meta loop (BASELINE.json configs[4]): bench line + rocprofv3 summary
mkdir -p $O/lm
python bench.py --workload lm > $O/lm/bench_lm.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lm/trace -o lm -- python bench.py --workload lm --steps 3 --warmup 1 --no-cpu-baseline > $O/lm/bench_lm_traced.json 2> /dev/null
# isolated products on the bf16-split engine vs the fp32 engines, PMC counters of one of them, and the DVFS probe of the convolutions
python tools/bench_gemm_x3.py 2> /dev/null | grep -v amdgpu.ids > $O/gemm_x3_vs_fp32.txt
bash tools/pmc_gemm_x3.sh "ffn fwd (enc)" > /dev/null 2>&1; cp gpurun_out/pmc_x3/summary.txt $O/pmc_gemm_x3_ffn_fwd.txt
python tools/bench_conv.py 2>/dev/null | grep h2 > $O/conv_random_data.txt
MTL_BENCH_ZERO=1 python tools/bench_conv.py 2>/dev/null | grep h2 > $O/conv_zero_data.txt
python tools/clock_probe.py 3 2>/dev/null | grep -v amdgpu.ids > $O/clock_probe.txt
# in-kernel stall breakdown of the convolutions, the tap-step model with controlled operands, LDS-DMA fill rate (tools/probe/build_probes.sh first)
if [ -f tools/probe/libmtl_prof.so ]; then MTL_LIB=$PWD/tools/probe/libmtl_prof.so python tools/probe/conv_prof.py 2>/dev/null | grep -v amdgpu.ids > $O/conv_stall_breakdown.txt; fi
if [ -x tools/probe/conv_step_model ]; then timeout 120 tools/probe/conv_step_model > $O/conv_step_model.txt 2>/dev/null; fi
if [ -x tools/probe/ldsdma_rate ]; then timeout 120 tools/probe/ldsdma_rate > $O/ldsdma_rate.txt 2>/dev/null; fi
rm -rf $O/pmc_fetch $O/pmc_write
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -R $O | head -40
