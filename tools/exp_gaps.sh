cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5/gaps; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o prod -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null
f=$(find $O/tr -name "prod_kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f --frac 0.4 --top 12 > $O/trace_gaps.txt 2>&1
python tools/timeline.py $f --skip-frac 0.6 > $O/timeline.txt 2>&1
rm -rf $O/tr
head -30 $O/trace_gaps.txt; head -60 $O/timeline.txt
