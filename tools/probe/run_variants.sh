run() { echo "$@"; env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1))"; }
run MTL_TASK_LANES=3
run MTL_SHARED_SIDE=1 MTL_TASK_LANES=3
run MTL_STREAM_PAD=1 MTL_TASK_LANES=3
run MTL_STREAM_PAD=2 MTL_TASK_LANES=3
run MTL_STREAM_PAD=3 MTL_TASK_LANES=3
run MTL_SHARED_SIDE=1 MTL_STREAM_PAD=1 MTL_TASK_LANES=3
run MTL_SHARED_SIDE=1 GPU_MAX_HW_QUEUES=8 MTL_TASK_LANES=3
