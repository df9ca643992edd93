run() { echo "$@"; env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1))"; }
run MTL_TASK_LANES=2
run MTL_TASK_LANES=3
run MTL_X3_CUS=224 MTL_TASK_LANES=3
run MTL_X3_CUS=240 MTL_TASK_LANES=3
run MTL_X3_CUS=224 MTL_TASK_LANES=4
run MTL_X3_CUS=192 MTL_TASK_LANES=3
run MTL_TASK_LANES=2
