#!/bin/bash
# Scheduling knobs of the 8-task step: task lanes x hardware queues x convolution CU cap.  Run on the GPU box.
run() { echo "$@"; env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', round(d['value'],3), round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],1))"; }
for lanes in 1 2 3 4; do run MTL_TASK_LANES=$lanes; done
for q in 2 3 5 6 8; do run GPU_MAX_HW_QUEUES=$q MTL_TASK_LANES=3; done
run MTL_X3_CUS=224 MTL_TASK_LANES=3
