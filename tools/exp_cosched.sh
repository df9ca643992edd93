#!/bin/bash
# Co-scheduling experiment: do the convolutions' persistent grids on fewer CUs leave room for the other tasks'
# transformer kernels?  Existing switches only: --lanes (one pass chain per task on concurrent lanes), MTL_TASK_LANES, MTL_X3_CUS.
out=gpurun_out/cosched; mkdir -p $out
B="python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" $B ${LANES:+--lanes} 2>/dev/null | tail -1 > $out/$name.json; python - $out/$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('%-28s %6.2f meta-steps/s  %6.2f ms  host %s' % (sys.argv[2], d['value'], d['ms_per_step'], d.get('host_enqueue_ms')))
except Exception as e: print(sys.argv[2], 'ERR', e)
P
}
LANES= run batched_256 X=1
LANES= run batched_224 MTL_X3_CUS=224
LANES=1 run lanes8_256 MTL_TASK_LANES=8
LANES=1 run lanes8_224 MTL_TASK_LANES=8 MTL_X3_CUS=224
LANES=1 run lanes8_192 MTL_TASK_LANES=8 MTL_X3_CUS=192
LANES=1 run lanes4_224 MTL_TASK_LANES=4 MTL_X3_CUS=224
LANES=1 run lanes4_192 MTL_TASK_LANES=4 MTL_X3_CUS=192
LANES=1 run lanes2_224 MTL_TASK_LANES=2 MTL_X3_CUS=224
LANES=1 run lanes2_192 MTL_TASK_LANES=2 MTL_X3_CUS=192
LANES=1 run lanes2_256 MTL_TASK_LANES=2
