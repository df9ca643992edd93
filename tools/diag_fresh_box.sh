#!/bin/bash
# bench.py under the driver's conditions (first GPU process of a fresh container, concurrent `rocm-smi` poll), with the host phase
# trace; container CPU quota / throttling counters before and after.  Output: gpurun_out/fresh/.
out=gpurun_out/fresh; rm -rf $out; mkdir -p $out
{ echo "nproc $(nproc)  cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat; } > $out/host.txt 2>&1
( while true; do rocm-smi -a --json > $out/smi_all.json 2>/dev/null; sleep 1; done ) &
poll=$!
MTL_BENCH_TRACE=1 MTL_TRACE_PHASES=1 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $out/a_poll_traced.json 2> $out/a_poll_traced.err
{ echo "after a:"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat; } >> $out/host.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/b_full_poll.json 2> $out/b_full_poll.err ) 2> $out/b_time.txt
{ echo "after b:"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat; } >> $out/host.txt
kill $poll
cp gpurun_out/bench_detail.json $out/b_detail.json
for f in a_poll_traced b_full_poll; do tail -1 $out/$f.json | cut -c1-3000; grep "^trace" $out/$f.err | tail -1 | cut -c1-500; done
cat $out/host.txt $out/b_time.txt
