#!/bin/bash
# bench.py under the driver's conditions (first GPU process of a fresh container, concurrent `rocm-smi` poll), with the host phase
# trace; container CPU quota / throttling counters before and after.  Output: gpurun_out/fresh/.
out=gpurun_out/fresh; rm -rf $out; mkdir -p $out
{ echo "nproc $(nproc)  cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat; } > $out/host.txt 2>&1
( while true; do rocm-smi -a --json > $out/smi_all.json 2>/dev/null; sleep 1; done ) &
poll=$!
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/b_full_poll.json 2> $out/b_full_poll.err ) 2> $out/b_time.txt
{ echo "after b:"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat; } >> $out/host.txt
kill $poll
cp gpurun_out/bench_detail.json $out/b_detail.json
tail -1 $out/b_full_poll.json | cut -c1-3000
cat $out/host.txt $out/b_time.txt
