"""Timeline digest of a rocprofv3 --kernel-trace CSV: where a step's wall time goes when kernels of several streams overlap.

usage: python tools/timeline.py KERNEL_TRACE.csv [--skip-frac 0.5] [--dump N]
Takes the dispatches of the LAST (1 - skip-frac) of the trace (the steady state), and prints
  * span, union-busy time (at least one kernel executing), idle time (none executing);
  * per queue: kernels, busy time, the gaps between consecutive kernels of that queue;
  * the idle intervals attributed to the kernel that ENDS them (what the GPU was waiting to start), by kernel class;
  * optionally the first N dispatches of the window as a text timeline.
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+)(<[^(]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def main():
    path = sys.argv[1]
    skip = 0.5
    dump = 0
    lo_us = hi_us = None
    for i, a in enumerate(sys.argv):
        if a == '--skip-frac':
            skip = float(sys.argv[i + 1])
        if a == '--dump':
            dump = int(sys.argv[i + 1])
        if a == '--window':          # microseconds relative to the first dispatch of the trace
            lo_us, hi_us = float(sys.argv[i + 1]), float(sys.argv[i + 2])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r.get('Stream_Id', '0'),
                         short(r['Kernel_Name'])))
    rows.sort()
    t00 = rows[0][0]
    if lo_us is not None:
        rows = [r for r in rows if lo_us * 1e3 <= r[0] - t00 <= hi_us * 1e3]
    else:
        t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
        rows = [r for r in rows if r[0] >= t_lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = (t1 - t0) / 1e3
    # union of busy intervals
    busy, idle_by = 0, collections.Counter()
    idle_n = collections.Counter()
    cur_end = rows[0][0]
    for s, e, q, st, n in rows:
        if s > cur_end:
            idle_by[n] += s - cur_end
            idle_n[n] += 1
            cur_end = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    print('window: %d dispatches, span %.1f us, busy(union) %.1f us, idle %.1f us (%.1f %%)' % (
        len(rows), span, busy / 1e3, span - busy / 1e3, 100 * (1 - busy / 1e3 / span)))
    ksum = sum(e - s for s, e, *_ in rows) / 1e3
    print('sum of kernel durations %.1f us (overlap factor %.2f)' % (ksum, ksum / (busy / 1e3)))
    byq = collections.defaultdict(list)
    for r in rows:
        byq[(r[2], r[3])].append(r)
    for q, rs in sorted(byq.items()):
        gaps = [b[0] - a[1] for a, b in zip(rs, rs[1:])]
        pos = [g for g in gaps if g > 0]
        print('queue %s stream %s: %d kernels, busy %.1f us, median gap %.2f us, mean positive gap %.2f us, gaps<0: %d' % (
            q[0], q[1], len(rs), sum(e - s for s, e, *_ in rs) / 1e3, (sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0),
            (sum(pos) / len(pos) / 1e3 if pos else 0), sum(1 for g in gaps if g < 0)))
    print('idle time by the kernel that ends the idle interval:')
    for n, v in idle_by.most_common(25):
        print('  %9.1f us  %5d x  %6.2f us each  %s' % (v / 1e3, idle_n[n], v / idle_n[n] / 1e3, n))
    dur = collections.defaultdict(lambda: [0, 0])
    for s, e, q, st, n in rows:
        dur[n][0] += e - s
        dur[n][1] += 1
    print('kernel time by symbol:')
    for n, (v, c) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:40]:
        print('  %9.1f us  %5d x  %7.2f us  %s' % (v / 1e3, c, v / c / 1e3, n))
    if dump:
        print('timeline (us from window start):')
        for s, e, q, st, n in rows[:dump]:
            print('  %9.2f  +%7.2f  s%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, st, n))


if __name__ == '__main__':
    main()
