"""Reads a rocprofv3 --kernel-trace CSV and reports, for the LAST `--frac` of the dispatches (steady state): wall time, union busy
time, kernel-time sum, the idle gaps between consecutive dispatches, and the kernels with the largest total time.

    rocprofv3 --kernel-trace -d /tmp/tr -o one -- python bench.py --tasks 1 --steps 10 --warmup 3
    python tools/trace_gaps.py /tmp/tr/**/one_kernel_trace.csv
"""
import argparse
import csv
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument('csv')
ap.add_argument('--frac', type=float, default=0.5)
ap.add_argument('--top', type=int, default=25)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
ev = ev[int(len(ev) * (1 - a.frac)):]
wall = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _ in ev)
print('dispatches %d  wall %.3f ms  busy(union) %.3f ms  kernel sum %.3f ms  idle %.3f ms in %d gaps (mean %.2f us)'
      % (len(ev), wall / 1e6, busy / 1e6, ksum / 1e6, sum(gaps) / 1e6, len(gaps), sum(gaps) / max(len(gaps), 1) / 1e3))
hist = {}
for g in gaps:
    b = min(int(g / 1000), 20)
    hist[b] = hist.get(b, 0) + 1
print('gap histogram (us: count):', ' '.join('%d:%d' % kv for kv in sorted(hist.items())))
agg = {}
for s, e, n in ev:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n)[:90]
    t = agg.setdefault(n, [0, 0])
    t[0] += e - s
    t[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print('%8.3f ms %6d x %7.2f us  %s' % (t / 1e6, c, t / c / 1e3, n))
