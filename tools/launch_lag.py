"""Host-vs-GPU lag per kernel from a rocprofv3 --hip-runtime-trace --kernel-trace run (CSV): for every dispatch, the time between the
END of its hipLaunchKernel call on the host and its START on the GPU.  A small lag at a kernel that was preceded by GPU idle time
means the GPU was waiting for the host there.
usage: python tools/launch_lag.py HIP_API_TRACE.csv KERNEL_TRACE.csv [--skip-frac 0.6] [--dump N]"""
import csv
import re
import sys

api, ker = sys.argv[1], sys.argv[2]
skip, dump = 0.6, 0
for i, a in enumerate(sys.argv):
    if a == '--skip-frac':
        skip = float(sys.argv[i + 1])
    if a == '--dump':
        dump = int(sys.argv[i + 1])
calls = {}
for r in csv.DictReader(open(api)):
    calls[r['Correlation_Id']] = (r['Function'], int(r['Start_Timestamp']), int(r['End_Timestamp']))
rows = []
for r in csv.DictReader(open(ker)):
    c = calls.get(r['Correlation_Id'])
    name = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name'])[:48]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'], name, c))
rows.sort()
t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
prev_end = {}
last_end_any = rows[0][0]
n_starved = 0
starved_us = 0.0
lines = []
for s, e, st, name, c in rows:
    idle = max(0, s - last_end_any) / 1e3            # GPU-wide idle right before this kernel
    lag = (s - c[2]) / 1e3 if c else float('nan')     # kernel start minus end of its launch call
    if idle > 2.0 and c and lag < 12.0:
        n_starved += 1
        starved_us += idle
    lines.append('%10.2f +%7.2f s%-2s idle %6.2f  lag %9.2f  %s' % ((s - rows[0][0]) / 1e3, (e - s) / 1e3, st, idle, lag, name))
    last_end_any = max(last_end_any, e)
span = (max(r[1] for r in rows) - rows[0][0]) / 1e3
print('%d dispatches over %.1f us; idle intervals > 2 us whose kernel had been launched < 12 us before it started: %d (%.1f us)'
      % (len(rows), span, n_starved, starved_us))
for l in lines[:dump]:
    print(l)
