"""Summarise a rocprofv3 (ROCm 7.x rocpd sqlite) kernel trace: per-kernel calls / total / avg / share."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
span = cur.execute("select max(end)-min(start) from kernels").fetchone()[0]
print('# kernels: %d dispatches, %.3f ms GPU-busy (sum of durations), %.3f ms first-start..last-end' % (sum(r[1] for r in rows), tot / 1e6, span / 1e6))
print('%-110s %8s %12s %10s %10s %10s %6s' % ('name', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct'))
for name, n, s, a, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    short = name if len(name) <= 108 else name[:105] + '...'
    print('%-110s %8d %12.3f %10.2f %10.2f %10.2f %6.2f' % (short, n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
