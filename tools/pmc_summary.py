"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel -> small CSV (avg per dispatch, KiB as reported).
usage: pmc_summary.py <counter_collection.csv> <COUNTER> <out.csv>"""
import collections
import csv
import sys

src, counter, out = sys.argv[1:4]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    if r['Counter_Name'] == counter:
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'dispatches', counter + '_avg_per_dispatch', counter + '_total'])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), '%.1f' % (sum(v) / len(v)), '%.1f' % sum(v)])
