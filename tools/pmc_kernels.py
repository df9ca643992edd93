"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel (substring filter), mean of every counter."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2:] or ['']
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if any(f in r['Kernel_Name'] for f in filt):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k[:100])
    print('   ' + '  '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
