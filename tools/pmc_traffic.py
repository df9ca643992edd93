"""Build profiles/pmc_traffic.json from two rocprofv3 counter_collection CSVs (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE
passes of `bench.py --tasks 1 --serial`): HBM bytes per launch of every conv class = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
(FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md's gfx950 correction).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys

CLASSES = {  # kernel template instance -> bench.py conv class (see PassEngine.forward/backward)
    'conv3x3_x3h_kernel<64, 1, false, 1>': 'conv2_fwd_pool', 'conv3x3_x3h_kernel<128, 2, false, 1>': 'conv7_fwd_pool',
    'conv3x3_x3h_kernel<128, 2, false, 0>': 'conv5_fwd', 'conv3x3_x3h_kernel<64, 1, true, 2>': 'conv2_dgrad',
    'conv3x3_x3h_kernel<128, 2, true, 2>': 'conv7_dgrad', 'conv3x3_x3h_kernel<64, 2, false, 2>': 'conv5_dgrad',
    'conv3x3_wgrad_x3_kernel<false>': 'conv5_wgrad',
    'conv3x3_wgrad_x3_kernel<true>': ('conv7_wgrad', 'conv2_wgrad'),      # same instance: the backward runs conv7 first, then conv2
}


def per_class(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    out, seen = collections.defaultdict(list), collections.Counter()
    for r in rows:
        m = re.search(r'(conv3x3_\w+<[^>]*>)', r['Kernel_Name'])
        if not m or m.group(1) not in CLASSES:
            continue
        cls = CLASSES[m.group(1)]
        if isinstance(cls, tuple):
            cls = cls[seen[m.group(1)] % 2]
            seen[m.group(1)] += 1
        out[cls].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch = per_class(sys.argv[1], 'FETCH_SIZE')
write = per_class(sys.argv[2], 'WRITE_SIZE')
traffic = {k: int((2 * fetch[k] + write.get(k, 0.0)) * 1024) for k in fetch}
json.dump(traffic, open(sys.argv[3], 'w'), indent=1)
for k in sorted(traffic):
    print('%-16s fetch %10.0f KiB (x2)  write %10.0f KiB  -> %.3f GB per launch' % (k, fetch[k], write.get(k, 0.0), traffic[k] / 1e9))
