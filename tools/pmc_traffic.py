"""Build profiles/pmc_traffic.json from two rocprofv3 counter_collection CSVs (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE
passes of `bench.py --steps 1 --serial`: the task-batched step): HBM bytes per launch of every kernel class = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
(FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md's gfx950 correction).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys

_BASE = {  # kernel template instance <BN, sub-tiles, UNPOOL, EPI (0 relu / 1 pool / 2 dgrad)> (the trailing piece count and consumer
    # layout arguments are stripped before the lookup) -> bench.py conv class (see PassEngine.forward/backward)
    'conv3x3_x3h_kernel<64, 1, false, 1': 'conv2_fwd_pool', 'conv3x3_x3h_kernel<64, 2, false, 1': 'conv2_fwd_pool',
    'conv3x3_x3h_kernel<128, 2, false, 1': 'conv7_fwd_pool',
    'conv3x3_x3h_kernel<128, 2, false, 0': 'conv5_fwd', 'conv3x3_x3h_kernel<64, 1, true, 2': 'conv2_dgrad',
    'conv3x3_x3h_kernel<128, 2, true, 2': 'conv7_dgrad', 'conv3x3_x3h_kernel<64, 2, false, 2': 'conv5_dgrad',
    'conv3x3_wgrad_x3_kernel<false': 'conv5_wgrad',
    'conv3x3_wgrad_x3_kernel<true': ('conv7_wgrad', 'conv2_wgrad'),      # same instance: the backward runs conv7 first, then conv2
    'conv3x3_wgrad_sp_kernel': ('conv7_wgrad', 'conv2_wgrad'),           # round 4: the pooled layers' 2:4-sparse form
}


def conv_class(kernel_name):
    m = re.search(r'(conv3x3_x3h_kernel<\d+, \d+, \w+, \d+|conv3x3_wgrad_x3_kernel<\w+|conv3x3_wgrad_sp_kernel)', kernel_name)
    return _BASE.get(m.group(1)) if m else None


# kernel-name substring -> bench.py class for the non-convolution classes (a class = one C-ABI call; calls that launch two
# kernels, e.g. mtl_attn_bwd, are averaged per kernel and summed)
GROUPS = {'gemm_x3': [', 3>((anonymous namespace)::X3P'], 'gemm_small': ['gemm16_kernel'],
          'gemm_h2': [', 2>((anonymous namespace)::X3P', 'gemm_nt_h2_kernel', 'gemm_h2_reduce_kernel'], 'gemm_big': ['gemm_kernel<', 'splitk_reduce_kernel'], 'attn_fwd': ['attn_fwd_kernel'],
          'attn_bwd': ['attn_bwd_kernel'], 'layernorm_fwd': ['layernorm_fwd_kernel'],
          'layernorm_bwd': ['layernorm_bwd_kernel', 'ln_param_reduce_kernel', 'ln_param_reduce_batch_kernel'], 'conv0_fwd': ['conv0_fwd_kernel'],
          'conv0_wgrad': ['conv0_wgrad_kernel', 'conv0_wgrad_final_kernel']}


def per_class(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    out, seen = collections.defaultdict(list), collections.Counter()
    parts = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        for cls, subs in GROUPS.items():
            for sub in subs:
                if sub in r['Kernel_Name']:
                    parts[cls][sub].append(float(r['Counter_Value']))
    for cls, d in parts.items():
        n = max(len(v) for v in d.values())
        if cls == 'gemm_big':
            n = len(d.get('gemm_kernel<', [])) or n          # a split-K reduction belongs to the call of its GEMM
        out[cls] = [sum(sum(v) for v in d.values()) / n]
    for r in rows:
        cls = conv_class(r['Kernel_Name'])
        if cls is None:
            continue
        if isinstance(cls, tuple):
            cls = cls[seen[cls] % 2]
            seen[conv_class(r['Kernel_Name'])] += 1
        out[cls].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch = per_class(sys.argv[1], 'FETCH_SIZE')
write = per_class(sys.argv[2], 'WRITE_SIZE')
traffic = {k: int((2 * fetch[k] + write.get(k, 0.0)) * 1024) for k in fetch}
json.dump(traffic, open(sys.argv[3], 'w'), indent=1)
for k in sorted(traffic):
    print('%-16s fetch %10.0f KiB (x2)  write %10.0f KiB  -> %.3f GB per launch' % (k, fetch[k], write.get(k, 0.0), traffic[k] / 1e9))
