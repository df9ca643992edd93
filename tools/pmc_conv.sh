#!/bin/bash
# PMC passes over the isolated convolution launches of tools/bench_conv.py (GPU box): matrix-pipe busy cycles, LDS bank conflicts and
# LDS activity per h2 kernel.  usage: bash tools/pmc_conv.sh  (writes gpurun_out/pmc_conv/summary.txt)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_conv; mkdir -p $O
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $O/p$i -o p -- python tools/bench_conv.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_conv/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'x3h_kernel' not in k and 'wgrad_x3_kernel' not in k:
            continue
        m = re.search(r'(conv3x3_\w+<[^>]*>)', k)
        acc[m.group(1) if m else k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_conv/summary.txt', 'w') as out:
    for k, cs in sorted(acc.items()):
        line = k + ': ' + '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items()))
        print(line); out.write(line + '\n')
PY
rm -rf $O/p*
