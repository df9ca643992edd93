#!/bin/bash
# PMC passes over isolated launches of the bf16-split GEMM engine (GPU box).  usage: bash tools/pmc_gemm_x3.sh  (writes gpurun_out/pmc_x3/summary.txt)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_x3; mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $O/p$i -o p -- python tools/bench_gemm_x3.py --only "${1:-ffn fwd (enc)}" > $O/log$i.txt 2>&1
done
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_x3/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm_x3_kernel' not in k:
            continue
        acc[re.sub(r'\(anonymous namespace\)::', '', k)[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_x3/summary.txt', 'w') as out:
    for k, cs in sorted(acc.items()):
        line = k + ':\n  ' + '\n  '.join('%s=%.5g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items()))
        print(line); out.write(line + '\n')
PY
rm -rf $O/p*
