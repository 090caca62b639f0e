#!/bin/bash
set -u
OUT=gpurun_out/pmc_rg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/pmc_relgather.py > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_rg/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rel_gather' not in k: continue
        key = k.split('(')[0][-44:] + ' grid=' + r['Grid_Size']
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_rg/summary.txt', 'w') as out:
    for k in sorted(agg):
        out.write(k + '\n')
        for c in sorted(agg[k]):
            v = agg[k][c]
            out.write('    %-30s %14.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open('gpurun_out/pmc_rg/summary.txt').read())
PY
