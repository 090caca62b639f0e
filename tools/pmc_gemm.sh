#!/bin/bash
# Run ON THE GPU BOX: PMC passes over tools/pmc_gemm.py (counter groups kept small: one pass each).
set -u
OUT=gpurun_out/pmc_gemm
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/pmc_gemm.py > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_gemm/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm' not in k: continue
        key = k.split('(')[0][-40:] + ' grid=' + r['Grid_Size']
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_gemm/summary.txt', 'w') as out:
    for k in sorted(agg):
        out.write(k + '\n')
        for c in sorted(agg[k]):
            v = agg[k][c]
            out.write('    %-34s %14.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open('gpurun_out/pmc_gemm/summary.txt').read())
PY
