#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ / LDS / memory counters of tipk_rgcn_row_products at config-5 size.
#   gpurun -- 'bash tools/pmc_row_products.sh <tag>'   -> profiles/<tag>_pmc_row_products.txt
set -u
TAG=${1:-r05}
OUT=gpurun_out/pmc_rp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/bench_row_products.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_rp/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'row_products' not in k: continue
        import re
        m = re.search(r'row_products(_s)?_kernel<[^>]*>', k)
        key = m.group(0) if m else k[:60]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open('profiles/${TAG}_pmc_row_products.txt', 'w') as out:
    out.write('tipk_rgcn_row_products at config-5 size (N = 10 000, R = 2 000, E = 50 M, 128 channels): counters summed over the device, mean of the launches\n')
    for k in sorted(agg):
        out.write(k + '\n')
        for c in sorted(agg[k]):
            v = agg[k][c]
            out.write('    %-34s %16.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
PY
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_pmc_row_products.txt gpurun_out/profiles_$TAG/
cat profiles/${TAG}_pmc_row_products.txt
for j in $(seq 1 $i); do tail -2 $OUT/p$j.log | head -1; done
rm -rf $OUT/p*/
