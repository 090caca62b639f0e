#!/bin/bash
# Run ON THE GPU BOX: SQ / cache counter passes over the backward D-D kernels (tools/pmc_node_products.py).
#   gpurun -- 'bash tools/pmc_node_products.sh [tag]'  -> profiles/<tag>_pmc_node_products.txt
set -u
TAG=${1:-r03}
OUT=gpurun_out/pmc_np
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/pmc_node_products.py > $OUT/p$i.log 2>&1
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_np/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'node_products' not in k and 'stream_gather' not in k: continue
        key = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0] + ' grid=' + r['Grid_Size']
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
lines = ['backward D-D kernels of the BioSNAP step (layer 1: d = 32, layer 2: d = 16), counters summed over the device, mean of 3 launches each']
for k in sorted(agg):
    lines.append(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        lines.append('    %-34s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
txt = '\n'.join(lines) + '\n'
open('profiles/%s_pmc_node_products.txt' % sys.argv[1], 'w').write(txt)
print(txt)
PY
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_pmc_node_products.txt gpurun_out/profiles_$TAG/
rm -rf $OUT/p*/
