#!/bin/bash
# Run ON THE GPU BOX: SQ / LDS counter passes over the fused objective (tools/pmc_decoder.py), with and without gradients.
#   gpurun -- 'bash tools/pmc_decoder.sh [tag]'   -> gpurun_out/pmc_dm/summary.txt (+ profiles/<tag>_pmc_decoder.txt)
set -u
TAG=${1:-r03}
OUT=gpurun_out/pmc_dm
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  for mode in grad nograd; do
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$mode$i -- python3 tools/pmc_decoder.py $mode > $OUT/$mode$i.log 2>&1
  done
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for mode in ('grad', 'nograd'):
    for f in glob.glob('gpurun_out/pmc_dm/%s[0-9]*/**/*counter_collection.csv' % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'distmult_objective' not in k: continue
            agg[mode][r['Counter_Name']].append(float(r['Counter_Value']))
lines = ['distmult_objective_kernel, BioSNAP size (8.3 M positions), counters summed over the device, mean of the launches']
for mode in ('grad', 'nograd'):
    lines.append('%s:' % ('objective + gradients' if mode == 'grad' else 'objective only'))
    for c in sorted(agg[mode]):
        v = agg[mode][c]
        lines.append('    %-28s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
txt = '\n'.join(lines) + '\n'
open('gpurun_out/pmc_dm/summary.txt', 'w').write(txt)
open('profiles/%s_pmc_decoder.txt' % sys.argv[1], 'w').write(txt)
print(txt)
PY
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_pmc_decoder.txt gpurun_out/profiles_$TAG/
