#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel timeline of one graphed training epoch (objective, sampler, encoder).
#   gpurun -- 'bash tools/quick_train.sh <tag>'
TAG=${1:-qt}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 tools/trace_train.py > $OUT/train.log 2>&1
python3 tools/step_timeline.py $OUT/train distmult_objective_kernel > $OUT/train_timeline.txt 2>&1
grep "ms/epoch" $OUT/train.log >> $OUT/train_timeline.txt
rm -rf $OUT/train
grep -E "distmult|negative|bitmap|sampl|ms/epoch|span" $OUT/train_timeline.txt | head -20
