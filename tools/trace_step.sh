#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the bench step and its timeline (tools/timeline.py).  usage: trace_step.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out/trace_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --launch eager > $OUT/bench.json 2> $OUT/bench.err
python3 tools/timeline.py $OUT > gpurun_out/timeline_$TAG.txt 2>&1
cut -c1-96 gpurun_out/timeline_$TAG.txt
