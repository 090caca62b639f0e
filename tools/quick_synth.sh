#!/bin/bash
# Run ON THE GPU BOX (via gpurun): config 5 (synthetic) step + its kernel timeline.
#   gpurun -- 'bash tools/quick_synth.sh <tag>'
TAG=${1:-qs}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload synthetic --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_trace.json 2> $OUT/bench_trace.err
python3 tools/step_timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt | awk '$4 > 50 || /span/'
rm -rf $OUT/trace
tail -2 $OUT/bench_trace.err
python3 - <<PY
import json
for l in open("$OUT/bench_trace.json"):
    if l.startswith("{"):
        j = json.loads(l)
        print(j["ms_per_step"], j.get("whole_step"), j.get("parity_in_bench"))
PY
