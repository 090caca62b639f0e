#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the default bench line without the side legs + (optionally) the timeline of one replayed step.
#   gpurun -- 'bash tools/quick_bench.sh <tag> [trace]'
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --steps 100 --warmup 20 --no-extras --no-pmc --no-cpu-baseline > $OUT/b100.json 2> $OUT/b100.err
if [ "${2:-}" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_trace.json 2> $OUT/bench_trace.err
  python3 tools/step_timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
  cat $OUT/timeline.txt
  rm -rf $OUT/trace
fi
python3 - <<PY
import json
for l in open("$OUT/b100.json"):
    if l.startswith("{"):
        j = json.loads(l)
        print(j["ms_per_step"], j["dd_launches_us"], j["roofline"]["kernel"], j["roofline"]["frac"], j["step_floor"]["frac"])
PY
tail -3 $OUT/b100.err
