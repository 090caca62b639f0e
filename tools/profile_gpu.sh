#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + the two HBM-traffic PMC passes of the bench step.
#   gpurun -- 'bash tools/profile_gpu.sh r01'
# Raw rocprofv3 output lands in gpurun_out/prof_<tag>/ (scratch); tools/summarize_prof.py turns it
# into the committed summaries under profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
# 1. per-kernel time of the default bench command (hipGraph replay of the step + eager timing pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# 2./3. HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager > $OUT/bench_write.json 2> $OUT/bench_write.err
python3 tools/summarize_prof.py $OUT $TAG
ls -la $OUT profiles | head -40
