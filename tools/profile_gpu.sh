#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + the two HBM-traffic PMC passes of the bench step
# (+ the SQ/LDS counter passes of the D-D aggregation kernels when no bench arguments are given).
#   gpurun -- 'bash tools/profile_gpu.sh r02 [bench args] '
# Raw rocprofv3 output lands in gpurun_out/prof_<tag>/ (scratch); tools/summarize_prof.py turns it
# into the committed summaries under profiles/.  The program itself follows `--` (no env / bash -c hop).
set -u
TAG=${1:-r02}
shift || true
ARGS="$@"
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
# 1. per-kernel time of the default bench command (hipGraph replay of the step)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# 2./3. HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --launch eager --step-only $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --launch eager --step-only $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
# 4. SQ / LDS counters of the D-D aggregation kernels (own passes, --kernel-trace only besides --pmc)
if [ -z "$ARGS" ]; then
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_sq$i -- python3 tools/pmc_relgather.py > $OUT/pmc_sq$i.log 2>&1
done
fi
python3 tools/summarize_prof.py $OUT $TAG
ls -la $OUT profiles | head -60
# 5. timeline of one replayed step, the graphed training epoch (trace + timeline), kept next to the summaries
python3 tools/step_timeline.py $OUT/trace > profiles/${TAG}_step_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 tools/trace_train.py > $OUT/train.log 2>&1
python3 tools/step_timeline.py $OUT/train distmult_objective_kernel > profiles/${TAG}_train_timeline.txt 2>&1
grep "ms/epoch" $OUT/train.log >> profiles/${TAG}_train_timeline.txt
# 6. SQ / LDS counters of the fused objective (with and without gradients)
if [ -z "$ARGS" ]; then bash tools/pmc_decoder.sh $TAG > $OUT/pmc_decoder.log 2>&1; fi
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
# only the summaries and the logs travel back (gpurun merges at most 64 MiB of gpurun_out/): drop the raw rocprofv3 output
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq* $OUT/train gpurun_out/pmc_dm/grad* gpurun_out/pmc_dm/nograd*
du -sh gpurun_out
