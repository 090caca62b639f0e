#!/bin/bash
# Run ON THE GPU BOX: step time under a few host-side tunables (two runs each; round 3: everything within the +-1 % run-to-run
# noise of the defaults, TIPK_CHUNK = 32 / 64 slower).   gpurun -- 'bash tools/sweep_env.sh'
run() { python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])"; }
echo base; run; run
for v in 256 1024; do echo KSPLIT_WGS=$v; TIPK_KSPLIT_WGS=$v run; TIPK_KSPLIT_WGS=$v run; done
for v in 8 12 24 32; do echo RS_WIDE_STEPS=$v; TIPK_RS_WIDE_STEPS=$v run; TIPK_RS_WIDE_STEPS=$v run; done
for v in 16 32 64; do echo CHUNK=$v; TIPK_CHUNK=$v run; TIPK_CHUNK=$v run; done
