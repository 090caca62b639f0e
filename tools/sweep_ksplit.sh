#!/bin/bash
# Run ON THE GPU BOX: bench step time for several split-K policies of the small products.
for cfg in "64 512 128" "32 512 128" "32 1024 256" "32 1024 512" "64 1024 512" "32 2048 512"; do
  set -- $cfg
  TIPK_KSPLIT_GRAIN=$1 TIPK_KSPLIT_WGS=$2 TIPK_KSPLIT_MAX=$3 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('grain $1 wgs $2 max $3 :', round(d['ms_per_step'],4))"
done
