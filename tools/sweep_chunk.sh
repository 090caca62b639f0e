#!/bin/bash
# sweep the gather plan chunk (edges per work item) and print step time + P-P/P->D kernel times
for c in 16 32 64 128; do
  python bench.py --no-cpu-baseline --chunk $c 2>/dev/null | grep "^{" > /tmp/sw.json
  python3 - $c <<'PY'
import json, sys
d = json.load(open('/tmp/sw.json')); k = d['kernels_ms']
print('chunk', sys.argv[1], 'ms/step %.4f' % d['ms_per_step'],
      {n: round(v['mean_ms'] * 1e3, 1) for n, v in k.items() if n.startswith('gather_sum')})
PY
done
