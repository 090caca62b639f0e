#!/bin/bash
# VGPR / SGPR / LDS / scratch of every gfx950 kernel in tip_amd/csrc (compile only, no GPU needed).
cd "$(dirname "$0")/../tip_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps=obj -c $f -o /tmp/kr_$$.o 2>/dev/null
  s=/tmp/${f%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s
  [ -f "$s" ] || s=$(ls -t /tmp/*-hip-amdgcn-amd-amdhsa-gfx950.s | head -1)
  python3 - "$s" "$f" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    get = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', body) or [0, '0'])[1]
    dem = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', name)[:60]
    print('%-22s %-60s vgpr %4s  lds %6s  scratch %s' % (sys.argv[2], dem, get('next_free_vgpr'), get('group_segment_fixed_size'), get('private_segment_fixed_size')))
PY
done
rm -f /tmp/kr_$$.o
