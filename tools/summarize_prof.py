#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile_gpu.sh) into the small summaries kept under profiles/.

  profiles/<tag>_kernel_stats.csv    per-kernel calls / total / average / min / max (ns): the rocprofv3
                                     --kernel-trace --stats table, names shortened
  profiles/<tag>_kernel_by_grid.csv  the same trace split by FULL grid XxYxZ (one line per launch shape)
  profiles/<tag>_pmc_traffic.json    per (kernel, grid XxYxZ): average FETCH_SIZE / WRITE_SIZE per launch
                                     and the HBM bytes derived as the MI355X guide prescribes
                                     (bytes = KB * 1024; gfx950 reads: FETCH_SIZE doubled)
  profiles/<tag>_lds.json            per (kernel, grid): SQ / LDS counters of the relation-local kernels
                                     (tools/pmc_relgather.sh passes, if present under <src>/pmc_sq*)

PMC rows carry only the flattened grid size; the x*y*z shape is joined in from the kernel trace of the
same pass by Dispatch_Id, so launches that differ only in their grid shape (d=32 forward: 131072x2,
d=16 forward: 262144x1) stay separate.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
os.makedirs('profiles', exist_ok=True)
csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)\(', name)
    if m:
        name = m.group(1)
    return name[:110]


def find(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[0] if hits else None


def grid_of(row):
    return '%sx%sx%s' % (row.get('Grid_Size_X', '?'), row.get('Grid_Size_Y', '?'), row.get('Grid_Size_Z', '?'))


stats = find('trace/**/*kernel_stats.csv')
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open('profiles/%s_kernel_stats.csv' % tag, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline`\n')
        f.write('kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n')
        for r in rows[:45]:
            f.write('"%s",%s,%s,%.0f,%s,%s,%s\n' % (short(r['Name']), r['Calls'], r['TotalDurationNs'],
                                                  float(r['AverageNs']), r['Percentage'], r['MinNs'], r['MaxNs']))

trace = find('trace/**/*kernel_trace.csv')
if trace:
    agg = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        n = short(r['Kernel_Name'])
        if n.startswith('at::') or n.startswith('rocprim') or 'elementwise' in n:
            continue                                                   # torch glue of the setup phase
        agg['%s grid=%s' % (n, grid_of(r))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    with open('profiles/%s_kernel_by_grid.csv' % tag, 'w') as f:
        f.write('# libtipk launches of the same command, split by full grid XxYxZ (one line per distinct launch shape)\n')
        f.write('kernel_and_grid,calls,avg_ns,min_ns,max_ns\n')
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.0f,%d,%d\n' % (k, len(v), sum(v) / len(v), min(v), max(v)))


def counters_by_launch(sub):
    """{(kernel, grid): {counter: [values]}} of one --pmc pass directory (joined with its kernel trace)."""
    fn = find(sub + '/**/*counter_collection.csv')
    if not fn:
        return {}
    tr = find(sub + '/**/*kernel_trace.csv')
    shape = {}
    if tr:
        for r in csv.DictReader(open(tr)):
            shape[r['Dispatch_Id']] = grid_of(r)
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(fn)):
        n = short(r['Kernel_Name'])
        g = shape.get(r['Dispatch_Id'], '%sx?x?' % r.get('Grid_Size', '?'))
        per['%s grid=%s' % (n, g)][r['Counter_Name']].append(float(r['Counter_Value']))
    return per


pmc = {}
for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    for key, cs in counters_by_launch(sub).items():
        v = cs.get(counter)
        if v:
            pmc.setdefault(key, {})[counter + '_KB_avg'] = sum(v) / len(v)
            pmc[key]['launches_' + counter] = len(v)
for n, d in pmc.items():
    f_kb, w_kb = d.get('FETCH_SIZE_KB_avg', 0.0), d.get('WRITE_SIZE_KB_avg', 0.0)
    d['hbm_bytes_per_launch'] = (2.0 * f_kb + w_kb) * 1024.0      # gfx950: FETCH_SIZE counts 1/2 of wide reads
    d['hbm_bytes_per_launch_uncorrected'] = (f_kb + w_kb) * 1024.0
step_bytes = step_bytes_raw = None
build_id = None
try:                                                 # the build the counters were collected on (bench.py drops stale summaries)
    build_id = json.loads([l for l in open(os.path.join(src, 'bench_fetch.json')) if l.startswith('{')][-1]).get('build_id')
except Exception as exc:
    print('no build id:', exc)
if pmc:
    # bytes of ONE step: every libtipk launch of the PMC passes (bench.py --step-only runs nothing but steps) / steps run
    try:
        line = [l for l in open(os.path.join(src, 'bench_fetch.json')) if l.startswith('{')][-1]
        meta = json.loads(line)
        # full-size steps the pass ran: prepare() (the plan-building first step) + warm-up + timed; a --step-only run has no
        # toy-graph step, so every libtipk launch of the pass belongs to one of them
        n_steps = int(meta['steps']) + int(meta['warmup']) + 1
        libtipk = lambda k: not (k.startswith('at::') or k.startswith('rocprim') or 'elementwise' in k or k.startswith('__amd_rocclr'))
        tot_raw = sum(d['hbm_bytes_per_launch_uncorrected'] * max(d.get('launches_FETCH_SIZE', 0), d.get('launches_WRITE_SIZE', 0))
                      for k, d in pmc.items() if libtipk(k))
        step_bytes_raw = tot_raw / n_steps
        tot = sum(d['hbm_bytes_per_launch'] * max(d.get('launches_FETCH_SIZE', 0), d.get('launches_WRITE_SIZE', 0))
                  for k, d in pmc.items() if not (k.startswith('at::') or k.startswith('rocprim') or 'elementwise' in k
                                                  or k.startswith('__amd_rocclr')))          # setup copies / fills
        step_bytes = tot / n_steps
    except Exception as exc:
        print('no per-step total:', exc)
if pmc:
    keep = {k: v for k, v in pmc.items() if not (k.startswith('at::') or k.startswith('rocprim'))}
    top = dict(sorted(keep.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:48])
    json.dump({'note': 'FETCH_SIZE/WRITE_SIZE in KB per launch (separate rocprofv3 --pmc passes of bench.py --launch eager); '
                       'hbm_bytes = (2*FETCH + WRITE)*1024 per MI355X_MICROARCH.md HBM section; keys = kernel + full grid XxYxZ '
                       '(joined from the kernel trace of the same pass by Dispatch_Id); step_hbm_bytes = all libtipk launches '
                       'of the pass / (steps + warmup + 1: the plan-building first step) of `bench.py --launch eager --step-only`; '
                       '*_uncorrected = (FETCH + WRITE)*1024: the factor 2 is stated for wide streaming reads, a row-per-lane '
                       'gather may not need it',
               'build_id': build_id, 'step_hbm_bytes': step_bytes, 'step_hbm_bytes_uncorrected': step_bytes_raw, 'kernels': top},
              open('profiles/%s_pmc_traffic.json' % tag, 'w'), indent=1)

lds = {}
for sub in sorted(glob.glob(os.path.join(src, 'pmc_sq*'))):
    for key, cs in counters_by_launch(os.path.basename(sub)).items():
        if not any(t in key for t in ('rel_gather', 'stream_gather', 'pair_product', 'pair_grads', 'gather_sum', 'dy_products', 'node_products', 'rgcn_')):
            continue
        for c, v in cs.items():
            lds.setdefault(key, {})[c] = sum(v) / len(v)
if lds:
    for key, d in lds.items():
        if d.get('SQ_LDS_IDX_ACTIVE'):
            d['lds_bank_conflict_share'] = d.get('SQ_LDS_BANK_CONFLICT', 0.0) / d['SQ_LDS_IDX_ACTIVE']
        if d.get('SQ_WAVE_CYCLES'):
            d['wait_any_share'] = d.get('SQ_WAIT_ANY', 0.0) / d['SQ_WAVE_CYCLES']
            d['wait_inst_lds_share'] = d.get('SQ_WAIT_INST_LDS', 0.0) / d['SQ_WAVE_CYCLES']
    json.dump({'note': 'rocprofv3 --pmc SQ/LDS counters per launch (averages; each counter group collected in its own pass of '
                       'tools/pmc_relgather.py); SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array '
                       'cycles, SQ_WAIT_* / SQ_WAVE_CYCLES in quad-cycles', 'kernels': lds},
              open('profiles/%s_lds.json' % tag, 'w'), indent=1)
print('summaries written for', tag)
