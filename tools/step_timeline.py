"""Timeline of ONE replayed step from a rocprofv3 kernel trace of bench.py (hipGraph replay):
   python tools/step_timeline.py gpurun_out/prof_<tag>/trace  -> kernel, grid, duration, gap; totals by kernel class."""
import csv, glob, os, sys
root = sys.argv[1]
f = sorted(glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ANCHOR = sys.argv[2] if len(sys.argv) > 2 else 'drug_mix_gather_fwd_kernel'   # a kernel that is launched once per step
fw = [i for i, r in enumerate(rows) if ANCHOR in r['Kernel_Name']]
steps = [(a, b) for a, b in zip(fw, fw[1:]) if b - a > 10]          # skip the back-to-back launch timing loops
per = {}
for a, b in steps:
    per[b - a] = per.get(b - a, 0) + 1
period = max(per, key=per.get)                                      # the replayed step (most frequent distance)
steps = [(a, b) for a, b in steps if b - a == period]
i0, i1 = steps[len(steps) // 2]
t0 = int(rows[i0]['Start_Timestamp'])
prev = None
small = big = 0.0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:46]
    dur = (e - s) / 1e3
    print('%8.1f  dur %6.1f  gap %5.1f  %-46s %sx%s' % ((s - t0) / 1e3, dur, ((s - prev) / 1e3 if prev else 0), name,
                                                     r['Grid_Size_X'], r['Grid_Size_Y']))
    if dur < 16.5 and 'rel_gather' not in name and 'rel_stream' not in name and 'dy_products' not in name and 'gemm_stream' not in name:
        small += dur
    else:
        big += dur
    prev = e
print('step span %.1f us: %d kernels; small kernels %.1f us, large %.1f us' % (
    (int(rows[i1]['Start_Timestamp']) - t0) / 1e3, i1 - i0, small, big))
