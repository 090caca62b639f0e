"""Print the kernel timeline of one replayed step from a rocprofv3 kernel trace of bench.py.
usage: python3 tools/timeline.py <dir with *kernel_trace.csv> [marker substring]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else 'transpose_kernel'
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); return n[:52]
# a step starts at the first transpose (W1^T) of the forward pass: every second transpose launch
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']][::2]
a, b = idx[-6], idx[-5]
t0 = int(rows[a]['Start_Timestamp']); tot = 0
for r in rows[a:b]:
    st = int(r['Start_Timestamp']) - t0; d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot += d
    print('%8.1f  %6.1f  %s  grid=%sx%sx%s' % (st / 1e3, d / 1e3, short(r['Kernel_Name']), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']))
print('kernels', b - a, 'span us', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'sum of durations', tot / 1e3)
