"""One training step out of a rocprofv3 kernel_trace.csv, in launch order: duration, grid, kernel name.
usage: python tools/trace_step.py KERNEL_TRACE.csv   (the step = the dispatches between the last two adam_kernel launches)"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(rows))
tot = 0.
for r in rows[lo:hi]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    nm = re.sub(r'\(.*', '', nm)[:60]
    print(f'{d:8.1f} us  grid {r.get("Grid_Size", "?"):>9s}  wg {r.get("Workgroup_Size", "?"):>4s}  {nm}')
print(f'{tot / 1e3:8.3f} ms total, {hi - lo} dispatches')
