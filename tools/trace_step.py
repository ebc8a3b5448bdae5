"""One training step out of a rocprofv3 kernel_trace.csv, in launch order: duration, idle gap in front of the launch (start minus the
latest end of any earlier kernel; negative = overlap), grid, kernel name.
usage: python tools/trace_step.py KERNEL_TRACE.csv   (the step = the dispatches between the last two adam_kernel launches)"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(rows))
tot = gaps = 0.
last_end = max(int(r['End_Timestamp']) for r in rows[:lo]) if lo else None
first = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    gap = (int(r['Start_Timestamp']) - last_end) / 1e3 if last_end is not None else 0.
    if gap > 0:
        gaps += gap
    last_end = max(last_end or 0, int(r['End_Timestamp']))
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    nm = re.sub(r'std::array<char\*, \d+ul>|at::native::|\(anonymous namespace\)::', '', nm)
    nm = (nm if 'elementwise' in nm or 'reduce_kernel' in nm else re.sub(r'\(.*', '', nm))[:150]
    print(f'{d:8.1f} us  gap {gap:6.1f}  grid {r.get("Grid_Size", "?"):>9s}  wg {r.get("Workgroup_Size", "?"):>4s}  {nm}')
print(f'{tot / 1e3:8.3f} ms total, {hi - lo} dispatches; idle gaps between them {gaps / 1e3:.3f} ms; '
      f'span {(last_end - first) / 1e6:.3f} ms (under the profiler)')
