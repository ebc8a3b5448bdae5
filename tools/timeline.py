"""Timeline statistics of a rocprofv3 kernel trace of bench.py: per step (delimited by adam_kernel launches) the wall time,
the time at least one kernel is running (union of the intervals), the idle time inside the step, and the busy time per
HIP stream (queue).  usage: python tools/timeline.py KERNEL_TRACE.csv [first_step last_step]"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name']))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'adam_kernel' in r[3]]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (len(ends) // 2, len(ends) // 2 + 5)
for s in range(lo, min(hi, len(ends) - 1)):
    seg = rows[ends[s] + 1: ends[s + 1] + 1]
    t0, t1 = rows[ends[s]][1], seg[-1][1]
    busy, cur_s, cur_e = 0, None, None
    for a, b, _, _ in seg:
        a = max(a, t0)
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    per_q = {}
    for a, b, q, _ in seg:
        per_q[q] = per_q.get(q, 0) + (b - a)
    gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1]), re.sub(r'\(anonymous namespace\)::', '', seg[i][3])[:40],
                   re.sub(r'\(anonymous namespace\)::', '', seg[i + 1][3])[:40]) for i in range(len(seg) - 1)), reverse=True)[:4]
    print(f'step {s}: wall {(t1 - t0) / 1e3:8.1f} us  any-kernel-busy {busy / 1e3:8.1f} us  idle {(t1 - t0 - busy) / 1e3:7.1f} us  '
          f'kernels {len(seg)}  per queue: ' + ', '.join(f'{q}: {v / 1e3:.0f} us' for q, v in sorted(per_q.items())))
    print('   largest gaps: ' + ' | '.join(f'{g[0] / 1e3:.1f} us after {g[1]} before {g[2]}' for g in gaps if g[0] > 0))
