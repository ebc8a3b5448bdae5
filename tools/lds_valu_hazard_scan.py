"""Build container: static scan of the gfx950 ISA of csrc/*.hip (hipcc -S) for the pattern behind round 5's two run-to-run
nondeterminisms: a VGPR written by a wide LDS read (ds_read_b64 / b96 / b128 ...) whose FIRST reader is a vector-ALU instruction (not an
MFMA, not a store) within a few instructions of the s_waitcnt that retires the read.  Prints, per kernel, the closest such
consumer: (instructions between the wait and the consumer, the read, the consumer).
usage: python tools/lds_valu_hazard_scan.py /tmp/isa/*.s"""
import re, sys
WIDE = re.compile(r'ds_read(2?_b64|2st64_b64|_b96|_b128|2_b32|2st64_b32)\s+v\[(\d+):(\d+)\]')
def regs(tok):
    out = set()
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok):
        out.add(int(a))
    return out
for path in sys.argv[1:]:
    s = open(path).read()
    for name in re.findall(r'^(_Z\w+):', s, re.M):
        a = s.index(name + ':'); b = s.find('.Lfunc_end', a)
        lines = [l.strip() for l in s[a:b].split('\n')]
        lines = [l for l in lines if l and not l.startswith(';') and not l.startswith('.')]
        pending = {}          # vgpr -> (index of the read, text)
        last_wait = None
        worst = None
        for i, l in enumerate(lines):
            m = WIDE.match(l)
            if m:
                for r in range(int(m.group(2)), int(m.group(3)) + 1):
                    pending[r] = (i, l)
                continue
            if l.startswith('s_waitcnt') and 'lgkmcnt' in l:
                last_wait = i
                continue
            if not l.startswith('v_') or l.startswith('v_mfma'):
                # any other writer of a pending register ends its life; readers that are not VALU are not of interest here
                ops = l.split(None, 1)
                continue
            ops = l.split(None, 1)[1] if ' ' in l else ''
            parts = ops.split(',')
            dst, srcs = parts[0], ','.join(parts[1:])
            hit = [r for r in regs(srcs) if r in pending]
            if hit and last_wait is not None and last_wait > pending[hit[0]][0]:
                d = i - last_wait
                if worst is None or d < worst[0]:
                    worst = (d, pending[hit[0]][1], l)
            for r in hit:
                pending.pop(r, None)
            for r in regs(dst):
                pending.pop(r, None)
        if worst and worst[0] <= 4:
            print(f'{path.split("/")[-1]:22s} {name[:60]:60s} gap {worst[0]}: {worst[1][:44]} -> {worst[2][:70]}')
