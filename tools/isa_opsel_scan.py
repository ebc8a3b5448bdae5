"""Build container: scan the gfx950 code of the built objects (csrc/build/*.o) for the instruction form behind the run-to-run
nondeterminisms of rounds 4-5 (profiles/NOTES.md, round 5): a packed fp32 vector instruction - v_pk_fma_f32 / v_pk_add_f32 /
v_pk_mul_f32 - whose LOW result half selects the HIGH register of a VGPR source pair (op_sel bit set on a vector-register source).
On MI355X that half intermittently (one 16-lane pass in ~1e5 executions, under MFMA / memory traffic of other waves) comes out as
if the selected register were 0.  The compiler emits the form when two values it wants to broadcast sit in adjacent registers
(coefficient quadruples read as 16-byte vectors; SLP-vectorised scalar code).  Scalar-register sources with op_sel (s_load-ed
coefficients of conv_smallco.hip) take another operand path, have been bit-stable since round 4 and are listed, not flagged.
usage: python tools/isa_opsel_scan.py [OBJ ...]      exit code 1 if any flagged instruction exists"""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
PK = re.compile(r'\b(v_pk_(?:fma|add|mul)_f32)\s+(.*)$')


def device_isa(obj):
    """disassembly of the gfx950 code object bundled into a host object built by hipcc"""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'k.co')
        r = subprocess.run(['objcopy', '--dump-section', '.hip_fatbin=' + fat, obj], capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return ''                                        # a host-only object
        subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--input=' + fat, '--output=' + co], check=True, capture_output=True)
        return subprocess.run([LLVM + '/llvm-objdump', '-d', co], check=True, capture_output=True, text=True).stdout


def scan(text):
    """(flagged, scalar_sources): instruction texts; flagged = op_sel bit i set where source i is a vector register"""
    flagged, scalar, kernel = [], [], '?'
    for line in text.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(\w+)>:', line)
        if m:
            kernel = m.group(1)
            continue
        line = line.split('//')[0].strip()
        m = PK.search(line)
        if not m:
            continue
        sel = re.search(r'op_sel:\[([01,]+)\]', m.group(2))
        if not sel:
            continue
        ops = [o.strip() for o in re.split(r'\s+op_sel', m.group(2))[0].split(',')][1:]           # sources
        for bit, src in zip(sel.group(1).split(','), ops):
            if bit == '1':
                (flagged if src.startswith('v') else scalar).append((kernel, m.group(1) + ' ' + m.group(2)))
                break
    return flagged, scalar


def main(objs):
    import shutil
    if not (os.path.exists(LLVM + '/llvm-objdump') and os.path.exists(LLVM + '/clang-offload-bundler') and shutil.which('objcopy')):
        print('isa_opsel_scan: llvm-objdump / clang-offload-bundler / objcopy not found - scan skipped')
        return 0
    objs = objs or sorted(glob.glob(os.path.join(REPO, 'joint-vae_amd', 'csrc', 'build', '*.o')))
    with ThreadPoolExecutor(8) as ex:
        texts = list(ex.map(device_isa, objs))
    bad = 0
    for obj, text in zip(objs, texts):
        flagged, scalar = scan(text)
        n = len(re.findall(r'\bv_pk_(?:fma|add|mul)_f32\b', text))
        print(f'{os.path.basename(obj):22s} packed fp32 instructions {n:5d}   high-half select on a scalar pair {len(scalar):4d}   on a VECTOR pair {len(flagged)}')
        for k, ins in flagged[:8]:
            print('    ', k[:60], '|', ins)
        bad += len(flagged)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
