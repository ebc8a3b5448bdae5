"""Build container: reconstructs the run-to-run NONdeterministic staging variant of round 4 (deferred-BatchNorm coefficients read
from the LDS table as 16-byte vectors inside the per-lane `live` branch; profiles/NOTES.md) on the FIRST 4-phase kernel
(convt2_x3_kernel, still in the tree as the JVAE_T2_V1=1 form) and three variations of it, each as a complete library
joint-vae_amd/jvae_hip/libjvae_nd_{A,B,C,D}.so for tools/nd_probe.py (JVAE_HIP_LIB=... JVAE_T2_V1=1):
  A  vector reads inside `if (AFF && live)`, table declared `__shared__ float ctab[512]` (the reconstruction)
  B  A with the table declared aligned(16)
  C  A with the reads hoisted out of the per-lane branch (unconditional)
  D  A with an explicit s_waitcnt lgkmcnt(0) behind the four reads
usage: python tools/nd_variants.py   (needs joint-vae_amd/csrc/build/*.o of a finished `make`)"""
import glob, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(REPO, 'joint-vae_amd', 'csrc')
src = open(os.path.join(CS, 'conv_t2_x3.hip')).read()
OLD = """                f32x2 vv[8];
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    f32x2 v = (live && kb * 16 + h * 8 + ci < p.C) ? rx[k][ci] : f32x2{0.f, 0.f};
                    if (AFF && live) {
                        const int ch = kb * 16 + h * 8 + ci;
                        const float sc = ctab[ch], sh = ctab[256 + ch];
"""
NEW = """                f32x2 vv[8];
                f32x4 c0, c1, h0, h1;
                if (AFF && live) {
                    const int ch0 = kb * 16 + h * 8;
                    c0 = *reinterpret_cast<const f32x4*>(&ctab[ch0]); c1 = *reinterpret_cast<const f32x4*>(&ctab[ch0 + 4]);
                    h0 = *reinterpret_cast<const f32x4*>(&ctab[256 + ch0]); h1 = *reinterpret_cast<const f32x4*>(&ctab[256 + ch0 + 4]);
                    /*WAIT*/
                }
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    f32x2 v = (live && kb * 16 + h * 8 + ci < p.C) ? rx[k][ci] : f32x2{0.f, 0.f};
                    if (AFF && live) {
                        const float sc = ci < 4 ? c0[ci & 3] : c1[ci & 3], sh = ci < 4 ? h0[ci & 3] : h1[ci & 3];
"""
assert OLD in src
A = src.replace(OLD, NEW)
variants = {'A': A,
            'B': A.replace('    __shared__ float ctab[AFF ? 2 * 256 : 1];', '    __shared__ __attribute__((aligned(16))) float ctab[AFF ? 2 * 256 : 4];'),
            'C': A.replace('                if (AFF && live) {\n                    const int ch0', '                if (AFF) {\n                    const int ch0'),
            'D': A.replace('/*WAIT*/', 'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')}
objs = [o for o in glob.glob(os.path.join(CS, 'build', '*.o')) if 'stamps_' not in o and not o.endswith('conv_t2_x3.o') and '/nd_' not in o]
for k, text in variants.items():
    assert text != src and (k == 'A' or text != A), k
    hip = os.path.join(CS, 'build', f'nd_{k}.hip')
    open(hip, 'w').write(text)
    obj = hip[:-4] + '.o'
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '--offload-arch=gfx950', '-std=c++17', '-I' + os.path.join(REPO, 'include'),
                    '-I' + CS, '-Wno-unused-function', '-c', hip, '-o', obj], check=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '--offload-arch=gfx950'] + objs + [obj, '-o',
                    os.path.join(REPO, 'joint-vae_amd', 'jvae_hip', f'libjvae_nd_{k}.so')], check=True)
    print('built', k)
