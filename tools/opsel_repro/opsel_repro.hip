// Stand-alone reproducer attempt for the packed-fp32 operand-select hazard of profiles/NOTES.md (round 5).
// Every wave runs: 16-byte LDS reads of a small coefficient table (few addresses, many lanes), global loads in flight, then
//   FORM 1: v_pk_fma_f32 d, x, c, h op_sel:[0,1,1]          (low half takes the HIGH register of the c / h pairs)
//   FORM 0: v_pk_fma_f32 d, x, c, h op_sel_hi:[1,0,0]       (high half takes the LOW register: the form used everywhere)
// and compares both halves with scalar v_fma_f32 results.  Half of the waves of a workgroup keep the matrix pipe and the LDS busy.
// build: hipcc -O3 --offload-arch=gfx950 opsel_repro.hip -o opsel_repro ; run: ./opsel_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ x, const float* __restrict__ coef, float* __restrict__ sink,
                                                unsigned* __restrict__ bad, int iters, int n) {
    __shared__ __attribute__((aligned(16))) float ctab[512];
    __shared__ __attribute__((aligned(16))) unsigned short fill[8192];
    const int tid = threadIdx.x;
    ctab[tid] = coef[tid]; ctab[256 + tid] = coef[256 + tid];
    for (int i = tid; i < 8192; i += 256) fill[i] = (unsigned short)(i * 2654435761u >> 16);
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    if (wave >= 2) {                                   // two waves: MFMA + LDS traffic, as the convolution kernels have around their staging
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters * 4; ++it) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&fill[((it * 64 + lane) * 8) & 8184]);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&fill[((it * 64 + lane) * 8 + 4096) & 8184]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
        return;
    }
    unsigned nbad = 0;
    float keep = 0.f;
    const long base = ((long)blockIdx.x * 128 + (wave * 64 + lane)) * 2;
    for (int it = 0; it < iters; ++it) {
        const int h = (it + (lane >> 4)) & 31;                                    // 4 distinct table addresses per wave (16 lanes each)
        const f32x2 xv = *reinterpret_cast<const f32x2*>(&x[(base + (long)it * 4099 * 2) % (n - 2) & ~1L]);   // a load in flight
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ctab[h * 8]);
        const f32x4 s = *reinterpret_cast<const f32x4*>(&ctab[256 + h * 8]);
        f32x2 cp = {c[0], c[1]}, sp = {s[0], s[1]}, d;
        if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" : "=v"(d) : "v"(xv), "v"(cp), "v"(sp));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(xv), "v"(cp), "v"(sp));
        const float k = FORM == 1 ? c[1] : c[0], t = FORM == 1 ? s[1] : s[0];
        float e0, e1;
        asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5" : "=&v"(e0), "=&v"(e1) : "v"(xv[0]), "v"(xv[1]), "v"(k), "v"(t));
        nbad += (d[0] != e0) + 2 * (d[1] != e1 ? 1 : 0) * 65536u / 2;            // low 16 bits: low-half mismatches, high: high-half
        keep += d[0] + d[1];
        fill[(tid * 8 + it) & 8191] = (unsigned short)it;                         // LDS stores behind it, as in the staging code
    }
    sink[blockIdx.x * 256 + tid] = keep;
    if (nbad) atomicAdd(&bad[FORM * 2 + 0], nbad & 0xffffu), atomicAdd(&bad[FORM * 2 + 1], nbad >> 16);
}

int main() {
    const int n = 1 << 24, blocks = 2048, iters = 2000;
    std::vector<float> hx(n), hc(512);
    for (int i = 0; i < n; ++i) hx[i] = 1.f + (i % 977) * 0.001f;
    for (int i = 0; i < 512; ++i) hc[i] = 1.f + i;
    float *x, *c, *sink; unsigned* bad;
    hipMalloc(&x, n * 4); hipMalloc(&c, 512 * 4); hipMalloc(&sink, blocks * 256 * 4); hipMalloc(&bad, 16);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(bad, 0, 16);
        hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, x, c, sink, bad, iters, n);
        hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, x, c, sink, bad, iters, n);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned h[4]; hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        const double execs = (double)blocks * 2 * iters;
        printf("rep %d: %.3g wave-instructions per form | op_sel_hi:[1,0,0] wrong lanes low %u high %u | op_sel:[0,1,1] wrong lanes low %u high %u\n",
               rep, execs, h[0], h[1], h[2], h[3]);
    }
    return 0;
}
