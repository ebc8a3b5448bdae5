// GPU box: checks half_wave_reduce32 (common.h) lane by lane against a host sum.
// Build (here): hipcc -O3 --offload-arch=gfx950 tests/diagnostics/reduce32_probe.hip -o joint-vae_amd/csrc/build/reduce32_probe ; run it through gpurun.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../joint-vae_amd/csrc/common.h"
__global__ void k(const float* in, float* out) {
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = in[threadIdx.x * 32 + i];
    out[threadIdx.x] = half_wave_reduce32(v);
}
int main() {
    float h[64 * 32], o[64];
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) h[l * 32 + i] = (float)((l * 37 + i * 101) % 97) + 1000.f * i;
    float *di, *dout;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float e = 0; const int i = l & 31;
        for (int m = 0; m < 32; ++m) e += h[((l & 32) + m) * 32 + i];
        if (e != o[l]) { ++bad; printf("lane %d: got %.1f expected %.1f\n", l, o[l], e); }
    }
    printf("%s\n", bad ? "MISMATCH" : "reduce32 ok");
    return bad != 0;
}
