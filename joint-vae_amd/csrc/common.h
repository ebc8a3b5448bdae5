// Shared device/host helpers for libjvae_hip.so (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JVAE_WAVE 64

// Every extern "C" entry point returns 0 on success, <0 for an invalid argument, >0 = hipError_t.
#define JVAE_EINVAL (-1)
#define JVAE_ENOTSUP (-2)
#define JVAE_EWORKSPACE (-3)

#define JVAE_LAUNCH_CHECK()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread.
// `red` must hold >= 17 floats of LDS.  Contains barriers: call from uniform control flow.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (w == 0) {
        float t = lane < nw ? red[lane] : 0.f;
        t = wave_sum(t);
        if (lane == 0) red[16] = t;
    }
    __syncthreads();
    return red[16];
}

// a = [relu](v*s + t) on a float4 (deferred BatchNorm of a convolution input; the same fmaf as bn_coef / bn_apply_kernel,
// so the ReLU mask BatchNorm-backward recomputes is the one applied here)
__device__ __forceinline__ f32x4 aff4(f32x4 v, float s, float t, int relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float x = fmaf(v[j], s, t); v[j] = relu ? fmaxf(x, 0.f) : x; }
    return v;
}

// the same transform on one B8 unit (8 bf16 channels of a pixel): a = [relu](x*sc[c] + sh[c]), rounded back to bf16
typedef __bf16 jvae_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int jvae_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ jvae_u32x4 aff8(jvae_u32x4 u, const float* sc8, const float* sh8, int relu) {
    jvae_bf16x8 v = __builtin_bit_cast(jvae_bf16x8, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = fmaf((float)v[j], sc8[j], sh8[j]);
        v[j] = (__bf16)(relu ? fmaxf(x, 0.f) : x);
    }
    return __builtin_bit_cast(jvae_u32x4, v);
}
