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

// Sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63) on the vector ALU's data-parallel-primitive path - no
// LDS traffic, unlike __shfl_xor (ds_bpermute): four steps inside each row of 16 lanes, then row_bcast15 adds lane 15
// of row 0 (2) into row 1 (3).  The total is valid in lanes 16-31 and 48-63 ONLY (JVAE_HALF_SUM_LANE = 31 reads it).
#define JVAE_HALF_SUM_LANE 31
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float jvae_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float half_wave_sum_hi(float v) {
    v += jvae_dpp<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += jvae_dpp<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v += jvae_dpp<0x141, 0xf>(v);     // row_half_mirror
    v += jvae_dpp<0x140, 0xf>(v);     // row_mirror
    v += jvae_dpp<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3 (rows 0 and 2 add the 0 of `old`)
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
