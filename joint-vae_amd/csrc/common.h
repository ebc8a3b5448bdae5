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

// Tile of workgroup w when the n workgroups of a launch are dealt round-robin to the 8 XCDs of an MI355X (hardware order:
// workgroup w runs on XCD w % 8): XCD x works through the CONTIGUOUS tile range [x * n / 8 ...), so tiles that share data
// (halo rows of one image) meet in one XCD's L2.  A bijection of [0, n) for every n.  JVAE_XCD_MAP=0 at build time: identity.
#ifndef JVAE_XCD_MAP
#define JVAE_XCD_MAP 1
#endif
__device__ __forceinline__ int xcd_tile(unsigned w, unsigned n) {
#if JVAE_XCD_MAP
    const unsigned per = n >> 3, rem = n & 7, x = w & 7, k = w >> 3;
    return (int)(x * per + (x < rem ? x : rem) + k);
#else
    return (int)w;
#endif
}

// Images [nb, ne) of part j when N images are dealt to `parts` workgroups in runs of ceil(N / parts).  The last parts can be
// EMPTY (nb >= N whenever (parts - 1) * ceil(N / parts) >= N, e.g. N = 49 over 64 parts): ne is clamped to nb so that ne - nb is
// never negative - every kernel that partitions a batch takes its range from here (and only from here).
struct ImageRange { int nb, ne; };
__host__ __device__ __forceinline__ ImageRange image_range(int N, int parts, int j) {
    const int per = (N + parts - 1) / parts;
    const long b = (long)j * per;
    ImageRange r;
    r.nb = b < N ? (int)b : N;
    const long e = b + per;
    r.ne = e < N ? (int)e : N;
    if (r.ne < r.nb) r.ne = r.nb;
    return r;
}

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

// Transposing reduction over the 16 lanes of each DPP row (lanes 16r .. 16r+15): every lane brings 16 values a[0..15];
// afterwards lane l holds the sum, over the 16 lanes of its row, of a[l & 15].  Steps 2-5 of half_wave_reduce32 (see there).
__device__ __forceinline__ float row_reduce16(const float (&a)[16]) {
    const int lane = threadIdx.x & 63;
    float b[8], c[4], d[2];
#pragma unroll
    for (int i = 0; i < 8; i += 4) {    // lane bit 3 (partner lane ^ 8 = row_ror:8): banks 0,1 keep a[i], banks 2,3 a[i + 8]
#pragma unroll
        for (int j = 0; j < 4; ++j) b[i + j] = a[i + j];
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc"
            : "+v"(b[i]), "+v"(b[i + 1]), "+v"(b[i + 2]), "+v"(b[i + 3])
            : "v"(a[i + 8]), "v"(a[i + 9]), "v"(a[i + 10]), "v"(a[i + 11]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = b[j];
    // lane bit 2 (partner = row_half_mirror, lane 7 - l of each 8): banks 0,2 keep b[i], banks 1,3 b[i + 4]
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
        : "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    const bool b1 = lane & 2, b0 = lane & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {       // lane bit 1 (partner lane ^ 2)
        const float keep = b1 ? c[i + 2] : c[i], send = b1 ? c[i] : c[i + 2];
        d[i] = keep + jvae_dpp<0x4E, 0xf>(send);
    }
    const float keep = b0 ? d[1] : d[0], send = b0 ? d[0] : d[1];
    return keep + jvae_dpp<0xB1, 0xf>(send);      // lane bit 0 (partner lane ^ 1)
}

// Transposing reduction over the 32 lanes of each half-wave: every lane brings 32 values v[0..31]; afterwards lane l
// holds the sum, over the 32 lanes of its half-wave, of v[l & 31].  A value-halving butterfly - at every step a lane
// keeps half of its values and hands the other half to its partner - needs 16 + 8 + 4 + 2 + 1 additions (65 vector
// instructions) where 32 independent all-lane reductions need 160, and leaves ONE value per lane (one LDS store instead
// of 32 single-lane ones).  Step 1 crosses the rows of 16 lanes with gfx950's v_permlane16_swap, steps 2 / 3 use DPP adds
// whose bank mask restricts the write to the lanes that keep the value, steps 4 / 5 (inside a quad, where no write mask
// exists) select with v_cndmask.  All lanes must be active.  Fixed order: results are run-to-run identical.
__device__ __forceinline__ float half_wave_reduce32(const float (&v)[32]) {
    float a[16];
    // s_nop 1: a vector write needs two wait states before a DPP / permlane-swap read of the same register (inline asm is
    // not covered by the compiler's hazard recogniser).  The swap is written as asm because this compiler's
    // __builtin_amdgcn_permlane16_swap returns its first result twice.
#pragma unroll
    for (int i = 0; i < 16; i += 4) {   // lane bit 4: even rows keep v[i], odd rows v[i + 16]
        float x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = v[i + j]; y[j] = v[i + j + 16]; }
        // odd rows of x <-> even rows of y: afterwards x + y = (x.row0 + x.row1 | y.row0 + y.row1 | x.row2 + x.row3 | ...)
        asm("s_nop 1\n\t"
            "v_permlane16_swap_b32 %0, %4\n\t"
            "v_permlane16_swap_b32 %1, %5\n\t"
            "v_permlane16_swap_b32 %2, %6\n\t"
            "v_permlane16_swap_b32 %3, %7"
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i + j] = x[j] + y[j];
    }
    return row_reduce16(a);
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

// Activation kinds of the `relu` / `in_relu` arguments of the C ABI (include/jvae_hip.h): 0 = none, 1 = ReLU, 2 = leaky ReLU with
// PyTorch's default negative slope - the reference's activation='leaky' is nn.LeakyReLU() (module/vae_layers/misc.py:24-27).
// max(x, slope * x) is x for x > 0 and slope * x otherwise: the same product and the same bits as torch's select.
#define JVAE_ACT_NONE 0
#define JVAE_ACT_RELU 1
#define JVAE_ACT_LEAKY 2
#define JVAE_LEAKY_SLOPE 0.01f
__host__ __device__ __forceinline__ int jvae_act_kind(int relu) { return relu == JVAE_ACT_LEAKY ? JVAE_ACT_LEAKY : (relu ? JVAE_ACT_RELU : JVAE_ACT_NONE); }
__device__ __forceinline__ float jvae_act(float t, int kind) {
    return kind == JVAE_ACT_RELU ? fmaxf(t, 0.f) : (kind == JVAE_ACT_LEAKY ? fmaxf(t, JVAE_LEAKY_SLOPE * t) : t);
}

// a = [relu](v*s + t) on a float4 (deferred BatchNorm of a convolution input; the same fmaf as bn_coef / bn_apply_kernel,
// so the ReLU mask BatchNorm-backward recomputes is the one applied here)
__device__ __forceinline__ f32x4 aff4(f32x4 v, float s, float t, int relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float x = fmaf(v[j], s, t); v[j] = relu ? fmaxf(x, 0.f) : x; }
    return v;
}
// ... with leaky ReLU instead (the AFF = 2 instantiations of the convolution kernels: the ReLU ones keep their code)
__device__ __forceinline__ f32x4 aff4_leaky(f32x4 v, float s, float t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float x = fmaf(v[j], s, t); v[j] = fmaxf(x, JVAE_LEAKY_SLOPE * x); }
    return v;
}

// ... with the activation kind at run time (the leaky-ReLU instantiations of kernels whose two operand sides may differ)
__device__ __forceinline__ f32x4 aff4_kind(f32x4 v, float s, float t, int kind) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = jvae_act(fmaf(v[j], s, t), kind);
    return v;
}

// the same transform on one B8 unit (8 bf16 channels of a pixel): a = [relu](x*sc[c] + sh[c]), rounded back to bf16
typedef __bf16 jvae_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int jvae_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ jvae_u32x4 aff8(jvae_u32x4 u, const float* sc8, const float* sh8, int relu) {
    // sc8 / sh8: 16-byte aligned LDS tables - four 16-byte reads instead of sixteen dependent 4-byte ones (round 4: config 5 bf16
    // 4.29 -> 4.15 ms).  Wide reads of such a table are safe; what was not (rounds 4-5, profiles/NOTES.md) is the packed fp32 FMA the
    // compiler may form behind them with the HIGH register of a coefficient pair selected for its LOW result half - the build fails
    // if any kernel contains that form (tools/isa_opsel_scan.py, run by the Makefile); here each product is a scalar FMA on a
    // converted bf16 value, and no such instruction exists.
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(sc8), c1 = *reinterpret_cast<const f32x4*>(sc8 + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh8), h1 = *reinterpret_cast<const f32x4*>(sh8 + 4);
    jvae_bf16x8 v = __builtin_bit_cast(jvae_bf16x8, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = fmaf((float)v[j], j < 4 ? c0[j & 3] : c1[j & 3], j < 4 ? h0[j & 3] : h1[j & 3]);
        v[j] = (__bf16)(relu ? fmaxf(x, 0.f) : x);
    }
    return __builtin_bit_cast(jvae_u32x4, v);
}
