// fp32 MFMA GEMM for gfx950:  C[b](m,n) (+)= sum_k A[b](m,k) * B[b](k,n)  (+ bias, ReLU)
//
// v_mfma_f32_32x32x2_f32: exact fp32 (k-ordered fmaf chain), 64 FLOP/clk/SIMD.  256-thread workgroups
// (4 waves as 2x2), block tile BMxBN, K step 16; operands are staged global -> registers -> LDS with the
// next tile's global loads in flight under the current tile's MFMAs.  LDS images are k-major
// (As[k][m], Bs[k][n]) so that every fragment read is 32 consecutive dwords per half-wave (conflict-free
// ds_read_b32, MI355X_MICROARCH.md §LDS).  Arbitrary element strides on A, B and C let one kernel serve
// NN / NT / TN products, per-image batched products (grid.z) and split-K with float-atomic accumulation.
//
// Serves: Linear layers fwd/dgrad/wgrad (reference: nn.Linear in layers.py:283-296, cvae.py:291-326),
// the 1x1 -> kxk first transposed conv of the upsampler, and the generic (im2col) convolution path.
#include "common.h"
#include "jvae_internal.h"

namespace {

constexpr int BK = 32;          // 64 MFMAs (128x128 tile) between two barriers
constexpr int KQ = BK / 4;      // float4 groups along k

struct GemmP {
    const float* A; long sAm, sAk, sAb;
    const float* B; long sBk, sBn, sBb;
    float* C; long sCm, sCn, sCb;
    const float* bias;   // nullptr or per-n (mode 1) / per-m (mode 2)
    int bias_mode;
    int bias_div;        // mode 1: bias[n / bias_div] (transposed conv on a 1x1 input: channel = column / (KH*KW))
    int M, N, K;
    int splitk;          // grid.z = batch * splitk
    int kchunk;          // K elements per split (multiple of BK)
    int flags;           // 1 = accumulate into C, 2 = ReLU, 4 = atomic add (split-K), 8 = split s stores to C + s*sCsplit
    long sCsplit;        // (flag 8) element distance between the partial results of consecutive K splits
    int vecA, vecB;      // 16-byte vector loads are legal for this operand
};

// 4 consecutive elements along the contiguous direction, zero-filled outside [0,lim).
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* p, long stride, int i0, int lim, bool row_ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!row_ok) return v;
    if (VEC) {
        if (i0 + 3 < lim) return *reinterpret_cast<const f32x4*>(p);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j < lim) v[j] = p[j * stride];
    return v;
}

// AK: A is contiguous along k (row-major MxK); otherwise thread groups run along m.
// BN_: B is contiguous along n (row-major KxN); otherwise along k.
template <int BM, int BN, bool AK, bool BNC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int WM = BM / 2, WN = BN / 2;          // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;        // 32x32 MFMA tiles per wave
    constexpr int GA = BM * BK / 4 / 256;            // float4 groups per thread (A)
    constexpr int GB = BN * BK / 4 / 256;
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int batch = blockIdx.z / p.splitk, split = blockIdx.z % p.splitk;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const float* A = p.A + (long)batch * p.sAb;
    const float* B = p.B + (long)batch * p.sBb;
    float* C = p.C + (long)batch * p.sCb + ((p.flags & 8) ? (long)split * p.sCsplit : 0L);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[GA], rb[GB];

    auto gload = [&](int k0) {
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            if (AK) {
                const int kq = tid % KQ, row = tid / KQ + g * (256 / KQ);
                const int m = m0 + row, k = k0 + kq * 4;
                const float* src = A + (long)m * p.sAm + (long)k * p.sAk;
                ra[g] = p.vecA ? load4<true>(src, p.sAk, k, kend, m < p.M) : load4<false>(src, p.sAk, k, kend, m < p.M);
            } else {
                constexpr int GPR = BM / 4;                 // groups per k-row
                const int mq = tid % GPR, kr = tid / GPR + g * (256 / GPR);
                const int m = m0 + mq * 4, k = k0 + kr;
                const float* src = A + (long)m * p.sAm + (long)k * p.sAk;
                ra[g] = p.vecA ? load4<true>(src, p.sAm, m, p.M, k < kend) : load4<false>(src, p.sAm, m, p.M, k < kend);
            }
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            if (BNC) {
                constexpr int GPR = BN / 4;
                const int nq = tid % GPR, kr = tid / GPR + g * (256 / GPR);
                const int n = n0 + nq * 4, k = k0 + kr;
                const float* src = B + (long)k * p.sBk + (long)n * p.sBn;
                rb[g] = p.vecB ? load4<true>(src, p.sBn, n, p.N, k < kend) : load4<false>(src, p.sBn, n, p.N, k < kend);
            } else {
                const int kq = tid % KQ, col = tid / KQ + g * (256 / KQ);
                const int n = n0 + col, k = k0 + kq * 4;
                const float* src = B + (long)k * p.sBk + (long)n * p.sBn;
                rb[g] = p.vecB ? load4<true>(src, p.sBk, k, kend, n < p.N) : load4<false>(src, p.sBk, k, kend, n < p.N);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            if (AK) {
                const int kq = tid % KQ, row = tid / KQ + g * (256 / KQ);
#pragma unroll
                for (int j = 0; j < 4; ++j) As[(kq * 4 + j) * LDA + row] = ra[g][j];
            } else {
                constexpr int GPR = BM / 4;
                const int mq = tid % GPR, kr = tid / GPR + g * (256 / GPR);
                *reinterpret_cast<f32x4*>(&As[kr * LDA + mq * 4]) = ra[g];
            }
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            if (BNC) {
                constexpr int GPR = BN / 4;
                const int nq = tid % GPR, kr = tid / GPR + g * (256 / GPR);
                *reinterpret_cast<f32x4*>(&Bs[kr * LDB + nq * 4]) = rb[g];
            } else {
                const int kq = tid % KQ, col = tid / KQ + g * (256 / KQ);
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[(kq * 4 + j) * LDB + col] = rb[g][j];
            }
        }
    };

    const int half = lane >> 5, l31 = lane & 31;
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                 // previous tile's fragment reads are done
        lstore();
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);   // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(2 * kk + half) * LDA + wm0 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * kk + half) * LDB + wn0 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: D[i][j]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool lead = (split == 0) && !(p.flags & 8);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            if (n >= p.N) continue;
            float bn = (p.bias_mode == 1 && lead) ? p.bias[n / p.bias_div] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bn;
                if (p.bias_mode == 2 && lead) v += p.bias[m];
                float* dst = C + (long)m * p.sCm + (long)n * p.sCn;
                if (p.flags & 8) {
                    *dst = v;
                } else if (p.flags & 4) {
                    atomicAdd(dst, v);
                } else {
                    if (p.flags & 1) v += *dst;
                    if (p.flags & 2) v = fmaxf(v, 0.f);
                    *dst = v;
                }
            }
        }
}

template <int BM, int BN>
int launch_tile(const GemmP& p, int batch, hipStream_t st) {
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), batch * p.splitk), block(256);
    const bool ak = (p.sAk == 1), bnc = (p.sBn == 1);
    if (ak && bnc)       hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true>), grid, block, 0, st, p);
    else if (ak && !bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false>), grid, block, 0, st, p);
    else if (!ak && bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, false, true>), grid, block, 0, st, p);
    else                 hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false>), grid, block, 0, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int jvae_gemm_launch(int M, int N, int K, int batch,
                     const float* A, long sAm, long sAk, long sAb,
                     const float* B, long sBk, long sBn, long sBb,
                     float* C, long sCm, long sCn, long sCb,
                     const float* bias, int bias_mode, int flags, int splitk, hipStream_t st) {
    return jvae_gemm_launch_ex(M, N, K, batch, A, sAm, sAk, sAb, B, sBk, sBn, sBb, C, sCm, sCn, sCb, bias, bias_mode, 1,
                               flags, splitk, st);
}

// Internal entry used by the other translation units (see jvae_internal.h).
int jvae_gemm_launch_ex(int M, int N, int K, int batch,
                        const float* A, long sAm, long sAk, long sAb,
                        const float* B, long sBk, long sBn, long sBb,
                        float* C, long sCm, long sCn, long sCb,
                        const float* bias, int bias_mode, int bias_div, int flags, int splitk, hipStream_t st) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (K < 0 || !A || !B || !C) return JVAE_EINVAL;
    if (bias_mode && !bias) return JVAE_EINVAL;
    GemmP p;
    p.A = A; p.sAm = sAm; p.sAk = sAk; p.sAb = sAb;
    p.B = B; p.sBk = sBk; p.sBn = sBn; p.sBb = sBb;
    p.C = C; p.sCm = sCm; p.sCn = sCn; p.sCb = sCb;
    p.bias = bias; p.bias_mode = bias_mode; p.bias_div = bias_div > 0 ? bias_div : 1;
    p.M = M; p.N = N; p.K = K;
    if (splitk < 1) splitk = 1;
    int ktiles = cdiv(K, BK);
    if (splitk > ktiles) splitk = ktiles > 0 ? ktiles : 1;
    p.kchunk = cdiv(ktiles, splitk) * BK;
    splitk = K > 0 ? cdiv(K, p.kchunk) : 1;
    p.splitk = splitk;
    p.flags = flags & ~8;
    p.sCsplit = 0;
    if (splitk > 1) {
        if (flags & 2) return JVAE_EINVAL;        // ReLU cannot follow a partial sum
        p.flags |= 4;                             // caller pre-zeroes C (or wants accumulation)
    }
    // vector loads: contiguous direction has unit stride, everything else keeps 16-byte alignment
    const bool ak = (sAk == 1), bnc = (sBn == 1);
    p.vecA = aligned16(A) && (sAb % 4 == 0) && (ak ? (sAm % 4 == 0) : (sAm == 1 && sAk % 4 == 0));
    p.vecB = aligned16(B) && (sBb % 4 == 0) && (bnc ? (sBk % 4 == 0) : (sBk == 1 && sBn % 4 == 0));
    const long tiles128 = (long)cdiv(M, 128) * cdiv(N, 128) * batch * splitk;
    if (M > 64 && N > 64 && tiles128 >= 512) return launch_tile<128, 128>(p, batch, st);
    return launch_tile<64, 64>(p, batch, st);
}

// Deterministic split-K: K is cut into *splits pieces whose partial products are STORED side by side (part + s*sCsplit,
// same strides as C); no atomics, no bias.  Fold them with jvae_splitk_fold.  Returns the number of pieces in *splits.
int jvae_gemm_launch_part(int M, int N, int K, int batch,
                          const float* A, long sAm, long sAk, long sAb,
                          const float* B, long sBk, long sBn, long sBb,
                          float* part, long sCm, long sCn, long sCb, long sCsplit, int want_splits, int* splits,
                          hipStream_t st) {
    if (splits) *splits = 0;
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (K <= 0 || !A || !B || !part || !splits) return JVAE_EINVAL;
    GemmP p;
    p.A = A; p.sAm = sAm; p.sAk = sAk; p.sAb = sAb;
    p.B = B; p.sBk = sBk; p.sBn = sBn; p.sBb = sBb;
    p.C = part; p.sCm = sCm; p.sCn = sCn; p.sCb = sCb;
    p.bias = nullptr; p.bias_mode = 0; p.bias_div = 1;
    p.M = M; p.N = N; p.K = K;
    int splitk = want_splits < 1 ? 1 : want_splits;
    const int ktiles = cdiv(K, BK);
    if (splitk > ktiles) splitk = ktiles;
    p.kchunk = cdiv(ktiles, splitk) * BK;
    splitk = cdiv(K, p.kchunk);
    p.splitk = splitk;
    p.flags = 8;
    p.sCsplit = sCsplit;
    const bool ak = (sAk == 1), bnc = (sBn == 1);
    p.vecA = aligned16(A) && (sAb % 4 == 0) && (ak ? (sAm % 4 == 0) : (sAm == 1 && sAk % 4 == 0));
    p.vecB = aligned16(B) && (sBb % 4 == 0) && (bnc ? (sBk % 4 == 0) : (sBk == 1 && sBn % 4 == 0));
    *splits = splitk;
    return launch_tile<64, 64>(p, batch, st);
}

// y[i] = [relu](bias[i % N] + sum_s part[s][i]) in a fixed order: the deterministic second half of a split-K product
// whose S partial products were written side by side by a batched launch (batch = K slices).
__global__ __launch_bounds__(256) void splitk_fold_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                          float* __restrict__ y, int S, long MN, int N, int relu,
                                                          int accumulate, int bias_div) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += (long)gridDim.x * blockDim.x) {
        float v = bias ? bias[(i / bias_div) % N] : 0.f;
        for (int s = 0; s < S; ++s) v += part[(long)s * MN + i];
        if (accumulate) v += y[i];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

int jvae_splitk_fold(const float* part, const float* bias, float* y, int S, long MN, int N, int relu, int accumulate,
                     hipStream_t st, int bias_div) {
    if (MN == 0) return 0;
    const int blocks = (int)((MN + 255) / 256 > 4096 ? 4096 : (MN + 255) / 256);
    hipLaunchKernelGGL(splitk_fold_kernel, dim3(blocks), dim3(256), 0, st, part, bias, y, S, MN, N, relu, accumulate,
                       bias_div > 0 ? bias_div : 1);
    JVAE_LAUNCH_CHECK();
    return 0;
}

extern "C" int jvae_splitk_fold_f32(const float* part, const float* bias, float* y, int S, long MN, int N, int relu,
                                    int accumulate, void* stream) {
    if (!part || !y || S < 1 || MN < 0 || N < 1) return JVAE_EINVAL;
    return jvae_splitk_fold(part, bias, y, S, MN, N, relu, accumulate, (hipStream_t)stream);
}

extern "C" int jvae_gemm_f32(int M, int N, int K, int batch,
                             const float* A, long sAm, long sAk, long sAb,
                             const float* B, long sBk, long sBn, long sBb,
                             float* C, long sCm, long sCn, long sCb,
                             const float* bias, int bias_mode, int flags, int splitk, void* stream) {
    return jvae_gemm_launch(M, N, K, batch, A, sAm, sAk, sAb, B, sBk, sBn, sBb, C, sCm, sCn, sCb,
                            bias, bias_mode, flags, splitk, (hipStream_t)stream);
}
