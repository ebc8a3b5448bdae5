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
#include <stdlib.h>
#include "common.h"
#include "conv_x3.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"

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
    int xcd;             // gemm_x3_kernel: XCD-aware tile order
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

// ---- the same product on the bf16 matrix cores: every fp32 operand split exactly into three bf16 terms, six
// v_mfma_f32_32x32x16_bf16 per fp32 product tile (the arithmetic of conv_x3.hip: dropped terms < 2^-24 of a product).
// 64 x 64 block tile, K step 64, 4 waves as 2 x 2 (wave tile 32 x 32).  LDS images are bf16 planes:
//   operand contiguous along k in memory   -> [row][k]  (pitch 144 B): a fragment = one ds_read_b128, conflict-free
//   operand contiguous along m / n         -> [k][row]  (pitch 192 B): the fragment is read TRANSPOSED with two
//                                             ds_read_b64_tr_b16 (4 k x 16 rows per 16 lanes, returned k-major per lane)
// so that both memory layouts are staged with 16-byte global loads and 8-byte LDS stores, no shuffles.
typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gx_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gx_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int gx_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) gx_bf16x4 gx_lds_bf16x4;
constexpr int BKX = 32;              // K step: 32 keeps the images at 32 KB -> four workgroups per CU (the products here are
                                     // latency-bound: few tiles, short K slices)

// (round 4: two values per conversion / residual instruction - x3_split2, conv_x3.h; the same bits as the scalar form)
__device__ __forceinline__ void gx_split4(const f32x4& v, gx_u32x2 (&out)[3]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned h, m, l;
        x3_split2(x3_f32x2{v[2 * j], v[2 * j + 1]}, h, m, l);
        out[0][j] = h; out[1][j] = m; out[2][j] = l;
    }
}

template <bool KC>           // KC: the operand is k-contiguous ([row][k] image), else [k][row]
struct GxImg {
    static constexpr int PITCH = KC ? (BKX * 2 + 16) : (64 * 2 + 64);
    static constexpr int ROWS = KC ? 64 : BKX;            // 64 rows of the tile, or the k of one step
    static constexpr int BYTES = PITCH * ROWS;            // one plane
};

// DEPTH: K steps whose global loads are in flight in registers ahead of the one being multiplied.  The deep variant (3) takes
// 16-byte loads only (vecA && vecB, K range a multiple of 4) and loads UNCONDITIONALLY - out-of-range groups read a valid stand-in
// address and are zeroed when they are stored to LDS - so that the compiler can count the loads in flight (a branch around a load
// forces s_waitcnt vmcnt(0)).  Measured (round 3, profiles/r03_step_trace_no_overlap.txt): -9 ... +7 % per product, -3 us per
// step - these products are bound neither by load latency nor by the operand split but by the L2 / Infinity-Cache traffic of
// 64 x 64 tiles ((64 + 64) K operand elements per 64 * 64 * K multiplications: the 7x7 head's forward moves 205 MB for 32 MB of
// distinct data).  (The one-stage form takes what the 16-byte loads cannot.)
template <bool AK, bool BNC, int DEPTH = 1>
__global__ __launch_bounds__(256, 4) void gemm_x3_kernel(GemmP p) {
    using IA = GxImg<AK>;
    using IB = GxImg<!BNC>;
    constexpr int BM = 64, BN = 64, G4 = BM * BKX / 4 / 256;   // float4 groups per thread and operand
    constexpr int KQX = BKX / 4;                               // float4 groups along k
    extern __shared__ __attribute__((aligned(16))) unsigned char gx_lds[];
    unsigned char* As = gx_lds;                                 // [3 planes][IA::BYTES]
    unsigned char* Bs = gx_lds + 3 * IA::BYTES;                 // [3 planes][IB::BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int cg = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;      // transposed-read roles
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs, so the tiles of
    // one (batch, K slice) - which share their A rows and B columns - used to be spread over all eight L2s.  Workgroup
    // w = (xcd, k) takes tile xcd * (tiles / 8) + k of the (z, y, x) order: one XCD works through whole (batch, K slice) planes.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd) {
        const unsigned lin = (unsigned)xcd_tile(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x * gridDim.y * gridDim.z);
        bx = lin % gridDim.x; by = (lin / gridDim.x) % gridDim.y; bz = lin / (gridDim.x * gridDim.y);
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int batch = bz / p.splitk, split = bz % p.splitk;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const float* A = p.A + (long)batch * p.sAb;
    const float* B = p.B + (long)batch * p.sBb;
    float* C = p.C + (long)batch * p.sCb + ((p.flags & 8) ? (long)split * p.sCsplit : 0L);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 ra[DEPTH][G4], rb[DEPTH][G4];

    // (row, k) coordinates of this thread's float4 group g of a K step starting at k0; the contiguous direction runs fastest
    auto coordA = [&](int g, int k0, int* fast, int* slow) {
        if (AK) { *fast = k0 + (tid % KQX) * 4; *slow = m0 + tid / KQX + g * (256 / KQX); }       // (k, m)
        else    { *fast = m0 + (tid % 16) * 4;  *slow = k0 + tid / 16 + g * 16; }                  // (m, k)
    };
    auto coordB = [&](int g, int k0, int* fast, int* slow) {
        if (BNC) { *fast = n0 + (tid % 16) * 4;  *slow = k0 + tid / 16 + g * 16; }                 // (n, k)
        else     { *fast = k0 + (tid % KQX) * 4; *slow = n0 + tid / KQX + g * (256 / KQX); }       // (k, n)
    };
    auto okA = [&](int fast, int slow) { return AK ? (fast < kend && slow < p.M) : (fast + 3 < p.M && slow < kend); };
    auto okB = [&](int fast, int slow) { return BNC ? (fast + 3 < p.N && slow < kend) : (fast < kend && slow < p.N); };

    auto gload = [&](int st, int k0) {
#pragma unroll
        for (int g = 0; g < G4; ++g) {
            if constexpr (DEPTH > 1) {
                int f, sl;
                coordA(g, k0, &f, &sl);
                const float* sa = AK ? A + (long)sl * p.sAm + f : A + (long)sl * p.sAk + f;
                ra[st][g] = *reinterpret_cast<const f32x4*>(okA(f, sl) ? sa : A);
                coordB(g, k0, &f, &sl);
                const float* sb = BNC ? B + (long)sl * p.sBk + f : B + (long)sl * p.sBn + f;
                rb[st][g] = *reinterpret_cast<const f32x4*>(okB(f, sl) ? sb : B);
                continue;
            }
            if (AK) {
                const int kq = tid % KQX, row = tid / KQX + g * (256 / KQX);
                const int m = m0 + row, k = k0 + kq * 4;
                const float* src = A + (long)m * p.sAm + (long)k * p.sAk;
                ra[st][g] = p.vecA ? load4<true>(src, p.sAk, k, kend, m < p.M) : load4<false>(src, p.sAk, k, kend, m < p.M);
            } else {
                const int mq = tid % 16, kr = tid / 16 + g * 16;
                const int m = m0 + mq * 4, k = k0 + kr;
                const float* src = A + (long)m * p.sAm + (long)k * p.sAk;
                ra[st][g] = p.vecA ? load4<true>(src, p.sAm, m, p.M, k < kend) : load4<false>(src, p.sAm, m, p.M, k < kend);
            }
            if (BNC) {
                const int nq = tid % 16, kr = tid / 16 + g * 16;
                const int n = n0 + nq * 4, k = k0 + kr;
                const float* src = B + (long)k * p.sBk + (long)n * p.sBn;
                rb[st][g] = p.vecB ? load4<true>(src, p.sBn, n, p.N, k < kend) : load4<false>(src, p.sBn, n, p.N, k < kend);
            } else {
                const int kq = tid % KQX, col = tid / KQX + g * (256 / KQX);
                const int n = n0 + col, k = k0 + kq * 4;
                const float* src = B + (long)k * p.sBk + (long)n * p.sBn;
                rb[st][g] = p.vecB ? load4<true>(src, p.sBk, k, kend, n < p.N) : load4<false>(src, p.sBk, k, kend, n < p.N);
            }
        }
    };
    auto lstore = [&](int st, int k0) {
#pragma unroll
        for (int g = 0; g < G4; ++g) {
            // the contiguous direction runs fastest over the threads, as in gload
            const int fa = AK ? tid % KQX : tid % 16, sa_ = AK ? tid / KQX + g * (256 / KQX) : tid / 16 + g * 16;
            const int fb = BNC ? tid % 16 : tid % KQX, sb_ = BNC ? tid / 16 + g * 16 : tid / KQX + g * (256 / KQX);
            f32x4 va = ra[st][g], vb = rb[st][g];
            if constexpr (DEPTH > 1) {               // stand-in loads of out-of-range groups: exact zeros
                int f, sl;
                coordA(g, k0, &f, &sl);
                if (!okA(f, sl)) va = f32x4{0.f, 0.f, 0.f, 0.f};
                coordB(g, k0, &f, &sl);
                if (!okB(f, sl)) vb = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            gx_u32x2 sa[3], sb[3];
            gx_split4(va, sa);
            gx_split4(vb, sb);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                *reinterpret_cast<gx_u32x2*>(As + pl * IA::BYTES + sa_ * IA::PITCH + fa * 8) = sa[pl];
                *reinterpret_cast<gx_u32x2*>(Bs + pl * IB::BYTES + sb_ * IB::PITCH + fb * 8) = sb[pl];
            }
        }
    };
    // fragment of K step ks (16 k) for the wave's 32 rows starting at r0
    auto frag = [&](const unsigned char* img, bool kc, int pitch, int r0, int ks) -> gx_bf16x8 {
        if (kc)
            return __builtin_bit_cast(gx_bf16x8, *reinterpret_cast<const gx_u32x4*>(img + (r0 + l31) * pitch + (ks * 16 + half * 8) * 2));
        const unsigned char* q = img + (ks * 16 + half * 8 + q4) * pitch + (r0 + cg * 16 + pp * 4) * 2;
        const gx_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((gx_lds_bf16x4*)q);
        const gx_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((gx_lds_bf16x4*)(q + 4 * pitch));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    auto multiply = [&]() {
#pragma unroll
        for (int ks = 0; ks < BKX / 16; ++ks) {
            gx_bf16x8 a[3], b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[pl] = frag(As + pl * IA::BYTES, AK, IA::PITCH, wm0, ks);
                b[pl] = frag(Bs + pl * IB::BYTES, !BNC, IB::PITCH, wn0, ks);
            }
            constexpr int APL[6] = {0, 2, 1, 0, 1, 0}, BPL[6] = {2, 0, 1, 1, 0, 0};      // small products first
#pragma unroll
            for (int t = 0; t < 6; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[APL[t]], b[BPL[t]], acc, 0, 0, 0);
        }
    };
    if constexpr (DEPTH > 1) {
        // prologue: DEPTH steps in flight (steps beyond kend load the stand-in address: harmless, never stored)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) gload(d, kbeg + d * BKX);
        // Round 4: the rounds of DEPTH full steps run in a loop WITHOUT branches, the 0 ... DEPTH-1 remaining steps behind it.  With
        // `if (kk < kend)` around every step the loop header was a join of paths with different numbers of loads in flight, and the
        // compiler put s_waitcnt vmcnt(0) in front of the first stage's LDS stores: every third K step waited for the loads that
        // had just been issued for three steps ahead - the deep prefetch was one step deep (tools/e4_probe.py: 37 us for the 7x7
        // head's product, MFMA busy 0.20, and nothing inside the step mattered).
        const int nst = (kend - kbeg + BKX - 1) / BKX;
        int k0 = kbeg;
        for (int it = 0; it < nst / DEPTH; ++it, k0 += DEPTH * BKX) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int kk = k0 + d * BKX;
                __syncthreads();                         // previous tile's fragment reads are done
                lstore(d, kk);
                __syncthreads();
                gload(d, kk + DEPTH * BKX);              // stage d is free again: three steps ahead, under the MFMAs below
                multiply();
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {            // (the loads issued for steps beyond kend read the stand-in address)
            const int kk = k0 + d * BKX;
            if (kk < kend) {                             // block-uniform
                __syncthreads();
                lstore(d, kk);
                __syncthreads();
                multiply();
            }
        }
    } else {
        if (kbeg < kend) gload(0, kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BKX) {
            __syncthreads();                 // previous tile's fragment reads are done
            lstore(0, k0);
            __syncthreads();
            if (k0 + BKX < kend) gload(0, k0 + BKX);   // in flight under the MFMAs below
            multiply();
        }
    }

    // epilogue (as gemm_kernel): D[i][j]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool lead = (split == 0) && !(p.flags & 8);
    const int n = n0 + wn0 + l31;
    if (n >= p.N) return;
    const float bn = (p.bias_mode == 1 && lead) ? p.bias[n / p.bias_div] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.M) continue;
        float v = acc[r] + bn;
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

template <bool AK, bool BNC, int DEPTH>
int launch_x3_variant(const GemmP& p, int batch, hipStream_t st) {
    constexpr int LDS = 3 * (GxImg<AK>::BYTES + GxImg<!BNC>::BYTES);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<AK, BNC, DEPTH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(cdiv(p.N, 64), cdiv(p.M, 64), batch * p.splitk), block(256);
    hipLaunchKernelGGL((gemm_x3_kernel<AK, BNC, DEPTH>), grid, block, LDS, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

template <bool AK, bool BNC>
int launch_x3_ab(const GemmP& p, int batch, hipStream_t st) {
    // deep prefetch: 16-byte loads on both operands, K range a multiple of 4 (kchunk is a multiple of 32)
    const bool deep = p.vecA && p.vecB && p.K % 4 == 0
                      && (AK || p.M % 4 == 0) && (!BNC || p.N % 4 == 0);
    if (deep) return launch_x3_variant<AK, BNC, 3>(p, batch, st);
    return launch_x3_variant<AK, BNC, 1>(p, batch, st);
}

int launch_x3(const GemmP& p0, int batch, hipStream_t st) {
    GemmP p = p0;
    p.xcd = 1;                                                 // XCD-aware tile order
    const bool ak = (p.sAk == 1), bnc = (p.sBn == 1);
    if (ak && bnc) return launch_x3_ab<true, true>(p, batch, st);
    if (ak) return launch_x3_ab<true, false>(p, batch, st);
    if (bnc) return launch_x3_ab<false, true>(p, batch, st);
    return launch_x3_ab<false, false>(p, batch, st);
}

inline bool gemm_x3_on(int K) {                               // jvae_conv2d_set_split(0): the fp32 matrix-core kernel
    return K >= 64 && jvae_conv5_x3_enabled();
}

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
    if (gemm_x3_on(K)) return launch_x3(p, batch, st);
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
    if (gemm_x3_on(K)) return launch_x3(p, batch, st);
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

// The same for partial products stored [slice][position q][image n][channel cs] - rows of cs: the product's lanes run along cs, so its
// stores are 128-byte segments; in the OUTPUT's own layout [n][cs][q] they were 4-byte stores 4 * Ps bytes apart (26 MB of
// partial-sector writes for 6.5 MB of partial products, ~6 of the 36 us of the 7x7 head's forward product) - folded into
// y[n][cs][q] = bias[cs] + sum_s part[s][q][n][cs]: coalesced reads, the small output written with the stride.
__global__ __launch_bounds__(256) void splitk_fold_qn_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                             float* __restrict__ y, int S, int N, int Cs, int Ps) {
    const long total = (long)Ps * N * Cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cs = (int)(i % Cs);
        const long r = i / Cs;
        const int n = (int)(r % N), q = (int)(r / N);
        float v = bias ? bias[cs] : 0.f;
        for (int s = 0; s < S; ++s) v += part[(long)s * total + i];
        y[((long)n * Cs + cs) * Ps + q] = v;
    }
}

int jvae_splitk_fold_qn(const float* part, const float* bias, float* y, int S, int N, int Cs, int Ps, hipStream_t st) {
    const long total = (long)Ps * N * Cs;
    if (total == 0) return 0;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(splitk_fold_qn_kernel, dim3(blocks), dim3(256), 0, st, part, bias, y, S, N, Cs, Ps);
    JVAE_LAUNCH_CHECK();
    return 0;
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
