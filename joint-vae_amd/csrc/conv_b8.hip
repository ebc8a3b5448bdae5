// bf16 activation path (config 5 of BASELINE.json: "bf16 activations / weights, fp32 master weights, fp32 loss math").
//
// Layout "B8": an activation tensor (N, C, H, W) is stored as (N, CB = ceil(C/8), H, W, 8) bf16 - eight consecutive
// channels of one pixel form one 16-byte unit, units of one channel block are in NCHW order.  Padding channels are 0.
// The unit is exactly one lane's operand of v_mfma_f32_32x32x16_bf16 (8 consecutive K values), so both the global
// loads (16 B per lane, consecutive pixels) and the LDS fragment reads (one ds_read_b128 per operand, any tap shift
// stays 16-byte aligned) are the natural ones; the accumulator tile (pixel on the lane, 4 consecutive channels in 4
// registers) is written back as 8-byte half units, 512 contiguous bytes per store instruction.
//
// This file: fp32 NCHW <-> B8 converters, the weight re-pack (fp32 master -> bf16 operand layout) and the
// "forward-type" 5x5 kernel, which serves Conv2d forward (S = 1, 2), ConvTranspose2d stride-1 forward, Conv2d
// stride-1 dgrad and ConvTranspose2d dgrad exactly as conv_mfma.hip does for fp32 (same weight roles: swap / flip).
// Accumulation, bias and the BatchNorm partial statistics stay fp32.
#include "common.h"
#include "jvae_internal.h"
#include "conv_b8.h"
#include "pack_elems.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// converters
// ---------------------------------------------------------------------------------------------------------------
// y[n][cb][q][ci] = bf16(x[n][cb*8+ci][q]);  one thread per unit, consecutive threads = consecutive pixels
__global__ __launch_bounds__(256) void b8_pack_kernel(const float* __restrict__ x, bf16x8* __restrict__ y,
                                                      int N, int C, int CB, long HW) {
    const long total = (long)N * CB * HW;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const long q = u % HW;
        const long t = u / HW;
        const int cb = (int)(t % CB);
        const long n = t / CB;
        bf16x8 v;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = cb * 8 + ci;
            v[ci] = (__bf16)(c < C ? x[(n * C + c) * HW + q] : 0.f);
        }
        y[u] = v;
    }
}

// x[n][c][q] (+)= float(y[n][c/8][q][c%8])
__global__ __launch_bounds__(256) void b8_unpack_kernel(const bf16x8* __restrict__ y, float* __restrict__ x,
                                                        int N, int C, int CB, long HW, int accumulate) {
    const long total = (long)N * CB * HW;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const long q = u % HW;
        const long t = u / HW;
        const int cb = (int)(t % CB);
        const long n = t / CB;
        const bf16x8 v = y[u];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = cb * 8 + ci;
            if (c < C) {
                float* dst = x + (n * C + c) * HW + q;
                *dst = (accumulate ? *dst : 0.f) + (float)v[ci];
            }
        }
    }
}

// partial[cb*8+ci][s] = sum over the images of split s and all pixels of t[n][cb][q][ci]   (bias gradients)
__global__ __launch_bounds__(256) void b8_channel_sum_kernel(const bf16x8* __restrict__ t, float* __restrict__ partial,
                                                             int N, int CB, long HW, int nsplit) {
    __shared__ float red[4][8];
    const int cb = blockIdx.x, s = blockIdx.y;
    const ImageRange ir = image_range(N, nsplit, s);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long cnt = (long)(ne - nb) * HW;
    for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
        const long n = nb + i / HW, q = i % HW;
        const bf16x8 v = t[(n * CB + cb) * HW + q];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) acc[ci] += (float)v[ci];
    }
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const float w = wave_sum(acc[ci]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][ci] = w;
    }
    __syncthreads();
    if (threadIdx.x < 8)
        partial[((long)cb * 8 + threadIdx.x) * nsplit + s] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void b8_channel_fold_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int nsplit,
                                       int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(long)c * nsplit + k];
    out[c] = (accumulate ? out[c] : 0.f) + s;
}

// ---------------------------------------------------------------------------------------------------------------
// weights: Wp[kb][tap][half][o][ci] = bf16( W[o][c = kb*16 + half*8 + ci][tap] )   (o < OP, zero padding)
// swap: source is [c][o][tap] (ConvTranspose2d layout / role swap), flip: tap -> 24 - tap
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void b8_wpack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp,
                                                       int C, int O, int KB, int OP, int swap, int flip) {
    const long total = (long)KB * 25 * 2 * OP * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        jvae_pack_b8_elem(w, wp, i, C, O, swap, flip);
}

// ---------------------------------------------------------------------------------------------------------------
// forward-type kernel
// ---------------------------------------------------------------------------------------------------------------
struct B8FwdP {
    const u32x4* in;     // B8 units (N, CBin, H, W)
    const u32x4* wp;     // packed weight units (KB, 25, 2, OP)
    const float* bias;   // (CoutReal) or null
    void* out;           // B8 units (N, CBout, OH, OW), or float (N, CoutReal, OH, OW) when out_f32
    int N, CBin, H, W, OP, P, CoutReal, CBout;
    float* stats;        // optional (CoutReal, gridDim.x, 2): per-workgroup sum / sum of squares of (out - bias), fp32
    int out_f32;
    InAff aff;           // deferred BatchNorm(+ReLU) of the input (sc == nullptr: none); CBin*8 coefficients
};

template <int S, int OW, int MT, int NT>
struct B8Geom {
    static constexpr int OH = OW;
    static constexpr int PIX = MT * 128;
    static constexpr int OHW = OH * OW;
    static constexpr int NIMG = PIX >= OHW ? PIX / OHW : 1;
    static constexpr int TH = PIX >= OHW ? OH : PIX / OW;
    static constexpr int ROWS = (TH - 1) * S + 5;
    static constexpr int WIN = OW * S;
    static constexpr int WP0 = (OW - 1) * S + 9;
    static constexpr int WP1 = WIN + 4;
    static constexpr int WP = WP0 > WP1 ? WP0 : WP1;          // units per patch row
    static constexpr int CH = ROWS * WP;                       // units per channel block per image
    static constexpr int XS = NIMG * 2 * CH;                   // patch units of one K step (16 channels)
    static constexpr int WCOLS = NT * 32;
    static constexpr int WS = 25 * 2 * WCOLS;                  // weight units of one K step
    static constexpr int LDS_BYTES = (XS + WS) * 16;
};

template <int S, int OW, int MT, int NT, bool AFF>
__global__ __launch_bounds__(256, 2) void conv5_b8_kernel(B8FwdP p) {
    using G = B8Geom<S, OW, MT, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Xs = reinterpret_cast<u32x4*>(lds_raw);
    u32x4* Ws = Xs + G::XS;
    __shared__ __attribute__((aligned(16))) float ctab[AFF ? 2 * 256 : 4];   // (scale, shift) of all input channels (<= 256)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (AFF)
        for (int i = tid; i < p.CBin * 8; i += 256) { ctab[i] = p.aff.sc[i]; ctab[256 + i] = p.aff.sh[i]; }
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int TILES_PER_IMG = G::OHW >= G::PIX ? G::OHW / G::PIX : 1;
    const int img0 = (G::OHW >= G::PIX) ? (int)(blockIdx.x / TILES_PER_IMG) : (int)blockIdx.x * G::NIMG;
    const int row0 = (G::OHW >= G::PIX) ? (int)(blockIdx.x % TILES_PER_IMG) * G::TH : 0;
    const int o0 = blockIdx.y * G::WCOLS;

    // halo columns / out-of-image rows / missing images are zeroed once and never written again
    for (int i = tid; i < G::XS; i += 256) Xs[i] = u32x4{0u, 0u, 0u, 0u};

    int pixoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pix = (wave * MT + mt) * 32 + l31;
        const int im = pix / (G::TH * OW), rem = pix % (G::TH * OW);
        const int r = rem / OW, c = rem % OW;
        pixoff[mt] = im * (2 * G::CH) + half * G::CH + (r * S) * G::WP + c * S + 4 - p.P;
    }

    f32x16 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int in_row0 = row0 * S - p.P;
    constexpr int XUNITS = G::NIMG * 2 * G::ROWS * G::WIN;      // interior units per K step
    constexpr int XU = (XUNITS + 255) / 256, WU = (G::WS + 255) / 256;
    u32x4 rx[XU], rw[WU];
    const int KB = (p.CBin + 1) / 2;

    auto gload = [&](int kb) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            const int x = u % G::WIN;
            int t = u / G::WIN;
            const int lr = t % G::ROWS; t /= G::ROWS;
            const int h = t % 2, im = t / 2;
            const int ir = in_row0 + lr, n = img0 + im, cb = kb * 2 + h;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (u < XUNITS && ir >= 0 && ir < p.H && n < p.N && cb < p.CBin)
                v = p.in[(((long)n * p.CBin + cb) * p.H + ir) * p.W + x];
            rx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * 256;
            const int col = u % G::WCOLS, th = u / G::WCOLS;      // th = tap*2 + half
            u32x4 v = {0u, 0u, 0u, 0u};
            if (u < G::WS) v = p.wp[((long)kb * 50 + th) * p.OP + o0 + col];
            rw[k] = v;
        }
    };
    auto lstore = [&](int kb) {                    // kb: the K step whose data sits in rx
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            if (u < XUNITS) {
                const int x = u % G::WIN;
                int t = u / G::WIN;
                const int lr = t % G::ROWS; t /= G::ROWS;
                const int h = t % 2, im = t / 2;
                u32x4 v = rx[k];
                if (AFF) {                         // deferred BatchNorm(+ReLU) of the input; padding cells stay zero
                    const int ir = in_row0 + lr, n = img0 + im, cb = kb * 2 + h;
                    if (ir >= 0 && ir < p.H && n < p.N && cb < p.CBin) v = aff8(v, &ctab[cb * 8], &ctab[256 + cb * 8], p.aff.relu);
                }
                Xs[(im * 2 + h) * G::CH + lr * G::WP + 4 + x] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * 256;
            if (u < G::WS) Ws[u] = rw[k];
        }
    };

    gload(0);
    for (int kb = 0; kb < KB; ++kb) {
        __syncthreads();
        lstore(kb);
        __syncthreads();
        if (kb + 1 < KB) gload(kb + 1);
        // one-deep software pipeline: the fragments of tap t+1 are read from LDS before the MFMAs of tap t are issued
        // (a 32-cycle bf16 MFMA leaves no slack for an LDS round trip in front of it)
        u32x4 fa[2][NT], fb[2][MT];
        auto frag = [&](int tap, u32x4 (&a)[NT], u32x4 (&b)[MT]) {
            const int kh = tap / 5, kw = tap % 5;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a[nt] = Ws[(tap * 2 + half) * G::WCOLS + nt * 32 + l31];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = Xs[pixoff[mt] + kh * G::WP + kw];
        };
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            if (tap + 1 < 25) frag(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[tap & 1][nt]),
                                                                          __builtin_bit_cast(bf16x8, fb[tap & 1][mt]),
                                                                          acc[nt][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- optional BatchNorm statistics (fp32, before rounding to bf16)
    if (p.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds_raw);       // [4 waves][WCOLS][2]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float sv[32];                                     // [sum | sum of squares][register row]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { const float v = acc[nt][mt][r]; s1 += v; s2 += v * v; }
                sv[r] = s1;
                sv[16 + r] = s2;
            }
            // lane l31 receives the half-wave total of sv[l31]
            const float tot = half_wave_reduce32(sv);
            const int r = l31 & 15, ch = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            red[(wave * G::WCOLS + ch) * 2 + (l31 >> 4)] = tot;
        }
        __syncthreads();
        if (tid < G::WCOLS && o0 + tid < p.CoutReal) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[(w * G::WCOLS + tid) * 2]; s2 += red[(w * G::WCOLS + tid) * 2 + 1]; }
            float* dst = p.stats + ((long)(o0 + tid) * gridDim.x + blockIdx.x) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }

    // ---- epilogue: lane holds pixel l31 of each 32-pixel group, rows (channels) (r&3) + 8*(r>>2) + 4*half
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pix = (wave * MT + mt) * 32 + l31;
        const int im = pix / (G::TH * OW), rem = pix % (G::TH * OW);
        const int n = img0 + im;
        if (n >= p.N) continue;
        const int oy = row0 + rem / OW, ox = rem % OW;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ob = o0 + nt * 32 + 8 * rg + 4 * half;       // first of this lane's 4 channels
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[nt][mt][rg * 4 + e];
                    if (p.bias && ob + e < p.CoutReal) v[e] += p.bias[ob + e];
                }
                if (p.out_f32) {
                    float* out = reinterpret_cast<float*>(p.out);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ob + e < p.CoutReal) out[(((long)n * p.CoutReal + ob + e) * G::OH + oy) * OW + ox] = v[e];
                } else {
                    const int cb = ob >> 3;
                    if (cb < p.CBout) {
                        const bf16x4 q = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        u32x2* dst = reinterpret_cast<u32x2*>(p.out) +
                                     ((((long)n * p.CBout + cb) * G::OH + oy) * OW + ox) * 2 + half;
                        *dst = __builtin_bit_cast(u32x2, q);
                    }
                }
            }
    }
}

thread_local int g_b8_splits = 0;

template <int S, int OW, int MT, int NT>
int launch_b8(const B8FwdP& p, hipStream_t st) {
    using G = B8Geom<S, OW, MT, NT>;
    static_assert(G::LDS_BYTES + 2048 <= 80 * 1024, "two workgroups per CU must fit the 160 KB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_b8_kernel<S, OW, MT, NT, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_b8_kernel<S, OW, MT, NT, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long pixels = (long)p.N * G::OHW;
    dim3 grid((unsigned)((pixels + G::PIX - 1) / G::PIX), (unsigned)(p.OP / G::WCOLS));
    if (G::OHW < G::PIX) grid.x = (unsigned)((p.N + G::NIMG - 1) / G::NIMG);
    g_b8_splits = (int)grid.x;
    if (p.aff.sc) {
        if (p.CBin * 8 > 256) return JVAE_ENOTSUP;
        hipLaunchKernelGGL((conv5_b8_kernel<S, OW, MT, NT, true>), grid, dim3(256), G::LDS_BYTES, st, p);
    } else {
        hipLaunchKernelGGL((conv5_b8_kernel<S, OW, MT, NT, false>), grid, dim3(256), G::LDS_BYTES, st, p);
    }
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool jvae_conv5_b8_fwd_ok(int Cin, int H, int W, int Cout, int OH, int OW, int S, int P) {
    if (S != 1 && S != 2) return false;
    if (OH != OW || H != W || W != OW * S) return false;
    if (S == 1 && OW != 4 && OW != 8 && OW != 16 && OW != 32 && OW != 64) return false;
    if (S == 2 && OW != 4 && OW != 8 && OW != 16 && OW != 32) return false;
    if (P < 0 || P > 4) return false;
    if ((OW - 1) * S + 4 - P >= W + 4) return false;
    return Cin >= 1 && Cout >= 1;
}

size_t jvae_conv5_b8_pack_bytes(int Cin, int Cout) {
    return (size_t)((Cin + 15) / 16) * 25 * 2 * ((Cout + 31) / 32 * 32) * 16;
}

int jvae_conv5_b8_max_splits(int N, int OW) { return (int)(((long)N * OW * OW + 127) / 128) + 1; }

int jvae_b8_pack(const float* x, void* y, int N, int C, long HW, hipStream_t st) {
    const int CB = (C + 7) / 8;
    const long total = (long)N * CB * HW;
    if (total == 0) return 0;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(b8_pack_kernel, dim3(blocks), dim3(256), 0, st, x, (bf16x8*)y, N, C, CB, HW);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_b8_unpack(const void* y, float* x, int N, int C, long HW, int accumulate, hipStream_t st) {
    const int CB = (C + 7) / 8;
    const long total = (long)N * CB * HW;
    if (total == 0) return 0;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(b8_unpack_kernel, dim3(blocks), dim3(256), 0, st, (const bf16x8*)y, x, N, C, CB, HW, accumulate);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// out[c] (+)= sum_{n,q} t[n][c][q]; ws: ceil(C/8)*8*64 floats
int jvae_b8_channel_sum(const void* t, float* out, int N, int C, long HW, int accumulate, float* ws, hipStream_t st) {
    const int CB = (C + 7) / 8;
    int ns = (int)((long)N * HW / 4096);
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    if (ns > N) ns = N > 0 ? N : 1;
    hipLaunchKernelGGL(b8_channel_sum_kernel, dim3(CB, ns), dim3(256), 0, st, (const bf16x8*)t, ws, N, CB, HW, ns);
    JVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(b8_channel_fold_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, (const float*)ws, out, C, ns, accumulate);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_conv5_b8_wpack(const float* w, void* wp, int C, int O, int swap, int flip, hipStream_t st) {
    const int KB = (C + 15) / 16, OP = (O + 31) / 32 * 32;
    const long total = (long)KB * 25 * 2 * OP * 8;
    const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(b8_wpack_kernel, dim3(blocks), dim3(256), 0, st, w, (__bf16*)wp, C, O, KB, OP, swap, flip);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// in: B8 (N, ceil(Cin/8), H, W); out: B8 (N, ceil(Cout/8), OW, OW) or fp32 NCHW (out_f32).  ws: packed weights.
int jvae_conv5_b8_fwd(const void* in, const float* w, int swap, int flip, const float* bias, void* out, int out_f32,
                      int N, int Cin, int H, int W, int Cout, int OW, int S, int P, void* ws, hipStream_t st,
                      float* stats, int* nsplit, const InAff* aff) {
    {
        bool fresh = true;
        void* slot = jvae_pack_cache_get(JVAE_PACK_B8, w, Cin, Cout, swap, flip, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            int rc = jvae_conv5_b8_wpack(w, ws, Cin, Cout, swap, flip, st);
            if (rc) return rc;
        }
    }
    B8FwdP p{(const u32x4*)in, (const u32x4*)ws, bias, out, N, (Cin + 7) / 8, H, W, (Cout + 31) / 32 * 32, P,
             Cout, (Cout + 7) / 8, stats, out_f32, aff ? *aff : InAff{nullptr, nullptr, 0}};
    struct Fin { int* n; ~Fin() { if (n) *n = g_b8_splits; } } fin{nsplit};
    if (S == 1) {
        switch (OW) {
            case 4: return launch_b8<1, 4, 1, 1>(p, st);      // 4x4 maps (deconv32+): 8 images per workgroup
            case 8: return launch_b8<1, 8, 2, 1>(p, st);
            case 16: return launch_b8<1, 16, 4, 1>(p, st);
            case 32: return launch_b8<1, 32, 4, 1>(p, st);
            case 64: return launch_b8<1, 64, 4, 1>(p, st);
        }
    } else {
        switch (OW) {
            case 4: return launch_b8<2, 4, 1, 1>(p, st);
            case 8: return launch_b8<2, 8, 1, 1>(p, st);
            case 16: return launch_b8<2, 16, 2, 1>(p, st);
            case 32: return launch_b8<2, 32, 2, 1>(p, st);
        }
    }
    return JVAE_ENOTSUP;
}
