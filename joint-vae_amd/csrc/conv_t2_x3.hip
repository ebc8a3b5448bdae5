// Stride-2 5x5 transposed convolution (padding 2, output_padding 1: H -> 2H), 4-phase sub-pixel form, on the bf16 matrix
// cores with exact 3-way operand splitting: the arithmetic of conv_x3.hip applied to the operator of conv_t2_mfma.hip.
//
//   big[n][o][2a + r][2b + q] = bias[o] + sum_c sum_{kh = r (mod 2), kw = q (mod 2)}
//                               small[n][c][a + (r + 2 - kh)/2][b + (q + 2 - kw)/2] * W[c][o][kh*5 + kw]
//
// Serves ConvTranspose2d(5, stride 2, padding 2, output_padding 1) forward (imager.6 / imager.12 of deconv32) and the
// dgrad of Conv2d(5, stride 2, padding 2) (features.3 / features.9 of conv32), fp32 NCHW in and out.
//
// Mapping (round 5; the first kernel - 32x32x16 MFMAs, 16-channel K step, weights staged per kernel row - left the tree after
// the A/B of profiles/NOTES.md): see the comment at convt2s_x3_kernel.
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"
#include "conv_x3.h"
#include "pack_elems.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef x3_bf16x8 bf16x8;
typedef x3_u32x4 u32x4;
typedef x3_f32x2 f32x2;

static thread_local int g_t2x3_splits = 0;

struct T2X3P {
    const float* in;     // small (N, C, HS, WS) fp32
    const u32x4* wp;     // split weights, JVAE_PACK_T2S (pack_elems.h): [K step of 32 channels][tap of the position-ordered sequence][plane][kq][o]
    const float* bias;   // (O) or null
    float* out;          // big (N, O, 2HS, 2WS)
    int N, C, O;
    float* stats;        // optional (O, gridDim.x, 2)
    InAff aff;           // deferred BatchNorm(+ReLU) of the input
};

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the 16x16x32 form ("t2s").  What conv_x3.hip's stride-1 kernel gained in round 4, rebuilt for the 4-phase operator:
//  * v_mfma_f32_16x16x32_bf16 with K = 32 CHANNELS of ONE tap (every channel count of the 4-phase layers is a multiple of 32:
//    no tap pairing, no empty half pair); lane group kq = lane >> 4 supplies channel block kq of the K step;
//  * the 25 taps read only the 3 x 3 neighbourhood of a small-grid pixel: they are walked in POSITION order (jvae_t2s_tap,
//    pack_elems.h), so the patch fragments of a position are read once and serve its 4 / 2 / 1 taps (one per output phase): 9
//    position reads + 25 weight reads per 25 taps where the first kernel read 25 + 25;
//  * weights are packed [K step][tap of the sequence][plane][kq][o] and staged per GROUP of 3 (the last two groups: 2) taps -
//    18 KB, double-buffered - i.e. 72 / 48 MFMAs of 16 cycles between two LDS barriers (the first kernel: 30 of 32), every group
//    and tap a compile-time constant: the K step is straight-line code, each fragment read-ahead sits in front of the MFMAs it
//    flies under and the compiler's s_waitcnt counts are exact (conv_x3.hip's npair lesson);
//  * staging with a WAVE-UNIFORM channel block: wave w stages channels 8w .. 8w + 7 of the K step, so the deferred BatchNorm's
//    (scale, shift) are scalar operands (s_load through the constant address space) - no LDS coefficient table, no per-lane
//    dependent ds_read_b32 - and channels beyond C load a clamped valid address and are zeroed;
//  * only the halo COLUMNS of the patch image are cleared (behind the first global loads), the bias sits in LDS from the start,
//    the output stores are issued BEFORE the BatchNorm sums are reduced (they drain under the reduction) and every barrier of the
//    kernel waits for LDS only; tiles are dealt XCD-aware (xcd_tile) so that the two row tiles of an image share their halo in L2.
// Same operator, tensors, BatchNorm partial sums (one slot per tile) and 3-way exact split as the first kernel; the summation
// order differs (32 channels per MFMA, position-ordered taps), so results agree to fp32 rounding, not bit for bit.
template <int WS>
struct T2SGeom {
    static constexpr int HS = WS;
    static constexpr int PIX = 128;
    static constexpr int HSWS = HS * WS;
    static constexpr int NIMG = PIX >= HSWS ? PIX / HSWS : 1;
    static constexpr int TH = PIX >= HSWS ? HS : PIX / WS;
    static constexpr int ROWS = TH + 2;
    static constexpr int WP = WS + 2;                          // units per patch row: data at column 1
    static constexpr int CH = ROWS * WP;                       // units per 8-channel block per image
    static constexpr int XS = NIMG * 4 * CH;                   // patch units of one plane (32 channels)
    static constexpr int WGS = 3 * 3 * 4 * 32;                 // weight units of one group: 3 taps x 3 planes x 4 lane groups x 32 o
    static constexpr int LDS_BYTES = (3 * XS + 2 * WGS) * 16;
};

// tap t of the position-ordered sequence: kernel row / column, output phase, patch offset relative to the centre of the 3 x 3
struct T2STap {
    static constexpr int NG = 9;                               // weight groups per K step
    __host__ __device__ static constexpr int gstart(int g) { return g < 7 ? 3 * g : (g == 7 ? 21 : (g == 8 ? 23 : 25)); }
    __host__ __device__ static constexpr int group(int t) { return t < 21 ? t / 3 : (t < 23 ? 7 : 8); }
    __host__ __device__ static constexpr int tap(int t) {
        constexpr int SEQ[25] = {0, 1, 5, 6,  2, 3, 7, 8,  4, 9,  10, 11, 15, 16,  12, 13, 17, 18,  14, 19,  20, 21,  22, 23,  24};
        return SEQ[t];
    }
    __host__ __device__ static constexpr int kh(int t) { return tap(t) / 5; }
    __host__ __device__ static constexpr int kw(int t) { return tap(t) % 5; }
    __host__ __device__ static constexpr int phase(int t) { return (kh(t) & 1) * 2 + (kw(t) & 1); }
    __host__ __device__ static constexpr int dh(int t) { return ((kh(t) & 1) + 2 - kh(t)) / 2; }
    __host__ __device__ static constexpr int dw(int t) { return ((kw(t) & 1) + 2 - kw(t)) / 2; }
    __host__ __device__ static constexpr int pos(int t) { return (1 - dh(t)) * 3 + (1 - dw(t)); }        // 0 .. 8, non-decreasing in t
};

template <int... I, class F>
__device__ __forceinline__ void t2s_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <int WS, int AFF>      // AFF: 0 = plain input, 1 = deferred BatchNorm (+ReLU by p.aff.relu), 2 = deferred BatchNorm + leaky ReLU
__global__ __launch_bounds__(256, 2) void convt2s_x3_kernel(T2X3P p) {
    using G = T2SGeom<WS>;
    using T = T2STap;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Xs = reinterpret_cast<u32x4*>(lds_raw);            // [3 planes][XS]
    u32x4* Ws = Xs + 3 * G::XS;                                // [2 buffers][WGS]
    __shared__ __attribute__((aligned(16))) float bias_s[32];
    __shared__ float red_s[4 * 32 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int TILES_PER_IMG = G::HSWS >= G::PIX ? G::HSWS / G::PIX : 1;
    const int bx = xcd_tile(blockIdx.x, gridDim.x);
    const int img0 = (G::HSWS >= G::PIX) ? bx / TILES_PER_IMG : bx * G::NIMG;
    const int row0 = (G::HSWS >= G::PIX) ? (bx % TILES_PER_IMG) * G::TH : 0;
    const int o0 = blockIdx.y * 32;
    const int KB = (p.C + 31) / 32;
    const int OP = p.O;                                        // multiple of 32 (jvae_convt2_x3_ok)
    if (tid < 32) bias_s[tid] = p.bias ? p.bias[o0 + tid] : 0.f;

    // two 16-pixel tiles per wave; the lane's pixel of each and the centre of its 3 x 3 neighbourhood in channel block kq
    constexpr int NPT = 2;
    int pixoff[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int pix = (wave * NPT + pt) * 16 + l15;
        const int im = pix / (G::TH * WS), rem = pix % (G::TH * WS);
        pixoff[pt] = (im * 4 + kq) * G::CH + (rem / WS + 1) * G::WP + rem % WS + 1;
    }

    // ---- staging: wave w owns channel block w of every K step; items = (2 pixels x 8 channels)
    constexpr int W2 = WS / 2;
    constexpr int PERB = G::NIMG * G::ROWS * W2;               // items per channel block and K step
    constexpr int XU = (PERB + 63) / 64;
    const int hq = __builtin_amdgcn_readfirstlane(wave);
    f32x2 rx[XU][8];
    // addresses = UNIFORM base (channel plane: scalar registers) + a 32-bit per-lane element offset (the host checks that the
    // tensor has fewer than 2^31 elements): one offset register per item instead of a 64-bit address per load
    unsigned xoff[XU];
    const long cstride = (long)G::HS * WS;
#pragma unroll
    for (int k = 0; k < XU; ++k) {
        const int u = lane + k * 64;
        const int xp = u % W2;
        const int t = u / W2;
        const int lr = t % G::ROWS, im = t / G::ROWS;
        const int ir = row0 - 1 + lr, n = img0 + im;
        const bool ok = u < PERB && ir >= 0 && ir < G::HS && n < p.N;
        xoff[k] = (unsigned)(((ok ? n : 0) * p.C * G::HS + (ok ? ir : 0)) * WS + 2 * xp);
    }
    auto gloadX = [&](int kb) {
        // channels beyond C: the last valid channel is loaded instead and zeroed in lstoreX (uniform address arithmetic: the loads
        // stay unconditional and countable)
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = kb * 32 + hq * 8 + ci;
            const float* ub = p.in + (long)(c < p.C ? c : p.C - 1) * cstride;
#pragma unroll
            for (int k = 0; k < XU; ++k) rx[k][ci] = *reinterpret_cast<const f32x2*>(ub + xoff[k]);
        }
    };
    auto lstoreX = [&](int kb) {
        typedef const __attribute__((address_space(4))) float* const_f32_p;
        float csc[8], csh[8];
        const float relu_lo = p.aff.relu ? 0.f : -__builtin_inff();
        if constexpr (AFF != 0) {
            const const_f32_p gsc = (const_f32_p)(unsigned long long)p.aff.sc, gsh = (const_f32_p)(unsigned long long)p.aff.sh;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const int ch = kb * 32 + hq * 8 + ci;
                const int cc = ch < p.C ? ch : p.C - 1;                    // (clamped: the value is zeroed below)
                csc[ci] = gsc[cc]; csh[ci] = gsh[cc];
            }
        }
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = lane + k * 64;
            if (u < PERB) {
                const int xp = u % W2;
                const int t = u / W2;
                const int lr = t % G::ROWS, im = t / G::ROWS;
                const int ir = row0 - 1 + lr, n = img0 + im;
                const bool live = ir >= 0 && ir < G::HS && n < p.N;
                f32x2 vv[8];
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    const bool keep = live && kb * 32 + hq * 8 + ci < p.C;
                    f32x2 v = rx[k][ci];
                    if constexpr (AFF == 1) {
                        v = f32x2{fmaxf(fmaf(v[0], csc[ci], csh[ci]), relu_lo), fmaxf(fmaf(v[1], csc[ci], csh[ci]), relu_lo)};
                    } else if constexpr (AFF == 2) {
                        const float a0 = fmaf(v[0], csc[ci], csh[ci]), a1 = fmaf(v[1], csc[ci], csh[ci]);
                        v = f32x2{fmaxf(a0, JVAE_LEAKY_SLOPE * a0), fmaxf(a1, JVAE_LEAKY_SLOPE * a1)};
                    }
                    vv[ci] = keep ? v : f32x2{0.f, 0.f};                   // padding rows / missing images and channels: exact zeros
                }
                u32x4 s[2][3];                                             // [pixel][plane]
#pragma unroll
                for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        unsigned hh, mm, ll;
                        x3_split2(f32x2{vv[2 * cp][j], vv[2 * cp + 1][j]}, hh, mm, ll);
                        s[j][0][cp] = hh; s[j][1][cp] = mm; s[j][2][cp] = ll;
                    }
                const int base = (im * 4 + hq) * G::CH + lr * G::WP + 1 + 2 * xp;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Xs[pl * G::XS + base + j] = s[j][pl];
            }
        }
    };

    // ---- weights: group g of K step kb = rows [(kb*25 + gstart(g)) * 12, +ntap*12) of 32-unit rows (OP units apart)
    constexpr int WU = 5;                                      // 1152 units of a 3-tap group over 256 threads (2-tap groups: 3)
    u32x4 rw[WU];
    const unsigned woff = (unsigned)((tid >> 5) * OP + o0 + (tid & 31));     // (uniform base + 32-bit lane offset, as for the patch)
    const unsigned wofft = tid < 128 ? woff : 0u;             // k = 4: units 1024 .. 1151 exist; the other threads read a valid stand-in
    auto gloadW = [&](int kb, auto g_c) {
        constexpr int g = decltype(g_c)::value;
        constexpr int nt = T::gstart(g + 1) - T::gstart(g);
        const u32x4* ub = p.wp + ((long)kb * 25 + T::gstart(g)) * 12 * OP;
#pragma unroll
        for (int k = 0; k < (nt == 3 ? 4 : 3); ++k) rw[k] = (ub + (long)k * 8 * OP)[woff];
        if constexpr (nt == 3) rw[4] = (ub + (long)32 * OP)[wofft];
    };
    auto lstoreW = [&](int buf, auto g_c) {
        constexpr int g = decltype(g_c)::value;
        constexpr int nt = T::gstart(g + 1) - T::gstart(g);
#pragma unroll
        for (int k = 0; k < (nt == 3 ? 4 : 3); ++k) Ws[buf * G::WGS + tid + k * 256] = rw[k];
        if constexpr (nt == 3) { if (tid < 128) Ws[buf * G::WGS + tid + 1024] = rw[4]; }
    };
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    typedef std::integral_constant<int, 0> G0;
    typedef std::integral_constant<int, 1> G1;

    // ---- prologue
    gloadX(0);
    gloadW(0, G0{});
    {   // halo columns (0 and WP - 1) of every patch row: cleared once, never written again; every other cell is rewritten by
        // lstoreX at every K step (out-of-image rows, missing images and channels as zeros)
        constexpr int NROW = 3 * G::NIMG * 4 * G::ROWS;
        for (int i = tid; i < NROW * 2; i += 256) {
            const int r = i >> 1;
            const int pl = r / (G::NIMG * 4 * G::ROWS), rr = r % (G::NIMG * 4 * G::ROWS);
            Xs[pl * G::XS + rr * G::WP + ((i & 1) ? G::WP - 1 : 0)] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    lstoreX(0);
    lstoreW(0, G0{});
    __builtin_amdgcn_sched_barrier(0);
    gloadW(0, G1{});
    lds_barrier();

    f32x4 acc[4][NPT][2];                                      // [output phase r*2 + q][pixel tile][16-channel tile]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ph][pt][ct][e] = 0.f;

    // ---- K steps: straight-line code over the 25 taps (9 weight groups) of the position-ordered sequence
    int gbase = 0;                                             // weight groups of the earlier K steps (buffer parity)
    for (int kb = 0; kb < KB; ++kb) {
        const bool nextk = kb + 1 < KB;
        // Fragment registers: the six products of a tap are ordered so that every plane dies as early as possible (weight planes
        // lo, mid, hi after products 0, 3, 5, patch planes hi, mid, lo after 2, 4, 5); see the read-ahead rules at the products.
        u32x4 fa[2][3][2], fb[3][NPT];                         // [tap parity][plane hi | mid | lo][channel tile] | [plane][pixel tile]
        auto fragA = [&](int buf, int tl, int pl, int par) {
            const u32x4* Wb = Ws + buf * G::WGS + kq * 32 + l15;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) fa[par][pl][ct] = Wb[((tl * 3 + pl) * 4) * 32 + ct * 16];
        };
        auto fragB = [&](int off, int pl) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) fb[pl][pt] = Xs[pl * G::XS + pixoff[pt] + off];
        };
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { fragA(gbase & 1, 0, pl, 0); fragB(T::dh(0) * G::WP + T::dw(0), pl); }
        auto tapstep = [&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            constexpr int g = T::group(t), tl = t - T::gstart(g);
            constexpr bool first = tl == 0, last = t + 1 == T::gstart(g + 1);
            constexpr bool newpos = t + 1 < 25 && T::pos(t + 1 < 25 ? t + 1 : t) != T::pos(t);
            constexpr int noff_h = T::dh(t + 1 < 25 ? t + 1 : t), noff_w = T::dw(t + 1 < 25 ? t + 1 : t);
            constexpr int par = t & 1;
            const int buf = (gbase + g) & 1;
            if constexpr (first) {
                // staging of the NEXT weight group behind this group's first fragment reads: buffer (gg + 1) & 1 was last read in
                // group gg - 1 (every wave is past its closing barrier); the store (which waits for the loads of rw) stays ahead of
                // the next loads
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g + 1 < T::NG) {
                    lstoreW(buf ^ 1, std::integral_constant<int, g + 1>{});
                } else {
                    if (nextk) lstoreW(buf ^ 1, G0{});
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g + 2 < T::NG) {
                    gloadW(kb, std::integral_constant<int, g + 2>{});
                } else {
                    if (nextk) gloadW(kb + 1, std::integral_constant<int, g + 2 - T::NG>{});
                }
                if constexpr (g == T::NG - 1) { if (nextk) gloadX(kb + 1); }
            }
            constexpr int ph = T::phase(t);
            // (weight plane, patch plane) of the six products; planes 0 = hi, 1 = mid, 2 = lo.  The patch planes die after products
            // 2 / 4 / 5 (hi, mid, lo) and a new POSITION's plane is read into the dead registers right there, 3-5 products ahead of
            // its first use; the next tap's weight planes go into the other register set behind products 0 (lo, mid) and 1 (hi) -
            // 4 or more products (16 MFMAs = 256+ matrix-pipe cycles) ahead: the lo plane's old registers are dead by then, the mid and
            // hi planes cost 16 registers.  (The order inside a tap is free: the accumulators hold the sums of all earlier taps.)
            constexpr int WPL[6] = {2, 1, 0, 1, 0, 0}, XPL[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[ph][pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[par][WPL[q]][ct]),
                                                                                  __builtin_bit_cast(bf16x8, fb[XPL[q]][pt]),
                                                                                  acc[ph][pt][ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!last) {                 // (same group: its weight buffer is valid)
                    if (q == 0) { fragA(buf, tl + 1, 2, par ^ 1); fragA(buf, tl + 1, 1, par ^ 1); }
                    if (q == 1) fragA(buf, tl + 1, 0, par ^ 1);
                }
                if constexpr (newpos) {                // (the patch does not change inside a K step: also across the closing barrier)
                    if (q == 2) fragB(noff_h * G::WP + noff_w, 0);
                    if (q == 4) fragB(noff_h * G::WP + noff_w, 1);
                    if (q == 5) fragB(noff_h * G::WP + noff_w, 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (last) {
                lds_barrier();
                if constexpr (t + 1 < 25) {                                // first tap of the next group (its buffer is complete now)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fragA(buf ^ 1, 0, pl, par ^ 1);
                }
            }
        };
        t2s_static_for(std::make_integer_sequence<int, 25>{}, tapstep);
        gbase += T::NG;
        if (nextk) {                                           // K step change: the patch is fully consumed
            lstoreX(kb + 1);
            lds_barrier();
        }
    }

    // ---- epilogue: lane holds pixel l15 of each 16-pixel tile, channels ct*16 + kq*4 + e, the four output phases.
    // The BatchNorm sums (pivot = bias) are taken from the bias-free accumulators first (registers only); then the bias goes INTO the
    // accumulators, four channels at a time, and the two column phases of a row are stored as 8-byte pairs.  Round 5: the first
    // version added the bias while forming each pair - `make_float2(acc0 + b, acc1 + b)` became `v_pk_add_f32 ... op_sel:[0,1]`
    // for the odd register of a bias pair (its HIGH register selected for the LOW result half) - and, rarely, 16 lanes of that low
    // half came out as if the bias were 0 (exactly -bias[c] on 16 outputs of channel 13 / 29: tools/t2_err_probe.py).  The same
    // operand-select form is what made round 4's 16-byte coefficient reads nondeterministic (profiles/NOTES.md, round 5:
    // the 2x2 experiment); the Makefile rejects any object that contains it (tools/isa_opsel_scan.py).  Adding a whole f32x4 of
    // bias to a whole accumulator quadruple needs no operand select at all.
    constexpr int HB = 2 * G::HS, WB = 2 * WS;
    float sv[16];                                              // [sum | sum of squares][channel tile][register]
    if (p.stats) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        // (pixels of images beyond N are exact zeros: zero patch, no bias in the sums)
                        const float v = acc[ph][pt][ct][e]; s1 += v; s2 += v * v;
                    }
                sv[ct * 4 + e] = s1;
                sv[8 + ct * 4 + e] = s2;
            }
    }
    {
        f32x4 bv[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(&bias_s[ct * 16 + kq * 4]);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ph][pt][ct] += bv[ct];
    }
    __builtin_amdgcn_sched_barrier(0);                         // every biased value exists before the first store is formed
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int pix = (wave * NPT + pt) * 16 + l15;
        const int im = pix / (G::TH * WS), rem = pix % (G::TH * WS);
        const int n = img0 + im;
        if (n >= p.N) continue;
        float* const dst = p.out + (((long)n * p.O + o0 + kq * 4) * HB + 2 * (row0 + rem / WS)) * WB + 2 * (rem % WS);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    *reinterpret_cast<float2*>(dst + ((long)(ct * 16 + e) * HB + r) * WB) =
                        make_float2(acc[r * 2][pt][ct][e], acc[r * 2 + 1][pt][ct][e]);
    }
    if (p.stats) {                                             // ... reduced after the stores have been issued (they drain meanwhile)
        {   // lane l15 of every 16-lane row receives the row total of sv[l15]
            const float tot = row_reduce16(sv);
            const int j = l15 & 7, ch = (j >> 2) * 16 + kq * 4 + (j & 3);
            red_s[(wave * 32 + ch) * 2 + (l15 >> 3)] = tot;
        }
        lds_barrier();                                         // LDS only: the output stores keep draining
        if (tid < 32) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red_s[(w * 32 + tid) * 2]; s2 += red_s[(w * 32 + tid) * 2 + 1]; }
            float* dst = p.stats + ((long)(o0 + tid) * gridDim.x + bx) * 2;      // slot of the TILE
            dst[0] = s1; dst[1] = s2;
        }
    }
}

template <int WS>
int launch_t2s(const T2X3P& p, hipStream_t st) {
    using G = T2SGeom<WS>;
    static_assert(G::LDS_BYTES + 2048 <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2s_x3_kernel<WS, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2s_x3_kernel<WS, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2s_x3_kernel<WS, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(G::HSWS >= G::PIX ? (unsigned)((long)p.N * G::HSWS / G::PIX) : (unsigned)((p.N + G::NIMG - 1) / G::NIMG),
              (unsigned)(p.O / 32));
    g_t2x3_splits = (int)grid.x;
    if (!p.aff.sc) hipLaunchKernelGGL((convt2s_x3_kernel<WS, 0>), grid, dim3(256), G::LDS_BYTES, st, p);
    else if (p.aff.relu == JVAE_ACT_LEAKY) hipLaunchKernelGGL((convt2s_x3_kernel<WS, 2>), grid, dim3(256), G::LDS_BYTES, st, p);
    else hipLaunchKernelGGL((convt2s_x3_kernel<WS, 1>), grid, dim3(256), G::LDS_BYTES, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void t2s_wpack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp,
                                                        int C, int O, long total, int swap, int flip) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        jvae_pack_t2s_elem(w, wp, i, C, O, swap, flip);
}

}  // namespace

// taken over from conv_t2_mfma.hip when the split-bf16 mode is on and there is at least one full K step of channels; the kernel
// addresses its input with 32-bit element offsets (larger tensors stay on the fp32 matrix-core kernel)
bool jvae_convt2_x3_ok(int N, int C, int WS, int O) {
    if (!jvae_conv5_x3_enabled()) return false;
    if ((long)N * C * WS * WS >= (1L << 31)) return false;
    return C >= 16 && C <= 256 && O % 32 == 0 && (WS == 8 || WS == 16 || WS == 32);
}

// w: the layer's weight read as [c][o][tap] (ConvTranspose2d layout / Conv2d dgrad); ws: jvae_conv5_x3_pack_bytes(C, O)
int jvae_convt2_x3(const float* in, const float* w, const float* bias, float* out, int N, int C, int WS, int O, float* ws,
                   hipStream_t st, float* stats, int* nsplit, const InAff* aff) {
    {
        bool fresh = true;
        float* slot = (float*)jvae_pack_cache_get(JVAE_PACK_T2S, w, C, O, 1, 0, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            const long total = jvae_pack_elems(JVAE_PACK_T2S, C, O);
            const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
            hipLaunchKernelGGL(t2s_wpack_kernel, dim3(blocks), dim3(256), 0, st, w, (__bf16*)ws, C, O, total, 1, 0);
            JVAE_LAUNCH_CHECK();
        }
    }
    T2X3P p{in, (const u32x4*)ws, bias, out, N, C, O, stats, aff ? *aff : InAff{nullptr, nullptr, 0}};
    struct Fin { int* n; ~Fin() { if (n) *n = g_t2x3_splits; } } fin{nsplit};
    switch (WS) {
        case 8: return launch_t2s<8>(p, st);
        case 16: return launch_t2s<16>(p, st);
        case 32: return launch_t2s<32>(p, st);
    }
    return JVAE_ENOTSUP;
}
