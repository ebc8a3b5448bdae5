// EXPERIMENT, NOT PART OF THE BUILD (round 5; result and reasons: profiles/r05_conv_s2_x3_ab.txt, profiles/NOTES.md).  To build it: copy this
// file to csrc/, apply conv_s2_x3_hooks.patch (pack kind JVAE_PACK_S2S, dispatch in jvae_conv5_fwd), make; tools/s2_probe.py measures it.
//
// Stride-2 5x5 convolution of the forward type (padding 2: 2H -> H) on the bf16 matrix cores with exact 3-way operand splitting -
// the polyphase form.  Round 5, built last; conv_t2_x3.hip's machinery (v_mfma_f32_16x16x32_bf16, K = 32 channels of ONE tap,
// 3 / 2-tap weight groups, straight-line tap sequence, wave-uniform staging with scalar BatchNorm coefficients) applied to
//
//   small[n][o][y][x] = bias[o] + sum_c sum_{kh, kw} big[n][c][2y + kh - 2][2x + kw - 2] * W[o][c][kh*5 + kw]
//
// With kh = 2a + p, kw = 2b + q the input index is 2 (y + a - 1) + p: the 25 taps fall into FOUR phase planes
// big_pq[r][s] = big[2r + p][2s + q] of the size of the OUTPUT, read at offsets (a - 1, b - 1), 9 + 6 + 6 + 4 taps.  A stride-2
// layer needs an input patch 4x its output tile - 128 KB for 128 pixels x 32 channels, which is what kept the two earlier split
// forms at one workgroup per CU or behind the fp32 kernel - but only ONE phase plane of it at a time: the K loop
// runs [32-channel block][phase (p, q)][taps of the phase], the LDS holds one (TH + 2) x (WS + 2) plane of 32 channels (34.5 KB, the
// geometry of the 4-phase kernel's patch), two workgroups share a CU.  The price is inherent: every staged element feeds 6.25 taps
// on average instead of 25, so the staging : MFMA ratio is four times that of a stride-1 layer and every tap reads its own patch
// fragments (12 ds_read_b128 per 24 MFMAs).
// Serves Conv2d(5, stride 2, padding 2) forward (features.3 / features.9 of conv32) and the dgrad of
// ConvTranspose2d(5, stride 2, padding 2, output_padding 1) (imager.6 / imager.12 of deconv32), fp32 NCHW in and out.
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

static thread_local int g_s2x3_splits = 0;

struct S2X3P {
    const float* in;     // big (N, C, 2HS, 2WS) fp32
    const u32x4* wp;     // split weights, JVAE_PACK_S2S (pack_elems.h): [K step of 32 channels][tap of the phase-ordered sequence][plane][kq][o]
    const float* bias;   // (O) or null
    float* out;          // small (N, O, HS, WS)
    int N, C, O;
    float* stats;        // optional (O, gridDim.x, 2)
    InAff aff;           // deferred BatchNorm(+ReLU) of the input
};

template <int WS>
struct S2SGeom {
    static constexpr int HS = WS;
    static constexpr int PIX = 128;
    static constexpr int HSWS = HS * WS;
    static constexpr int NIMG = PIX >= HSWS ? PIX / HSWS : 1;
    static constexpr int TH = PIX >= HSWS ? HS : PIX / WS;
    static constexpr int ROWS = TH + 2;
    static constexpr int WP = WS + 2;                          // units per row of a phase plane: columns s = -1 .. WS
    static constexpr int CH = ROWS * WP;                       // units per 8-channel block per image
    static constexpr int XS = NIMG * 4 * CH;                   // units of one split plane of the phase plane (32 channels)
    static constexpr int WGS = 3 * 3 * 4 * 32;                 // weight units of one group: 3 taps x 3 planes x 4 lane groups x 32 o
    static constexpr int LDS_BYTES = (3 * XS + 2 * WGS) * 16;
};

// tap t of the PHASE-ordered sequence (jvae_s2s_tap, pack_elems.h): phases (p, q) = (0,0) (0,1) (1,0) (1,1) with 9 / 6 / 6 / 4 taps; the
// weight groups of the 4-phase kernel (3,3,3,3,3,3,3,2,2 taps) end exactly on the phase boundaries 9 / 15 / 21 / 25
struct S2STap {
    static constexpr int NG = 9;
    __host__ __device__ static constexpr int gstart(int g) { return g < 7 ? 3 * g : (g == 7 ? 21 : (g == 8 ? 23 : 25)); }
    __host__ __device__ static constexpr int group(int t) { return t < 21 ? t / 3 : (t < 23 ? 7 : 8); }
    __host__ __device__ static constexpr int tap(int t) { return jvae_s2s_tap(t); }
    __host__ __device__ static constexpr int kh(int t) { return tap(t) / 5; }
    __host__ __device__ static constexpr int kw(int t) { return tap(t) % 5; }
    __host__ __device__ static constexpr int phase(int t) { return (kh(t) & 1) * 2 + (kw(t) & 1); }
    __host__ __device__ static constexpr int a(int t) { return kh(t) >> 1; }      // plane row of the tap relative to the output row: a - 1
    __host__ __device__ static constexpr int b(int t) { return kw(t) >> 1; }
    __host__ __device__ static constexpr int pstart(int ph) { return ph == 0 ? 0 : (ph == 1 ? 9 : (ph == 2 ? 15 : (ph == 3 ? 21 : 25))); }
};

template <int... I, class F>
__device__ __forceinline__ void s2s_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <int WS, int AFF>      // AFF: 0 = plain input, 1 = deferred BatchNorm (+ReLU by p.aff.relu), 2 = deferred BatchNorm + leaky ReLU
__global__ __launch_bounds__(256, 2) void convs2s_x3_kernel(S2X3P p) {
    using G = S2SGeom<WS>;
    using T = S2STap;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Xs = reinterpret_cast<u32x4*>(lds_raw);            // [3 planes][XS]: ONE phase plane of 32 channels
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
    const int OP = p.O;                                        // multiple of 32 (jvae_convs2_x3_ok)
    constexpr int HB = 2 * G::HS, WB = 2 * WS;
    if (tid < 32) bias_s[tid] = p.bias ? p.bias[o0 + tid] : 0.f;

    // two 16-pixel tiles per wave; the lane's output pixel (ly, lx) of each: plane cell of tap (a, b) = row ly + a, column lx + b
    // (plane row 0 is r = row0 - 1, plane column 0 is s = -1)
    constexpr int NPT = 2;
    int pixoff[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int pix = (wave * NPT + pt) * 16 + l15;
        const int im = pix / (G::TH * WS), rem = pix % (G::TH * WS);
        pixoff[pt] = (im * 4 + kq) * G::CH + (rem / WS) * G::WP + rem % WS;
    }

    // ---- staging: wave w owns channel block w of every K step; an item = 4 consecutive input columns x 8 channels of one input
    // row of parity p = the plane cells s = 2 xp, 2 xp + 1 of BOTH column phases (q = 0: elements 0, 2; q = 1: elements 1, 3)
    constexpr int W2 = WS / 2;
    constexpr int PERB = G::NIMG * G::ROWS * W2;               // items per channel block and row parity
    constexpr int XU = (PERB + 63) / 64;
    const int hq = __builtin_amdgcn_readfirstlane(wave);
    f32x2 rx[2][XU][8];                                        // [phase parity]: the two cells of ONE column phase per item (two 4-byte loads 8 bytes
                                                               // apart), loaded TWO phases ahead of their restage (a phase is 0.7-1.5 us of MFMAs)
    unsigned xoff[XU];                                         // element offset of the item's first cell at phase (0, 0) (32-bit: the host checks the size)
    const long cstride = (long)HB * WB;
#pragma unroll
    for (int k = 0; k < XU; ++k) {
        const int u = lane + k * 64;
        const int xp = u % W2;
        const int t = u / W2;
        const int lr = t % G::ROWS, im = t / G::ROWS;
        const int r = row0 - 1 + lr, n = img0 + im;
        const bool ok = u < PERB && r >= 0 && r < G::HS && n < p.N;
        xoff[k] = (unsigned)(((ok ? n : 0) * p.C * HB + (ok ? 2 * r : 0)) * WB + 4 * xp);
    }
    auto gloadX = [&](int kb, auto ph_c) __attribute__((always_inline)) {                     // phase ph = 2 p + q of channel block kb -> rx[ph & 1]
        constexpr int ph = decltype(ph_c)::value;
        // channels beyond C: the last valid channel is loaded instead and zeroed in lstoreX (the loads stay unconditional)
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = kb * 32 + hq * 8 + ci;
            const float* ub = p.in + (long)(c < p.C ? c : p.C - 1) * cstride + (ph >> 1) * WB + (ph & 1);
#pragma unroll
            for (int k = 0; k < XU; ++k) rx[ph & 1][k][ci] = f32x2{ub[xoff[k]], ub[xoff[k] + 2]};
        }
    };
    auto lstoreX = [&](int kb, auto ph_c) __attribute__((always_inline)) {                    // phase ph of channel block kb, held in rx[ph & 1]
        constexpr int ph = decltype(ph_c)::value;
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
                const int r = row0 - 1 + lr, n = img0 + im;
                const bool live = r >= 0 && r < G::HS && n < p.N;
                f32x2 vv[8];
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    const bool keep = live && kb * 32 + hq * 8 + ci < p.C;
                    f32x2 v = rx[ph & 1][k][ci];
                    if constexpr (AFF == 1) {
                        v = f32x2{fmaxf(fmaf(v[0], csc[ci], csh[ci]), relu_lo), fmaxf(fmaf(v[1], csc[ci], csh[ci]), relu_lo)};
                    } else if constexpr (AFF == 2) {
                        const float a0 = fmaf(v[0], csc[ci], csh[ci]), a1 = fmaf(v[1], csc[ci], csh[ci]);
                        v = f32x2{fmaxf(a0, JVAE_LEAKY_SLOPE * a0), fmaxf(a1, JVAE_LEAKY_SLOPE * a1)};
                    }
                    vv[ci] = keep ? v : f32x2{0.f, 0.f};                   // rows outside the image, missing images / channels: exact zeros
                }
                u32x4 s[2][3];                                             // [cell][plane]
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
    constexpr int WU = 5;
    u32x4 rw[WU];
    const unsigned woff = (unsigned)((tid >> 5) * OP + o0 + (tid & 31));
    const unsigned wofft = tid < 128 ? woff : 0u;
    auto gloadW = [&](int kb, auto g_c) __attribute__((always_inline)) {
        constexpr int g = decltype(g_c)::value;
        constexpr int nt = T::gstart(g + 1) - T::gstart(g);
        const u32x4* ub = p.wp + ((long)kb * 25 + T::gstart(g)) * 12 * OP;
#pragma unroll
        for (int k = 0; k < (nt == 3 ? 4 : 3); ++k) rw[k] = (ub + (long)k * 8 * OP)[woff];
        if constexpr (nt == 3) rw[4] = (ub + (long)32 * OP)[wofft];
    };
    auto lstoreW = [&](int buf, auto g_c) __attribute__((always_inline)) {
        constexpr int g = decltype(g_c)::value;
        constexpr int nt = T::gstart(g + 1) - T::gstart(g);
#pragma unroll
        for (int k = 0; k < (nt == 3 ? 4 : 3); ++k) Ws[buf * G::WGS + tid + k * 256] = rw[k];
        if constexpr (nt == 3) { if (tid < 128) Ws[buf * G::WGS + tid + 1024] = rw[4]; }
    };
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    typedef std::integral_constant<int, 0> G0;
    typedef std::integral_constant<int, 1> G1;
    typedef std::integral_constant<int, 2> G2;
    typedef std::integral_constant<int, 3> G3;

    // ---- prologue
    gloadX(0, G0{});
    gloadX(0, G1{});
    gloadW(0, G0{});
    {   // halo columns (s = -1 and s = WS) of every plane row: cleared once, never written again
        constexpr int NROW = 3 * G::NIMG * 4 * G::ROWS;
        for (int i = tid; i < NROW * 2; i += 256) {
            const int r = i >> 1;
            const int pl = r / (G::NIMG * 4 * G::ROWS), rr = r % (G::NIMG * 4 * G::ROWS);
            Xs[pl * G::XS + rr * G::WP + ((i & 1) ? G::WP - 1 : 0)] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    lstoreX(0, G0{});
    lstoreW(0, G0{});
    gloadX(0, G2{});                                           // rx[0] is free again
    __builtin_amdgcn_sched_barrier(0);
    gloadW(0, G1{});
    lds_barrier();

    f32x4 acc[NPT][2];                                         // [pixel tile][16-channel tile]
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pt][ct][e] = 0.f;

    // ---- K steps: [32-channel block][phase (p, q)]; straight-line code over the 25 taps (9 weight groups) of a channel block
    int gbase = 0;
    for (int kb = 0; kb < KB; ++kb) {
        const bool nextk = kb + 1 < KB;
        u32x4 fa[3][2], fb[3][NPT];                            // [plane hi | mid | lo][channel tile] | [plane][pixel tile]: single-buffered,
                                                               // each plane re-read for the next tap as soon as its last product is issued
        auto fragA = [&](int buf, int tl, int pl) __attribute__((always_inline)) {
            const u32x4* Wb = Ws + buf * G::WGS + kq * 32 + l15;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) fa[pl][ct] = Wb[((tl * 3 + pl) * 4) * 32 + ct * 16];
        };
        auto fragB = [&](int off, int pl) __attribute__((always_inline)) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) fb[pl][pt] = Xs[pl * G::XS + pixoff[pt] + off];
        };
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { fragA(gbase & 1, 0, pl); fragB(T::a(0) * G::WP + T::b(0), pl); }
        auto tapstep = [&](auto t_c) __attribute__((always_inline)) {
            constexpr int t = decltype(t_c)::value;
            constexpr int g = T::group(t), tl = t - T::gstart(g);
            constexpr bool first = tl == 0, last = t + 1 == T::gstart(g + 1);
            constexpr int ph = T::phase(t);
            constexpr bool phase_end = t + 1 == T::pstart(ph + 1);             // the plane changes behind this tap
            constexpr int tn = t + 1 < 25 ? t + 1 : t;
            constexpr int noff = T::a(tn) * G::WP + T::b(tn);
            const int buf = (gbase + g) & 1;
            if constexpr (first) {
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
                // the cells of the phase AFTER the next one are loaded under the first group behind every restage (its rx half is free
                // from there on)
                if constexpr (g == 0) { if (kb > 0) gloadX(kb, G2{}); }     // (phase 0 staged: rx[0] free; channel block 0: in the prologue)
                if constexpr (g == 3) gloadX(kb, G3{});                     // (phase 1 staged: rx[1] free)
                if constexpr (g == 5) { if (nextk) gloadX(kb + 1, G0{}); }  // (phase 2 staged)
                if constexpr (g == 7) { if (nextk) gloadX(kb + 1, G1{}); }  // (phase 3 staged)
            }
            constexpr int WPL[6] = {2, 1, 0, 1, 0, 0}, XPL[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[WPL[q]][ct]),
                                                                              __builtin_bit_cast(bf16x8, fb[XPL[q]][pt]),
                                                                              acc[pt][ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!last) {                 // (same group: its weight buffer is valid)
                    if (q == 0) fragA(buf, tl + 1, 2);
                    if (q == 3) fragA(buf, tl + 1, 1);
                    if (q == 5) fragA(buf, tl + 1, 0);
                }
                if constexpr (!phase_end) {            // every tap reads its own cells; the plane stays until the phase ends
                    if (q == 2) fragB(noff, 0);
                    if (q == 4) fragB(noff, 1);
                    if (q == 5) fragB(noff, 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (last) {
                lds_barrier();                                             // every wave is done with this group's weights (and plane)
                if constexpr (phase_end && t + 1 < 25) {                   // next phase plane of the same channel block
                    __builtin_amdgcn_sched_barrier(0);
                    lstoreX(kb, std::integral_constant<int, ph + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    lds_barrier();
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fragB(noff, pl);
                }
                if constexpr (t + 1 < 25) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fragA(buf ^ 1, 0, pl);
                }
            }
        };
        s2s_static_for(std::make_integer_sequence<int, 25>{}, tapstep);
        gbase += T::NG;
        if (nextk) {                                           // next channel block: plane (0, 0) of its rows
            lstoreX(kb + 1, G0{});
            lds_barrier();
        }
    }

    // ---- epilogue: lane holds pixel l15 of each 16-pixel tile, channels ct*16 + kq*4 + e.  BatchNorm sums (pivot = bias) from the
    // bias-free accumulators; the bias is added as whole quadruples (no operand select: profiles/NOTES.md round 5)
    float sv[16];                                              // [sum | sum of squares][channel tile][register]
    if (p.stats) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) { const float v = acc[pt][ct][e]; s1 += v; s2 += v * v; }
                sv[ct * 4 + e] = s1;
                sv[8 + ct * 4 + e] = s2;
            }
    }
    {
        f32x4 bv[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(&bias_s[ct * 16 + kq * 4]);
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[pt][ct] += bv[ct];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int pix = (wave * NPT + pt) * 16 + l15;
        const int im = pix / (G::TH * WS), rem = pix % (G::TH * WS);
        const int n = img0 + im;
        if (n >= p.N) continue;
        float* const dst = p.out + (((long)n * p.O + o0 + kq * 4) * G::HS + row0 + rem / WS) * WS + rem % WS;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(long)(ct * 16 + e) * G::HSWS] = acc[pt][ct][e];
    }
    if (p.stats) {
        {
            const float tot = row_reduce16(sv);
            const int j = l15 & 7, ch = (j >> 2) * 16 + kq * 4 + (j & 3);
            red_s[(wave * 32 + ch) * 2 + (l15 >> 3)] = tot;
        }
        lds_barrier();
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
int launch_s2s(const S2X3P& p, hipStream_t st) {
    using G = S2SGeom<WS>;
    static_assert(G::LDS_BYTES + 2048 <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convs2s_x3_kernel<WS, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convs2s_x3_kernel<WS, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convs2s_x3_kernel<WS, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(G::HSWS >= G::PIX ? (unsigned)((long)p.N * G::HSWS / G::PIX) : (unsigned)((p.N + G::NIMG - 1) / G::NIMG),
              (unsigned)(p.O / 32));
    g_s2x3_splits = (int)grid.x;
    if (!p.aff.sc) hipLaunchKernelGGL((convs2s_x3_kernel<WS, 0>), grid, dim3(256), G::LDS_BYTES, st, p);
    else if (p.aff.relu == JVAE_ACT_LEAKY) hipLaunchKernelGGL((convs2s_x3_kernel<WS, 2>), grid, dim3(256), G::LDS_BYTES, st, p);
    else hipLaunchKernelGGL((convs2s_x3_kernel<WS, 1>), grid, dim3(256), G::LDS_BYTES, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void s2s_wpack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp,
                                                        int C, int O, long total, int swap, int flip) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        jvae_pack_s2s_elem(w, wp, i, C, O, swap, flip);
}

}  // namespace

// taken over from conv_mfma.hip's stride-2 kernel when the split-bf16 mode is on; WS = OUTPUT width (input 2 WS); 32-bit element
// offsets into the input (larger tensors stay on the fp32 matrix-core kernel)
bool jvae_convs2_x3_ok(int N, int C, int WS, int O, int P) {
    if (!jvae_conv5_x3_enabled()) return false;
    { static const int off = [] { const char* e = getenv("JVAE_EXP_S2_OFF"); return (e && e[0] == '1') ? 1 : 0; }(); if (off) return false; }   // A/B of the experiment
    if (P != 2) return false;
    if ((long)N * C * 4 * WS * WS >= (1L << 31)) return false;
    return C >= 16 && C <= 256 && O % 32 == 0 && (WS == 8 || WS == 16 || WS == 32);
}

// w, swap, flip: as jvae_conv5_fwd (pack_elems.h: source [o][c][tap], swap = [c][o][tap], flip = tap -> 24 - tap); ws: jvae_conv5_x3_pack_bytes(C, O)
int jvae_convs2_x3(const float* in, const float* w, int swap, int flip, const float* bias, float* out, int N, int C, int WS, int O,
                   float* ws, hipStream_t st, float* stats, int* nsplit, const InAff* aff) {
    {
        bool fresh = true;
        float* slot = (float*)jvae_pack_cache_get(JVAE_PACK_S2S, w, C, O, swap, flip, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            const long total = jvae_pack_elems(JVAE_PACK_S2S, C, O);
            const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
            hipLaunchKernelGGL(s2s_wpack_kernel, dim3(blocks), dim3(256), 0, st, w, (__bf16*)ws, C, O, total, swap, flip);
            JVAE_LAUNCH_CHECK();
        }
    }
    S2X3P p{in, (const u32x4*)ws, bias, out, N, C, O, stats, aff ? *aff : InAff{nullptr, nullptr, 0}};
    struct Fin { int* n; ~Fin() { if (n) *n = g_s2x3_splits; } } fin{nsplit};
    switch (WS) {
        case 8: return launch_s2s<8>(p, st);
        case 16: return launch_s2s<16>(p, st);
        case 32: return launch_s2s<32>(p, st);
    }
    return JVAE_ENOTSUP;
}
