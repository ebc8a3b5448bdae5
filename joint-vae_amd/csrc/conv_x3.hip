// fp32 5x5 convolution on the bf16 matrix cores by 3-way operand splitting ("x3").
//
//   out[n][o][y][x] = bias[o] + sum_{c,kh,kw} in[n][c][y*S + kh - P][x*S + kw - P] * W[c][kh*5+kw][o]
//
// Same operator, tensors (fp32 NCHW in, fp32 NCHW out) and weight roles (swap / flip) as conv_mfma.hip; only the
// arithmetic unit differs.  v_mfma_f32_32x32x2_f32 retires 64 FLOP/cycle/SIMD (157 TFLOP/s per MI355X),
// v_mfma_f32_32x32x16_bf16 1024 (2.5 PFLOP/s).  Every fp32 operand is written EXACTLY as a sum of three bf16 numbers
//      v = hi + mid + lo,   hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)      (8 + 8 + 8 significand bits)
// and a product x*w is accumulated (fp32 accumulators inside the MFMA) from the six partial products whose weight is
// >= 2^-24 of the full one: x_lo w_hi, x_hi w_lo, x_mid w_mid, x_mid w_hi, x_hi w_mid, x_hi w_hi.  Products of bf16
// numbers are exact in fp32, the three dropped terms are <= 2^-24 |x w|: the result differs from an fp32 FMA chain
// by less than that chain's own rounding (measured against an fp64 convolution on wide-dynamic-range data: 3-15e-7 of the
// output scale for the fp32-MFMA kernel, 2-13e-7 for this one: both are the fp32 ACCUMULATION; tests/test_ops_gpu.py).  Six bf16 MFMAs cover K = 16 in 6*32 = 192 cycles, the fp32 MFMA needs
// 8*64 = 512: the fp32-equivalent ceiling is 2.5 PFLOP/s / 6 = 417 TFLOP/s.
//
// Mapping (one workgroup = 4 waves = MT*128 output pixels x 32 output channels, K step = 16 input channels):
//  * the input patch (with zero halo) is loaded as fp32 NCHW rows (8-byte loads, 2 pixels x 8 channels per thread),
//    the deferred BatchNorm(+ReLU) of conv_mfma.hip is applied in registers, the value is split and stored as three
//    planes of 16-byte units (8 channels of one pixel = one lane's MFMA operand): every fragment read is one
//    ds_read_b128 at lane offset + immediate;
//  * weights are split once per call by the re-pack kernel ([K step][kernel row][plane][kw][half][o] units) and
//    staged per KERNEL ROW (5 taps, 15 KB) into a double-buffered LDS area: one barrier per 60 MFMAs, two workgroups
//    per CU (77 KB each);
//  * accumulators, bias, BatchNorm partial sums and the coalesced NCHW epilogue are those of the fp32 kernel.
// Default since round 4 (maps up to 32 wide): the 16x16x32 form (template argument SH) - K = 2 consecutive taps x 16 channels per
// MFMA, weights staged per group of two tap pairs, a wave-uniform channel block per staging item (the deferred BatchNorm's
// coefficients are scalar operands), straight-line weight groups with their read-ahead, two tiles per workgroup on the large
// launches: see the comments at computeSH, lstoreX and the tile loop, and DESIGN.md section 4.
#include <stdlib.h>
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"
#include "conv_x3.h"
#include <type_traits>
#include "pack_elems.h"

namespace {

typedef x3_bf16x8 bf16x8;
typedef x3_u32x4 u32x4;
typedef x3_f32x2 f32x2;

// Wp[(kb*5 + kh)][(plane*5 + kw)*2 + half][o][ci] = plane(W[o][c = kb*16 + half*8 + ci][tap = kh*5 + kw])  (o < OP)
// swap: source is [c][o][tap] (ConvTranspose2d layout / role swap), flip: tap -> 24 - tap
__global__ __launch_bounds__(256) void x3_wpack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp,
                                                       int C, int O, int KB, int OP, int swap, int flip) {
    const long total = (long)KB * 25 * 2 * OP * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        jvae_pack_x3_elem(w, wp, i, C, O, swap, flip);
}

__global__ __launch_bounds__(256) void x3s_wpack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp,
                                                        int C, int O, long total, int swap, int flip) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        jvae_pack_x3s_elem(w, wp, i, C, O, swap, flip);
}

struct X3P {
    const float* in;     // (N, Cin, H, W) fp32
    const u32x4* wp;     // packed split weights (KB*5, 30, OP) units
    const float* bias;   // (CoutReal) or null
    float* out;          // (N, CoutReal, OH, OW) fp32
    int N, Cin, H, W, OP, P, CoutReal;
    float* stats;        // optional (CoutReal, gridDim.x, 2)
    InAff aff;           // deferred BatchNorm(+ReLU) of the input (sc == nullptr: none)
    int tpw;             // tiles per workgroup (16x16x32 form: 1, or 2 = tiles w and w + gridDim.x; gridDim.x * tpw = number of tiles)
#ifdef JVAE_X3_STAMPS
    unsigned long long* dbg;   // DIAGNOSTIC BUILD ONLY (tools/x3_stamps.py): 96 s_memtime stamps of wave 0 per workgroup
#endif
};

// In-kernel anatomy (cdna_hip_programming.md section 7, In-kernel stamps): a separate diagnostic build (-DJVAE_X3_STAMPS, make stamps)
// records s_memtime at the phase boundaries of wave 0 of every workgroup into a buffer of its own; the product build contains
// no stamp.
#ifdef JVAE_X3_STAMPS
#define X3_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (p.dbg && threadIdx.x == 0)                                                                           \
            p.dbg[((long)blockIdx.y * gridDim.x + blockIdx.x) * 96 + (i)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
#define X3_STAMP_RT(i)                                                                                           \
    do {                                                                                                         \
        if (p.dbg && threadIdx.x == 0)                                                                           \
            p.dbg[((long)blockIdx.y * gridDim.x + blockIdx.x) * 96 + (i)] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
#else
#define X3_STAMP(i) do {} while (0)
#define X3_STAMP_RT(i) do {} while (0)
#endif

// SH = false: v_mfma_f32_32x32x16_bf16, K index = 16 channels of one tap, weights staged per kernel row (5 taps).
// SH = true:  v_mfma_f32_16x16x32_bf16, K index = 2 CONSECUTIVE taps (of the 25-tap sequence) x 16 channels - lane group
//             kq = lane >> 4 supplies tap 2*pair + (kq >> 1), channel block kq & 1 - weights staged per GROUP of 2 tap pairs
//             (7 groups per K step: 12 pairs + tap 24 with an all-zero partner; pack_elems.h JVAE_PACK_X3S).  Same output tile
//             per wave, same patch image, same LDS reads per MFMA cycle; 4 % more MFMA cycles (the empty half pair).  The chip
//             holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back item 7).
//             (A stride-2 variant of this form on a de-interleaved 64-pixel patch was built in round 4, never beat the fp32 kernel and
//             left the tree in round 5: profiles/NOTES.md.)
template <int S, int OW, int MT, bool SH = false>
struct X3Geom {
    static constexpr int OH = OW;
    static constexpr int PIX = MT * 128;
    static constexpr int OHW = OH * OW;
    static constexpr int NIMG = PIX >= OHW ? PIX / OHW : 1;
    static constexpr int TH = PIX >= OHW ? OH : PIX / OW;
    static constexpr int ROWS = (TH - 1) * S + 5;
    static constexpr int WIN = OW * S;
    static constexpr int WP0 = (OW - 1) * S + 9;
    static constexpr int WP1 = WIN + 4;
    static constexpr int WP = WP0 > WP1 ? WP0 : WP1;           // units per patch row
    static constexpr int CH = ROWS * WP;                       // units per 8-channel block per image
    static constexpr int XS = NIMG * 2 * CH;                   // patch units of one plane (16 channels)
    static constexpr int WGS = SH ? 2 * 3 * 4 * 32             // weight units of one group: 2 pairs x 3 planes x 4 lane groups
                                  : 3 * 5 * 2 * 32;            // ... of one kernel row (3 planes x 5 taps x 2 halves)
    static constexpr int GPK = SH ? 7 : 5;                     // weight groups per K step
    static constexpr int WROWS = SH ? JVAE_X3S_PAIRS * 12 : 150;   // 32-unit rows of packed weights per K step and 32 channels
    static constexpr int NPT = SH ? PIX / 64 : MT;             // pixel tiles per wave (16 pixels each in the SH form, else 32)
    static constexpr int LDS_BYTES = (3 * XS + 2 * WGS) * 16;
};

template <bool SH, int NPT> struct X3Acc { f32x16 t[NPT]; };
template <int NPT> struct X3Acc<true, NPT> { f32x4 t[NPT][2]; };        // [16-pixel tile][16-channel tile]

template <int S, int OW, int MT, int AFF, bool SH = false>      // AFF: 0 plain input, 1 deferred BatchNorm (+ReLU by flag), 2 ... + leaky ReLU
__global__ __launch_bounds__(256, 2) void conv5_x3_kernel(X3P p) {
    using G = X3Geom<S, OW, MT, SH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Xs = reinterpret_cast<u32x4*>(lds_raw);            // [3 planes][XS]
    u32x4* Ws = Xs + 3 * G::XS;                                // [2 buffers][WGS]
    __shared__ float bias_s[32];                              // this workgroup's 32 bias values: fetched while the first patch
                                                              // loads are in flight (a global load in the epilogue is an exposed
                                                              // round trip per workgroup: 36 us of the 32-wide layer)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    X3_STAMP(0);
    X3_STAMP_RT(64);
    constexpr int TILES_PER_IMG = G::OHW >= G::PIX ? G::OHW / G::PIX : 1;
    // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so the row tiles
    // of ONE image - which share their halo rows - used to sit on different XCDs and every halo row came from HBM again
    // (profiles/r02_x3_fwd_pmc.json: 188 MB read for 134 MB of input).  Workgroup w = (xcd, k) takes tile xcd * (tiles / 8) + k:
    // neighbouring tiles run on the same XCD at about the same time, the halo is an L2 hit.
    // Two tiles per workgroup (16x16x32 form, p.tpw = 2; round 4), so that the second tile's first patch and
    // first weight groups are loaded under the last weight groups of the first one and its prologue (entry, address set-up, zero
    // fill, the round trip of the first loads: 3 500 of a tile's 63 000 cycles) is paid once.
    // Its tiles are w and w + gridDim.x, NOT 2w and 2w + 1: the workgroups that run side by side then hold neighbouring tiles in both
    // passes and share their halo rows in L2 (with consecutive tiles per workgroup the PMC passes read 172 MB instead of 136).
    const int TPW = SH ? p.tpw : 1;
    int bx = xcd_tile(blockIdx.x, gridDim.x);
    const int ntiles = gridDim.x * TPW;
    int img0 = (G::OHW >= G::PIX) ? bx / TILES_PER_IMG : bx * G::NIMG;
    int row0 = (G::OHW >= G::PIX) ? (bx % TILES_PER_IMG) * G::TH : 0;
    const int o0 = blockIdx.y * 32;
    const int KB = (p.Cin + 15) / 16;
    const int NG = KB * G::GPK;                                // weight groups: (K step, kernel row | group of 2 tap pairs)

    if (tid < 32) bias_s[tid] = (p.bias && o0 + tid < p.CoutReal) ? p.bias[o0 + tid] : 0.f;
    // pixel tiles of this wave: MT groups of 32 pixels (32x32x16: the pixel on lane & 31, channel block `half`), or 2*MT tiles of 16
    // (16x16x32: the pixel on lane & 15, lane group kq = lane >> 4 = (tap of the pair, channel block))
    constexpr int NPT = G::NPT, TPX = SH ? 16 : 32;
    const int l15 = lane & 15, kq = lane >> 4;
    int pixoff[NPT];
#pragma unroll
    for (int mt = 0; mt < NPT; ++mt) {
        const int pix = (wave * NPT + mt) * TPX + (SH ? l15 : l31);
        const int im = pix / (G::TH * OW), rem = pix % (G::TH * OW);
        const int r = rem / OW, c = rem % OW;
        pixoff[mt] = im * (2 * G::CH) + (SH ? (kq & 1) : half) * G::CH + (r * S) * G::WP + c * S + 4 - p.P;
    }

    // two accumulator sets: the three small partial products are summed apart from the three large ones (added in
    // the epilogue), which also doubles the distance between dependent MFMAs
    X3Acc<SH, NPT> accA, accS;
    auto& acc = accA.t;
    auto& acs = accS.t;
    if constexpr (SH) {
#pragma unroll
        for (int b = 0; b < NPT; ++b)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[b][ct][r] = 0.f; acs[b][ct][r] = 0.f; }
    } else {
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[b][r] = 0.f; acs[b][r] = 0.f; }
    }

    int in_row0 = row0 * S - p.P;
    constexpr int W2 = G::WIN / 2;
    // Staging items = (2 pixels x 8 channels).  The 8-channel block `hq` of an item is WAVE-UNIFORM (round 4): waves 0, 1 stage
    // block 0, waves 2, 3 block 1, each pair walking the PERH (image, row, pixel pair) items of its block in XU passes of 128 - so
    // the channel of rx[k][ci] is the same for every lane of a wave and the deferred BatchNorm's coefficients are SCALAR operands
    // (s_load from the coefficient vectors, no LDS table).  The first mapping (block = a per-lane function of the item index) read
    // them from LDS inside the per-lane `live` branch: 16 dependent ds_read_b32 per item, 3 600 + 2 400 of a workgroup's 74 000
    // cycles (tools/x3_stamps.py with AFF=0 / 1).
    constexpr int PERH = G::NIMG * G::ROWS * W2;               // items per 8-channel block and K step
    constexpr int XU = (PERH + 127) / 128, WU = (G::WGS + 255) / 256;
    const int hq = __builtin_amdgcn_readfirstlane(tid >> 7);   // this wave's channel block (scalar)
    f32x2 rx[XU][8];
    u32x4 rw[WU];

    // Loads are unconditional (out-of-range items read a valid stand-in address and are zeroed when they are stored to
    // LDS): with branches around them the compiler cannot count outstanding loads and falls back to vmcnt(0).  Their
    // addresses are computed once: per K step / weight group only a uniform stride is added.
    const float* xsrc[XU];
    const float* xsrcN[XU];                                    // ... of the workgroup's next tile
    const u32x4* wsrc[WU];
    const long cstride = (long)p.H * p.W;
    auto tile_src = [&](int timg0, int tin_row0, const float* (&dst)[XU]) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = (tid & 127) + k * 128;
            const int xp = u % W2;
            const int t = u / W2;
            const int lr = t % G::ROWS, im = t / G::ROWS;
            const int ir = tin_row0 + lr, n = timg0 + im;
            const bool ok = u < PERH && ir >= 0 && ir < p.H && n < p.N;
            dst[k] = p.in + (((long)(ok ? n : 0) * p.Cin + hq * 8) * p.H + (ok ? ir : 0)) * p.W + 2 * xp;
        }
    };
    tile_src(img0, in_row0, xsrc);
#pragma unroll
    for (int k = 0; k < XU; ++k) xsrcN[k] = xsrc[k];
#pragma unroll
    for (int k = 0; k < WU; ++k) {
        const int u = min(tid + k * 256, G::WGS - 1);          // the tail threads re-read the last unit (not stored)
        wsrc[k] = p.wp + (long)(u / 32) * p.OP + o0 + u % 32;
    }
    auto gloadX = [&](int kb, bool next_tile = false) {
        // channels beyond Cin: a clamped (valid) channel is loaded instead and zeroed in lstoreX - pure address arithmetic on
        // wave-uniform values, so that the loads stay unconditional (a per-lane select between two ADDRESSES was compiled into
        // branches around the loads with a vmcnt(0) behind each: 16 serialised round trips in the prologue)
        const int cmax = p.Cin - 1 - (kb * 16 + hq * 8);       // last valid ci of this block (< 0: the whole block is beyond Cin)
        const int cm = cmax < 0 ? 0 : (cmax > 7 ? 7 : cmax);
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const float* base = next_tile ? xsrcN[k] : xsrc[k];       // (uniform select)
            const float* src = cmax < 0 ? base : base + (long)kb * 16 * cstride;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
                rx[k][ci] = *reinterpret_cast<const f32x2*>(src + (long)(ci < cm ? ci : cm) * cstride);
        }
    };
    auto gloadW = [&](int g) {
#pragma unroll
        for (int k = 0; k < WU; ++k) rw[k] = wsrc[k][(long)g * (G::WGS / 32) * p.OP];
    };
    auto lstoreX = [&](int kb) {                    // kb: the K step whose data sits in rx
        // deferred BatchNorm: (scale, shift) of this wave's 8 channels as scalars (uniform addresses in the constant address space)
        typedef const __attribute__((address_space(4))) float* const_f32_p;
        float csc[8], csh[8];
        const float relu_lo = p.aff.relu ? 0.f : -__builtin_inff();
        if constexpr (AFF != 0) {
            const const_f32_p gsc = (const_f32_p)(unsigned long long)p.aff.sc, gsh = (const_f32_p)(unsigned long long)p.aff.sh;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const int ch = kb * 16 + hq * 8 + ci;
                const int cc = ch < p.Cin ? ch : p.Cin - 1;                // (clamped: the value is zeroed below)
                csc[ci] = gsc[cc]; csh[ci] = gsh[cc];
            }
        }
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = (tid & 127) + k * 128;
            if (u < PERH) {
                const int xp = u % W2;
                const int t = u / W2;
                const int lr = t % G::ROWS, im = t / G::ROWS;
                const int h = hq;
                // out-of-image rows / missing images / channels were loaded from a stand-in address: exact zeros
                const int ir = in_row0 + lr, n = img0 + im;
                const bool live = ir >= 0 && ir < p.H && n < p.N;
                f32x2 vv[8];                                               // [channel](pixel 0, pixel 1)
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    const bool keep = live && kb * 16 + h * 8 + ci < p.Cin;
                    f32x2 v = rx[k][ci];                                   // (a stand-in value where !keep: replaced below)
                    if constexpr (AFF == 1) {                              // ReLU as max(., lo) with lo = 0 or -inf: no select
                        v = f32x2{fmaxf(fmaf(v[0], csc[ci], csh[ci]), relu_lo), fmaxf(fmaf(v[1], csc[ci], csh[ci]), relu_lo)};
                    } else if constexpr (AFF == 2) {                       // leaky ReLU: max(a, 0.01 a)
                        const float a0 = fmaf(v[0], csc[ci], csh[ci]), a1 = fmaf(v[1], csc[ci], csh[ci]);
                        v = f32x2{fmaxf(a0, JVAE_LEAKY_SLOPE * a0), fmaxf(a1, JVAE_LEAKY_SLOPE * a1)};
                    }
                    vv[ci] = keep ? v : f32x2{0.f, 0.f};                   // padding rows / missing channels stay exact zeros
                }
                u32x4 s[2][3];                                             // [pixel][plane]: 8 channels = 4 packed pairs
#pragma unroll
                for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                    {
                        unsigned hh, mm, ll;
                        x3_split2(f32x2{vv[2 * cp][j], vv[2 * cp + 1][j]}, hh, mm, ll);
                        s[j][0][cp] = hh; s[j][1][cp] = mm; s[j][2][cp] = ll;
                    }
                const int base = (im * 2 + h) * G::CH + lr * G::WP + 4 + 2 * xp;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        Xs[pl * G::XS + base + j] = s[j][pl];
            }
        }
    };
    auto lstoreW = [&](int buf) {
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * 256;
            if (u < G::WGS) Ws[buf * G::WGS + u] = rw[k];
        }
    };

    // one kernel row (5 taps) of one K step: 6 MFMAs per tap and pixel group, fragments of tap kw+1 read ahead
    auto compute = [&](int buf, int rowoff) {
      if constexpr (!SH) {
        const u32x4* Wb = Ws + buf * G::WGS + half * 32 + l31;
        u32x4 fa[2][3], fb[2][3][MT];
        auto frag = [&](int kw, u32x4 (&a)[3], u32x4 (&b)[3][MT]) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[pl] = Wb[(pl * 5 + kw) * 64];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) b[pl][mt] = Xs[pl * G::XS + pixoff[mt] + rowoff + kw];
            }
        };
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
            if (kw + 1 < 5) frag(kw + 1, fa[(kw + 1) & 1], fb[(kw + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            // (weight plane, input plane): small terms (even t) -> acs, large terms (odd t) -> acc
            constexpr int WPL[6] = {0, 0, 2, 1, 1, 0}, XPL[6] = {2, 1, 0, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x16& d = (t & 1) ? acc[mt] : acs[mt];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[kw & 1][WPL[t]]),
                                                                __builtin_bit_cast(bf16x8, fb[kw & 1][XPL[t]][mt]), d, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    // 16x16x32 form.  One group = 2 tap pairs (the last group of a K step: 1); per pair the wave issues 6 products x NPT pixel
    // tiles x 2 channel tiles; the pair is worked through in halves of 2 pixel tiles (24 MFMAs), the fragments of the next half
    // are read ahead.  Lane group kq reads tap 2*pair + (kq >> 1): the second tap of a pair lies one unit to the right, or - when
    // the pair crosses a kernel row (taps 4|5, 14|15) - one row down and four units to the left; tap 25 does not exist (its
    // weights are zero): those lanes re-read tap 24 so that no value from outside the receptive field enters a 0 * x.
    // npair (1 for the last group of a K step) is a COMPILE-TIME constant (round 4): as a run-time value it put the read-ahead
    // of the next half behind a branch, and at the join the compiler's s_waitcnt had to assume the path WITHOUT the new reads -
    // lgkmcnt(2) / (0) right behind twelve fresh ds_read_b128, i.e. the read-ahead was waited for before the MFMAs it was meant
    // to hide under (one exposed LDS round trip per pair).
    // pre(): the staging of the next weight group, issued BEHIND the group's first fragment reads (their LDS round trip runs under
    // its ~500 cycles of address arithmetic, waits for the weight loads and LDS stores) and in front of the first MFMA.
    auto computeSH = [&](int buf, int gi, auto npair_c, auto&& pre) {
        constexpr int TPH = NPT >= 2 ? 2 : 1;                    // pixel tiles per half
        constexpr int NH = SH ? NPT / TPH : 1;                   // halves per pair
        const u32x4* Wb = Ws + buf * G::WGS + kq * 32 + l15;
        constexpr int npair = decltype(npair_c)::value;
        u32x4 fa[2][3][2], fb[2][3][TPH];
        auto tapoff = [&](int t) {                               // LDS unit offset of tap t relative to the lane's pixel
            const int kh = t / 5, kw = t - 5 * kh;
            return kh * G::WP + kw;
        };
        auto offs = [&](int pq) {                                // ... of THIS lane's tap of pair pq (tap 25: tap 24 again)
            const int ta = 4 * gi + 2 * pq;
            const int oa = tapoff(ta), ob = ta == 24 ? oa : tapoff(ta + 1);
            return (kq >> 1) ? ob : oa;
        };
        auto fragA = [&](int pq, u32x4 (&a)[3][2]) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) a[pl][ct] = Wb[((pq * 3 + pl) * 4) * 32 + ct * 16];
        };
        auto fragB = [&](int off, int hp, u32x4 (&b)[3][TPH]) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < TPH; ++j) b[pl][j] = Xs[pl * G::XS + pixoff[(hp * TPH + j) % NPT] + off];
        };
        int off = offs(0);
        fragA(0, fa[0]);
        fragB(off, 0, fb[0]);
#pragma unroll
        for (int pq = 0; pq < 2; ++pq) {
            if (pq < npair) {
                const int offn = pq + 1 < npair ? offs(pq + 1) : off;
#pragma unroll
                for (int hp = 0; hp < NH; ++hp) {
                    const int cur = (pq * NH + hp) & 1;
                    if (hp + 1 < NH) fragB(off, hp + 1, fb[cur ^ 1]);
                    else if (pq + 1 < npair) { fragA(pq + 1, fa[(pq + 1) & 1]); fragB(offn, 0, fb[cur ^ 1]); }
                    if (pq == 0 && hp == 0) { __builtin_amdgcn_sched_barrier(0); pre(); }
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr int WPL[6] = {0, 0, 2, 1, 1, 0}, XPL[6] = {2, 1, 0, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int j = 0; j < TPH; ++j)
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) {
                                if constexpr (SH) {
                                    f32x4& d = (t & 1) ? acc[hp * TPH + j][ct] : acs[hp * TPH + j][ct];
                                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[pq & 1][WPL[t]][ct]),
                                                                                __builtin_bit_cast(bf16x8, fb[cur][XPL[t]][j]), d, 0, 0, 0);
                                }
                            }
                    __builtin_amdgcn_sched_barrier(0);
                }
                off = offn;
            }
        }
    };

    // LDS-only barrier: __syncthreads() also waits for vmcnt(0), i.e. for the global prefetches in flight
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- epilogue of one tile, 16x16x32 form (called per tile; resets the accumulators)
    __shared__ float red_s[SH ? 4 * 32 * 2 : 1];
    auto epilogueSH = [&] {
      if constexpr (SH) {
        // ---- epilogue, 16x16x32: lane holds pixel l15 of each 16-pixel tile, channels ct*16 + kq*4 + r
#pragma unroll
        for (int b = 0; b < NPT; ++b)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[b][ct] += acs[b][ct];
        float bv[2][4];                            // (bias_s holds zeros without a bias)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[ct][r] = bias_s[ct * 16 + kq * 4 + r];
        const bool fullc = o0 + 32 <= p.CoutReal;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int pix = (wave * NPT + pt) * 16 + l15;
            const int im = pix / (G::TH * OW), rem = pix % (G::TH * OW);
            const int n = img0 + im;
            if (n >= p.N) continue;
            const int oy = row0 + rem / OW, ox = rem % OW;
            // one base address per pixel tile, the lane's 8 channels at constant strides from it; a full 32-channel block (the
            // uniform, usual case) needs no per-channel bound check - it was a compare + exec branch + 64-bit address chain in
            // front of each of the 32 stores
            float* const dst = p.out + (((long)n * p.CoutReal + o0 + kq * 4) * G::OH + oy) * OW + ox;
            if (fullc) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(ct * 16 + r) * G::OHW] = acc[pt][ct][r] + bv[ct][r];
            } else {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (o0 + ct * 16 + kq * 4 + r < p.CoutReal) dst[(ct * 16 + r) * G::OHW] = acc[pt][ct][r] + bv[ct][r];
            }
        }
        X3_STAMP(6);                               // output stores issued
        if (p.stats) {
            float* red = red_s;                                   // [4 waves][32][2] (not the patch area: its halo cells stay zero
                                                                  // for the workgroup's next tile)
            float sv[16];                                         // [sum | sum of squares][channel tile][register]
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) { const float v = acc[pt][ct][r]; s1 += v; s2 += v * v; }
                    sv[ct * 4 + r] = s1;
                    sv[8 + ct * 4 + r] = s2;
                }
            {   // lane l15 of every 16-lane row receives the row total of sv[l15]
                const float tot = row_reduce16(sv);
                const int j = l15 & 7, ch = (j >> 2) * 16 + kq * 4 + (j & 3);
                red[(wave * 32 + ch) * 2 + (l15 >> 3)] = tot;
            }
            lds_barrier();                                        // LDS only: the output stores keep draining
            if (tid < 32 && o0 + tid < p.CoutReal) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s1 += red[(w * 32 + tid) * 2]; s2 += red[(w * 32 + tid) * 2 + 1]; }
                float* dst = p.stats + ((long)(o0 + tid) * ntiles + bx) * 2;
                dst[0] = s1; dst[1] = s2;
            }
        }
#pragma unroll
        for (int b = 0; b < NPT; ++b)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[b][ct][r] = 0.f; acs[b][ct][r] = 0.f; }
      }
    };

    X3_STAMP(1);                                   // 0 = kernel entry (below the declarations), 1 = zero fill issued
    gloadX(0);
    gloadW(0);
    // The halo COLUMNS (4 units left of the image row, WP - 4 - WIN right of it) are zeroed once and never written again;
    // every other cell - out-of-image rows, missing images and channels included - is rewritten by lstoreX at every K step.
    // Issued behind the first global loads (they fly meanwhile).  Round 4: the whole 46 KB image used to be cleared, in front
    // of the loads: 2 200 of a workgroup's 73 500 cycles (tools/x3_stamps.py).
    {
        constexpr int HALO = G::WP - G::WIN, NROW = 3 * G::NIMG * 2 * G::ROWS;
        for (int i = tid; i < NROW * HALO; i += 256) {
            const int r = i / HALO, c = i % HALO;
            const int pl = r / (G::NIMG * 2 * G::ROWS), rr = r % (G::NIMG * 2 * G::ROWS);
            Xs[pl * G::XS + rr * G::WP + (c < 4 ? c : c + G::WIN)] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();                               // halo zero fill complete (it overlaps nothing lstoreX writes, but orders bias_s)
    X3_STAMP(2);
    lstoreX(0);
    lstoreW(0);
    __builtin_amdgcn_sched_barrier(0);
    if (NG > 1) gloadW(1);
    lds_barrier();
    X3_STAMP(3);
    if constexpr (SH) {
        // 16x16x32 form: K steps outside, the GPK - 1 two-pair groups of a K step inside, its one-pair group with the K-step change
        // behind them - so that the registers of the next patch (rx: loaded at the start of the one-pair group, split and stored
        // behind it) are live in that tail only and not across the two-pair code, which needs the room for its read-ahead
        typedef std::integral_constant<int, 1> one_pair;
        typedef std::integral_constant<int, 2> two_pairs;
        int g = 0, gbase = 0;                          // group inside the tile; groups of the workgroup's earlier tiles
        bool has_next = false;
        auto stageW = [&] {
            // buffer (gg+1)&1 was last read in group gg-1: every wave is past it.  The store (which waits for the loads of
            // rw) must stay ahead of the next loads: the scheduler would otherwise issue them first and wait for all.  The group
            // sequence runs on into the workgroup's next tile (whose weight groups are the same ones again).
            const int gg = gbase + g;
            if (g + 1 < NG || has_next) lstoreW((gg + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG || has_next) gloadW(g + 2 < NG ? g + 2 : g + 2 - NG);
        };
        for (int ti = 0; ti < TPW; ++ti) {
            has_next = ti + 1 < TPW;
            if (has_next) {                                // the next tile's coordinates and patch addresses
                const int nb = bx + gridDim.x;
                const int nimg0 = (G::OHW >= G::PIX) ? nb / TILES_PER_IMG : nb * G::NIMG;
                const int nrow0 = (G::OHW >= G::PIX) ? (nb % TILES_PER_IMG) * G::TH : 0;
                tile_src(nimg0, nrow0 * S - p.P, xsrcN);
            }
            g = 0;
            for (int kb = 0; kb < KB; ++kb) {
                // invariant: patch of K step kb in Xs, weight group g in buffer (gbase+g)&1, rw = the next weight group (in flight)
                for (int kh = 0; kh < G::GPK - 1; ++kh, ++g) {
                    if (g < 14) X3_STAMP(8 + 4 * g);           // group start
                    computeSH((gbase + g) & 1, kh, two_pairs{}, [&] {
                        stageW();
                        if (g < 14) X3_STAMP(9 + 4 * g);       // first fragment reads issued, weights stored, next loads issued
                    });
                    if (g < 14) X3_STAMP(10 + 4 * g);          // MFMAs issued
                    lds_barrier();
                    if (g < 14) X3_STAMP(11 + 4 * g);          // barrier passed
                }
                const bool nextk = kb + 1 < KB;
                if (g < 14) X3_STAMP(8 + 4 * g);
                computeSH((gbase + g) & 1, G::GPK - 1, one_pair{}, [&] {
                    stageW();
                    if (nextk || has_next) gloadX(nextk ? kb + 1 : 0, !nextk);     // (the next tile's first patch behind the last K step)
                    if (g < 14) X3_STAMP(9 + 4 * g);
                });
                if (g < 14) X3_STAMP(10 + 4 * g);
                lds_barrier();
                if (g < 14) X3_STAMP(11 + 4 * g);
                if (nextk) {                                   // K step change: the patch is fully consumed
                    lstoreX(kb + 1);
                    lds_barrier();
                    X3_STAMP(4);                               // (first) K step change done
                }
                ++g;
            }
            epilogueSH();                                      // (its LDS use is red_s; the stores drain under what follows)
            if (has_next) {                                    // enter the next tile: its first patch sits in rx
                gbase += NG;
                bx += gridDim.x;
                img0 = (G::OHW >= G::PIX) ? bx / TILES_PER_IMG : bx * G::NIMG;
                row0 = (G::OHW >= G::PIX) ? (bx % TILES_PER_IMG) * G::TH : 0;
                in_row0 = row0 * S - p.P;
#pragma unroll
                for (int k = 0; k < XU; ++k) xsrc[k] = xsrcN[k];
                lstoreX(0);
                lds_barrier();
            }
        }
    } else {
        int kb = 0, kh = 0;
        for (int g = 0; g < NG; ++g) {
            // invariant: patch of K step kb in Xs, weight group g in buffer g&1, rw = weight group g+1 (in flight)
            const bool more = g + 1 < NG, last_row = kh == G::GPK - 1;
            if (more) lstoreW((g + 1) & 1);                // (order: see stageW above)
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG) gloadW(g + 2);
            if (kh == G::GPK - 2 && kb + 1 < KB) gloadX(kb + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(g & 1, kh * G::WP);
            lds_barrier();
            if (more && last_row) {                        // K step change: the patch is fully consumed
                lstoreX(kb + 1);
                lds_barrier();
            }
            if (++kh == G::GPK) { kh = 0; ++kb; }
        }
    }

    X3_STAMP(5);                                   // main loop done
    if constexpr (SH) {
        X3_STAMP(7);                               // end of the workgroup's program
        X3_STAMP_RT(65);
    } else {
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] += acs[b][r];

    // ---- epilogue: lane holds pixel l31 of each 32-pixel group, rows (channels) (r&3) + 8*(r>>2) + 4*half
    auto store_tile = [&](const float (&bv)[16], bool biased) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pix = (wave * MT + mt) * 32 + l31;
            const int im = pix / (G::TH * OW), rem = pix % (G::TH * OW);
            const int n = img0 + im;
            if (n >= p.N) continue;
            const int oy = row0 + rem / OW, ox = rem % OW;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.CoutReal) continue;
                p.out[(((long)n * p.CoutReal + o) * G::OH + oy) * OW + ox] = biased ? acc[mt][r] + bv[r] : acc[mt][r];
            }
        }
    };
    float bv[16];
    if (p.bias) {
        // the lane's 16 bias values from LDS; the bias-free directions keep the plain stores
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias_s[(r & 3) + 8 * (r >> 2) + 4 * half];
        store_tile(bv, true);
    } else {
        store_tile(bv, false);
    }

    // ---- optional BatchNorm statistics of this workgroup's tile (the loop ended with a barrier: LDS is free).  AFTER the
    // output stores: they drain while the sums are reduced, and no wave waits at the reduction's barrier with its tile unsent
    if (p.stats) {
        float* red = reinterpret_cast<float*>(lds_raw);       // [4 waves][32][2]
        float sv[32];                                         // [sum | sum of squares][register row]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { const float v = acc[mt][r]; s1 += v; s2 += v * v; }
            sv[r] = s1;
            sv[16 + r] = s2;
        }
        {   // lane l31 receives the half-wave total of sv[l31]
            const float tot = half_wave_reduce32(sv);
            const int r = l31 & 15, ch = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[(wave * 32 + ch) * 2 + (l31 >> 4)] = tot;
        }
        lds_barrier();                                        // LDS only: the output stores keep draining
        if (tid < 32 && o0 + tid < p.CoutReal) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[(w * 32 + tid) * 2]; s2 += red[(w * 32 + tid) * 2 + 1]; }
            float* dst = p.stats + ((long)(o0 + tid) * gridDim.x + bx) * 2;         // slot of the TILE: order independent of the mapping
            dst[0] = s1; dst[1] = s2;
        }
    }
    }
}

thread_local int g_x3_splits = 0;

template <int S, int OW, int MT, bool SH = false>
int launch_x3(const X3P& p, hipStream_t st) {
    using G = X3Geom<S, OW, MT, SH>;
    static_assert(S == 1 && G::LDS_BYTES + 2048 <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_x3_kernel<S, OW, MT, 0, SH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_x3_kernel<S, OW, MT, 1, SH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_x3_kernel<S, OW, MT, 2, SH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long pixels = (long)p.N * G::OHW;
    dim3 grid((unsigned)((pixels + G::PIX - 1) / G::PIX), (unsigned)(p.OP / 32));
    if (G::OHW < G::PIX) grid.x = (unsigned)((p.N + G::NIMG - 1) / G::NIMG);
    g_x3_splits = (int)grid.x;                                 // BatchNorm partial sums: one per TILE
    // Two tiles per workgroup (16x16x32 form, stride 1) when that still leaves two residency rounds of 512 workgroups:
    // the second tile's prologue hides under the first one's last weight groups.
    X3P q = p;
    q.tpw = 1;
    if (SH && grid.x % 2 == 0 && (long)grid.x * grid.y >= 2048) { q.tpw = 2; grid.x /= 2; }
    if (q.aff.sc) {
        if (q.Cin > 256) return JVAE_ENOTSUP;
        if (q.aff.relu == JVAE_ACT_LEAKY) hipLaunchKernelGGL((conv5_x3_kernel<S, OW, MT, 2, SH>), grid, dim3(256), G::LDS_BYTES, st, q);
        else hipLaunchKernelGGL((conv5_x3_kernel<S, OW, MT, 1, SH>), grid, dim3(256), G::LDS_BYTES, st, q);
    } else {
        hipLaunchKernelGGL((conv5_x3_kernel<S, OW, MT, 0, SH>), grid, dim3(256), G::LDS_BYTES, st, q);
    }
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#ifdef JVAE_X3_STAMPS
static unsigned long long* g_x3_dbg = nullptr;
extern "C" void jvae_x3_set_stamp_buffer(void* buf) { g_x3_dbg = (unsigned long long*)buf; }     // diagnostic build only
#endif
// JVAE_X3=0 (read once, when the library is loaded) or jvae_conv2d_set_split(0): every layer, dense product and weight gradient
// stays on the fp32 matrix-core kernels - the one selector left, exercised by the parity tests through the setter
static int g_x3 = [] { const char* e = getenv("JVAE_X3"); return (e && e[0] == '0') ? 0 : 1; }();
static int g_x3_sh16 = 1;    // jvae_conv2d_set_split_shape16(0): the 32x32x16 MFMA shape also for maps up to 32 wide (tests run both)

// Layers the split kernel takes over from conv_mfma.hip: stride 1, at least one full K step of input channels.
int jvae_conv5_x3_set(int mode) {
    const int old = g_x3;
    g_x3 = mode ? 1 : 0;
    return old;
}

bool jvae_conv5_x3_enabled() {
    return g_x3 != 0;
}

static bool x3_sh16() { return g_x3_sh16 != 0; }

// MFMA shape of the stride-1 forward-type kernel: 1 = v_mfma_f32_16x16x32_bf16 (K = 2 taps x 16 channels), 0 = 32x32x16.
// Returns the previous setting.
int jvae_conv5_x3_set_shape16(int on) {
    const int old = x3_sh16() ? 1 : 0;
    g_x3_sh16 = on ? 1 : 0;
    return old;
}

// split weights for conv5_x3_kernel / convt2_x3_kernel: ws must hold jvae_conv5_x3_pack_bytes(C, O)
int jvae_conv5_x3_wpack(const float* w, float* ws, int C, int O, int swap, int flip, hipStream_t st) {
    const int KB = (C + 15) / 16, OP = (O + 31) / 32 * 32;
    const long total = (long)KB * 25 * 2 * OP * 8;
    const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(x3_wpack_kernel, dim3(blocks), dim3(256), 0, st, w, (__bf16*)ws, C, O, KB, OP, swap, flip);
    JVAE_LAUNCH_CHECK();
    return 0;
}

bool jvae_conv5_x3_ok(int Cin, int H, int W, int Cout, int OH, int OW, int S, int P) {
    if (!g_x3) return false;
    if (Cin < 16 || Cin > 256) return false;
    // Stride-2 forward-type layers (E1 / E3 forward, D2 / D4 dgrad of conv32 / deconv32) stay on the fp32 matrix-core kernel: their
    // patch is 4x the output pixels, and both split-bf16 forms built for them (round 3: 128-pixel tile, one workgroup per CU; round 4:
    // 16x16x32 on a de-interleaved 64-pixel patch, two per CU) measured at best at parity with it (profiles/NOTES.md).
    if (S != 1) return false;
    if (OW != 8 && OW != 16 && OW != 32 && OW != 64) return false;
    return jvae_conv5_fwd_ok(Cin, H, W, Cout, OH, OW, S, P);
}

// bytes of the largest of the packed forms (kernel-row groups / tap-pair groups / position-sorted taps of the 4-phase kernel)
size_t jvae_conv5_x3_pack_bytes(int Cin, int Cout) {
    const size_t a = jvae_pack_bytes(JVAE_PACK_X3, Cin, Cout), b = jvae_pack_bytes(JVAE_PACK_X3S, Cin, Cout);
    const size_t c = jvae_pack_bytes(JVAE_PACK_T2S, Cin, Cout);      // (conv_t2_x3.hip's 32-channel K step)
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

static int x3s_wpack(const float* w, float* ws, int C, int O, int swap, int flip, hipStream_t st) {
    const long total = jvae_pack_elems(JVAE_PACK_X3S, C, O);
    const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(x3s_wpack_kernel, dim3(blocks), dim3(256), 0, st, w, (__bf16*)ws, C, O, total, swap, flip);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_conv5_x3_fwd(const float* in, const float* w, int swap, int flip, const float* bias, float* out,
                      int N, int Cin, int H, int W, int Cout, int OW, int S, int P, float* ws, hipStream_t st,
                      float* stats, int* nsplit, const InAff* aff) {
    const int OP = (Cout + 31) / 32 * 32;
    const bool sh = OW <= 32 && x3_sh16();                  // 64-wide maps (config 5 in fp32) keep the 32x32x16 form
    {   // split weights: the step's cache slot (refreshed once per step, pack_cache.hip) or this call's workspace
        bool fresh = true;
        const int kind = sh ? JVAE_PACK_X3S : JVAE_PACK_X3;
        float* slot = (float*)jvae_pack_cache_get(kind, w, Cin, Cout, swap, flip, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            const int rc = sh ? x3s_wpack(w, ws, Cin, Cout, swap, flip, st) : jvae_conv5_x3_wpack(w, ws, Cin, Cout, swap, flip, st);
            if (rc) return rc;
        }
    }
    X3P p{in, (const u32x4*)ws, bias, out, N, Cin, H, W, OP, P, Cout, stats, aff ? *aff : InAff{nullptr, nullptr, 0}, 1};
#ifdef JVAE_X3_STAMPS
    p.dbg = g_x3_dbg;
#endif
    struct Fin { int* n; ~Fin() { if (n) *n = g_x3_splits; } } fin{nsplit};
    if (S != 1) return JVAE_ENOTSUP;
    if (sh) {
        switch (OW) {
            case 8: return launch_x3<1, 8, 1, true>(p, st);
            case 16: return launch_x3<1, 16, 2, true>(p, st);
            case 32: return launch_x3<1, 32, 2, true>(p, st);
        }
    }
    switch (OW) {
        case 8: return launch_x3<1, 8, 1>(p, st);
        case 16: return launch_x3<1, 16, 2>(p, st);
        case 32: return launch_x3<1, 32, 2>(p, st);
        case 64: return launch_x3<1, 64, 1>(p, st);
    }
    return JVAE_ENOTSUP;
}
