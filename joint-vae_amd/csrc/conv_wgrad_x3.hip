// Weight gradient of the 5x5 (transposed) convolutions, fp32 in / fp32 out, on the bf16 matrix cores by exact 3-way
// operand splitting (the arithmetic of conv_x3.hip; the operator, roles and slab reduction of conv_wgrad_mfma.hip).
//
//   dW[a][b][kh][kw] = sum_{n,u,v} Ps[n][a][u][v] * Q[n][b][u*S + kh - P][v*S + kw - P]
//
// (Ps = tensor on the folded "small" grid, Q = the unfolded "big" one.  Reference: autograd of nn.Conv2d /
// nn.ConvTranspose2d, module/vae_layers/conv.py:186-196.)
//
// The contraction runs over PIXELS.  Both tensors are fp32 NCHW in HBM; while a tile is staged into LDS every value is
// (optionally) put through the deferred BatchNorm(+ReLU) of the layer input, split EXACTLY into three bf16 terms
// (v = hi + mid + lo, conv_x3.h) and stored as three planes of 16-byte units = 8 channels of one pixel - the pixel-major
// layout of conv_x3.hip / conv_wgrad_b8.hip.  In that layout both MFMA operands are k-major, which is what gfx950's
// ds_read_b64_tr_b16 is made for: per 16 lanes it reads 4 pixels x 16 channels and returns them column-major, so a lane
// receives 4 consecutive pixels of ITS channel; two such reads are one operand of v_mfma_f32_32x32x16_bf16 and a tap
// shift is a plain 16-byte-aligned unit offset (a bf16 NCHW patch would put the tap shift at a 2-byte granularity).
// Each fp32 product is accumulated from the six bf16 products that weigh >= 2^-24 of it (small ones first).
//
// A workgroup owns 32 channels `a` x one column group of (b, tap) and loops over its share of the images in tiles of
// TPIX pixels (full rows); its 4 waves split the column tiles of 32:
//   MODE 0 (S = 1, Cb >= 9): tile = 32 channels b x 1 tap   -> 25 tiles (wave w: taps w, w+4, ...)
//   MODE 2 (S = 2, Cb >= 9): tile = 16 channels b x 2 taps  -> 13 tiles (the stride-2 patch is 4x the pixels: 16 channels
//                                                              keep two workgroups per CU)
//   MODE 1 (4 < Cb <= 8):    tile = 8 channels b x 4 taps   ->  7 tiles
//   MODE 3 (Cb <= 4):        tile = 4 channels b x 8 taps   ->  4 tiles, one per wave (3-channel image side of the first / last
//                            layer; round 4).  The transposing read takes 8 bytes = 4 channels per lane at the lane's OWN address,
//                            so a 16-column block is four taps' first half units (the 16-byte units hold 8 channels, the upper
//                            four are padding that is simply never read): 75 real columns in 4 x 32 instead of 7 x 32 - the
//                            MFMA work of these layers falls by 3/7 (it was two thirds padding).
// The A fragments (3 planes x 2 transposed reads per 16 pixels) are shared by all tiles of a wave; every tile needs its
// own B fragments.  Accumulators stay in registers over the whole image loop; each workgroup writes one fp32 slab,
// reduced in a fixed order by jvae_wgrad_slab_reduce (deterministic, no float atomics).
#include <stdlib.h>
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"
#include "conv_x3.h"

namespace {

typedef x3_bf16x8 bf16x8;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef x3_u32x4 u32x4;
typedef x3_f32x2 f32x2;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct WgX3P {
    const void* ps;      // (N, Ca, HS, WS) fp32 NCHW, or B8 units (N, ceil(Ca/8), HS, WS) for the one-plane (bf16) form
    const void* q;       // (N, Cb, HB, WB) / B8 units (N, ceil(Cb/8), HB, WB)
    float* slab;         // (G, Ca, 25, Cb)
    int N, Ca, Cb, P, G;
    InAff aff_p, aff_q;  // deferred BatchNorm(+ReLU) of ps / q (whichever is the layer input; sc == nullptr: none)
};

template <int S, int WS, int MODE, int NPL = 3>
struct WgX3Geom {
    static constexpr int HS = WS;
    static constexpr int TPIX = (S == 1 && WS >= 16) ? 128 : 64;
    static constexpr int TH = TPIX / WS;
    static constexpr int TILES = HS * WS / TPIX;
    static constexpr int ROWS = (TH - 1) * S + 5;
    static constexpr int WB = WS * S;
    // patch row: image column x is padded column P + x (P halo columns on the left), so a tap reads padded column
    // v*S + kw whatever the padding: WP = (WS-1)*S + 5 columns per row; the host checks P + WB <= WP.  Stride 2 stores the
    // columns de-interleaved (even ones first: slot = (col & 1)*WPH + col/2) so that the pixels v, v+1, .. of one tap are
    // consecutive slots as for stride 1.  The NCBQ channel blocks of a slot are adjacent units ([row][slot][block]): the 32
    // lanes of one transposed read (4 pixels x NCBQ blocks x 2 halves) then cover distinct banks (a [block][row][slot]
    // layout put all blocks on the same banks: 4-way conflicts, LDS-bound; profiles/r02_wgrad_x3_pmc_v1.json).
    static constexpr int WP = (WS - 1) * S + 5;
    static constexpr int WPH = (WP + 1) / 2;
    static constexpr int WPS = S == 1 ? WP : 2 * WPH;          // slots per row
    static constexpr int CH = ROWS * WPS;                      // slots per plane
    static constexpr int NCBQ = MODE == 0 ? 4 : (MODE == 2 ? 2 : 1);   // 8-channel blocks of Q staged per item
    static constexpr int QS = NCBQ * CH;                       // units per plane
    static constexpr int PS = 4 * TPIX;                        // units per plane (32 channels a): [pixel][block]
    static constexpr int NTILE = MODE == 0 ? 25 : (MODE == 2 ? 13 : (MODE == 3 ? 4 : 7));
    static constexpr int NBT = (NTILE + 3) / 4;                // per wave
    static constexpr int LDS_BYTES = NPL * (QS + PS) * 16;
    static constexpr int QITEMS = NCBQ * ROWS * (WB / 2);      // staging items: 2 pixels x 8 channels
    static constexpr int PITEMS = 4 * (TPIX / 2);
    static constexpr int QUNITS = NCBQ * ROWS * WB;            // one-plane form: staging items = B8 units
    static constexpr int PUNITS = 4 * TPIX;
};

// HI = byte distance of 4 pixels (4 slots x the channel blocks per slot x 16 bytes)
template <int HI>
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* base, int off) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + HI));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// SH16: the products run on v_mfma_f32_16x16x32_bf16 (K step = 32 pixels, a 32 x 32 tile = 2 x 2 blocks of 16 x 16) instead of
// v_mfma_f32_32x32x16_bf16: the same LDS image, the same reads and MFMA cycles per FLOP; the chip holds a higher clock under
// the 16x16x32 shape (MI355X_MICROARCH.md, DVFS give-back item 7).  (The 32x32x16 form of this kernel left the tree in round 5.)
// NPL = 1: the operands are bf16 "B8" tensors (conv_b8.hip: 16-byte units of 8 channels per pixel - exactly the units of the
// LDS image): no split, ONE plane, one MFMA per product; everything else (LDS image, transposed reads, read-ahead, slabs) is
// shared with the split-bf16 form.  This is the weight-gradient kernel of the bf16 mode (BASELINE configs[4]).
//
// SWZ (16x16x32 form, 32 channels b x 1 tap, stride 1, maps of 16 / 32 / 64; both the split and the one-plane form).  The two
// 16-lane groups of a 32-lane transposed read fetch 4-pixel blocks 8 slots apart = 512 bytes = the same banks: a 2-way conflict
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.58 in the one-plane form, profiles/r02_wgrad_b8_pmc_v1.json; 0.49 in the split
// form, profiles/r02_wgrad_x3_pmc.json).  The image therefore swaps the two 32-byte halves of a slot (channel blocks {0,1} <->
// {2,3}) in slots whose COLUMN (padded column for Q, pixel index for Ps) has bit 3 set, and the K index of the MFMA is dealt to
// the lanes so that a lane never needs two different halves: the MFMA only requires that A and B agree on which pixel a K
// index is, so lane group g supplies the 4-pixel blocks {0,2,1,3}[g] and that + 4 (16 pixels = 16 slots or one row further:
// bit 3 of the column unchanged) of each 32-pixel K step.  The groups of one 32-lane access then sit 8 columns apart (opposite
// halves, whatever the tap shift), a lane's two reads share ONE address (+ an immediate), and the half is a constant of
// (lane, tap) folded into the tile's byte offset; the second 16-channel block is that offset ^ 32.  Keyed on the column, not
// the slot index, the swizzle needs no padded row pitch (round 2's one-plane version padded it to 16 slots, which the three
// planes of the split form cannot afford within 80 KB).
template <int S, int WS, int MODE, int AFF, bool SH16, int NPL = 3>     // AFF: 0 plain operands, 1 deferred BatchNorm (+ReLU by flag), 2 ... with a leaky ReLU (fp32 form only)
__global__ __launch_bounds__(256, 2) void conv5_wgrad_x3_kernel(WgX3P p) {
    using G = WgX3Geom<S, WS, MODE, NPL>;
    constexpr bool SWZ = SH16 && MODE == 0 && S == 1 && WS >= 16;
    static_assert(NPL == 3 || (NPL == 1 && SH16), "one-plane form: 16x16x32 shape only");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Qs = reinterpret_cast<u32x4*>(lds_raw);             // [NPL planes][QS]
    u32x4* Pt = Qs + NPL * G::QS;                              // [NPL planes][PS]
    const unsigned char* Qb = lds_raw;
    const unsigned char* Pb = lds_raw + NPL * G::QS * 16;
    constexpr int NT8 = 8 * (G::NCBQ + 4);
    __shared__ __attribute__((aligned(16))) float ctab[AFF ? 2 * NT8 : 4];   // (scale, shift) of this workgroup's q / ps channels

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int a0 = blockIdx.y * 32;
    const int cbq0 = blockIdx.z * G::NCBQ;                     // first 8-channel block of Q of this workgroup

    // halo columns are zeroed once and never written again (rows are rewritten per item, with zeros outside the image)
    for (int i = tid; i < NPL * G::QS; i += 256) Qs[i] = u32x4{0u, 0u, 0u, 0u};
    if (AFF && tid < NT8) {
        constexpr int NQ = 8 * G::NCBQ;
        const bool isq = tid < NQ;
        const InAff& a = isq ? p.aff_q : p.aff_p;
        const int ch = isq ? cbq0 * 8 + tid : a0 + (tid - NQ);
        const bool ok = a.sc && ch < (isq ? p.Cb : p.Ca);
        ctab[tid] = ok ? a.sc[ch] : 0.f;
        ctab[NT8 + tid] = ok ? a.sh[ch] : 0.f;
    }

    // transposed-read roles of this lane: row (pixel) q4 of the 4x16 block, channel quad pp; cg = 16-column group
    const int cg = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int pl = 8 * half + q4;                              // pixel of this lane inside a 16-pixel K step
    const int aoff = (pl * 4 + cg * 2 + (pp >> 1)) * 16 + (pp & 1) * 8;
    const int lane_pix = (pl / WS) * S * G::WPS + (pl % WS);             // slot; WS = 8: the K step spans two rows
    static_assert(WS >= 8, "a lane's 8 pixels must lie in one row");

    int boff[G::NBT];
#pragma unroll
    for (int t = 0; t < G::NBT; ++t) {
        const int tile = wave + 4 * t;
        int tap, cbl, sub = pp & 1;
        if (MODE == 0) { tap = tile; cbl = cg * 2 + (pp >> 1); }
        else if (MODE == 2) { tap = tile * 2 + cg; cbl = pp >> 1; }
        else if (MODE == 3) { tap = tile * 8 + cg * 4 + pp; cbl = 0; sub = 0; }
        else { tap = tile * 4 + cg * 2 + (pp >> 1); cbl = 0; }
        if (tap > 24) tap = 24;                                // unused slots: any valid address
        const int kw = tap % 5;
        const int tslot = S == 1 ? kw : (kw & 1) * G::WPH + (kw >> 1);
        boff[t] = ((lane_pix + (tap / 5) * G::WPS + tslot) * G::NCBQ + cbl) * 16 + sub * 8;
    }

    // 16x16x32 roles: 16-lane group g16 = K group (8 pixels), the lanes of a group supply (pixel q4, channel quad pp) of a
    // 16-channel block chosen per MFMA (rb for Ps, cb for the columns)
    const int g16 = lane >> 4, c16 = lane & 15;
    // pixel of this lane inside a 32-pixel K step (its second four pixels: + 4, swizzled image: + 16)
    const int pl16 = SWZ ? 4 * ((g16 >> 1) | (g16 & 1) << 1) + q4 : 8 * g16 + q4;
    const int aoff16 = (pl16 * 4 + (pp >> 1)) * 16 + (pp & 1) * 8;        // + rb * 32 bytes
    const int lane_pix16 = (pl16 / WS) * S * G::WPS + (pl16 % WS);
    int boff16[SH16 ? G::NBT : 1][MODE == 0 ? 1 : 2];      // MODE 0: the second column block is 2 units (32 bytes) further
    if (SH16) {
#pragma unroll
        for (int t = 0; t < G::NBT; ++t)
#pragma unroll
            for (int cb = 0; cb < (MODE == 0 ? 1 : 2); ++cb) {
                const int tile = wave + 4 * t;
                int tap, blk;
                int sub16 = pp & 1;
                if (MODE == 0) { tap = tile; blk = cb * 2 + (pp >> 1); }
                else if (MODE == 2) { tap = wave + 4 * (2 * t + cb); blk = pp >> 1; }     // 16-column blocks dealt tap by tap (below)
                else if (MODE == 3) { tap = tile * 8 + cb * 4 + pp; blk = 0; sub16 = 0; }
                else { tap = tile * 4 + cb * 2 + (pp >> 1); blk = 0; }
                if (tap > 24) tap = 24;
                const int kw = tap % 5;
                const int tslot = S == 1 ? kw : (kw & 1) * G::WPH + (kw >> 1);
                const int slot = lane_pix16 + (tap / 5) * G::WPS + tslot;
                if (SWZ) {
                    const int bit = ((pl16 % WS) + kw) >> 3 & 1;       // bit 3 of the padded column (both reads of the lane)
                    boff16[t][cb] = (slot * G::NCBQ + (blk ^ (bit << 1))) * 16 + sub16 * 8;
                } else {
                    boff16[t][cb] = (slot * G::NCBQ + blk) * 16 + sub16 * 8;
                }
            }
    }
    const int aswz = (SWZ && (pl16 >> 3 & 1)) ? 32 : 0;        // Ps: row block rb sits at (rb ^ bit 3 of the pixel) * 32 bytes

    f32x16 acc[SH16 ? 1 : G::NBT];
    f32x4 acc16[SH16 ? G::NBT : 1][2][2];
    f32x4 accq = {0.f, 0.f, 0.f, 0.f};                         // MODE 0, 16x16x32: this wave's 16x16 block of tap 24
    if (SH16) {
#pragma unroll
        for (int t = 0; t < G::NBT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc16[t][i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int t = 0; t < G::NBT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }

    const ImageRange ir = image_range(p.N, p.G, blockIdx.x);   // trailing workgroups may be empty
    const int n_beg = ir.nb, n_end = ir.ne;
    constexpr int QU = (G::QITEMS + 255) / 256, PU = (G::PITEMS + 255) / 256;
    constexpr int HB = G::HS * S;
    constexpr int W2 = G::WB / 2;
    const long qcs = (long)HB * G::WB, pcs = (long)G::HS * WS;  // channel strides
    f32x2 rq[QU][8], rp[PU][8];                 // (dead in the one-plane form)
    // Stride 2 (round 4): the channel block of a thread's staging items is a function of the thread index alone (u % NCBQ and
    // u % 4 with u = tid + 256 k), and only the layer-input side carries a deferred BatchNorm: its 8 (scale, shift) pairs live in
    // 16 registers for the whole kernel. The per-channel LDS reads inside the per-lane branch cost these launches 10-12 us each
    // (74 vs 64 us on E1); the stride-1 forms have no registers to spare and keep the table in LDS.
    constexpr bool REGC = AFF && S == 2 && NPL == 3;
    static_assert(!REGC || (256 % G::NCBQ == 0), "one channel block per thread");
    float cr[16];
    if constexpr (REGC) {
        const bool onq = p.aff_q.sc != nullptr;
        const InAff& a = onq ? p.aff_q : p.aff_p;
        const int ch0 = onq ? (cbq0 + threadIdx.x % G::NCBQ) * 8 : a0 + (threadIdx.x % 4) * 8;
        const int cn = onq ? p.Cb : p.Ca;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const bool ok = a.sc && ch0 + ci < cn;
            cr[ci] = ok ? a.sc[ch0 + ci] : 0.f;
            cr[8 + ci] = ok ? a.sh[ch0 + ci] : 0.f;
        }
    }
    const float* const qf = (const float*)p.q;
    const float* const psf = (const float*)p.ps;
    // one-plane form: B8 units straight from HBM
    constexpr int QU8 = (G::QUNITS + 255) / 256, PU8 = (G::PUNITS + 255) / 256;
    u32x4 rq8[QU8], rp8[PU8];                   // (dead in the split form)
    const u32x4* const q8 = (const u32x4*)p.q;
    const u32x4* const ps8 = (const u32x4*)p.ps;
    const int CBa = (p.Ca + 7) / 8, CBb = (p.Cb + 7) / 8;

    // Loads are unconditional (out-of-range items read a valid stand-in address and are zeroed when they are stored to
    // LDS) so that the compiler can count the loads in flight.
    auto gload = [&](int item) {
        int tid = threadIdx.x;
        if constexpr (NPL == 3) asm volatile("" : "+v"(tid));    // as in lstore
        const int n = item / G::TILES, tile = item % G::TILES;
        const int row0 = tile * G::TH;
        const int in_row0 = row0 * S - p.P;
        if constexpr (NPL == 1) {
#pragma unroll
            for (int k = 0; k < QU8; ++k) {                    // channel block fastest: adjacent lanes -> adjacent LDS units
                const int u = tid + k * 256;                   // (x fastest makes the 16-byte LDS stores 4-way conflicted)
                const int c = u % G::NCBQ, x = (u / G::NCBQ) % G::WB;
                const int lr = u / (G::NCBQ * G::WB);
                const int ir = in_row0 + lr, cb = cbq0 + c;
                const bool ok = u < G::QUNITS && ir >= 0 && ir < HB && cb < CBb;
                rq8[k] = ok ? q8[(((long)n * CBb + cb) * HB + ir) * G::WB + x] : q8[0];     // stand-in address: zeroed in lstore
            }
#pragma unroll
            for (int k = 0; k < PU8; ++k) {
                const int u = tid + k * 256;
                const int c = u % 4, px = u / 4;
                const int cb = (a0 >> 3) + c;
                const bool ok = u < G::PUNITS && cb < CBa;
                rp8[k] = ok ? ps8[(((long)n * CBa + cb) * G::HS + row0) * WS + px] : ps8[0];
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < QU; ++k) {
            const int u = tid + k * 256;
            const int c = u % G::NCBQ;                         // channel block fastest: adjacent lanes -> adjacent LDS units
            const int xp = (u / G::NCBQ) % W2;
            const int lr = u / (G::NCBQ * W2);
            const int ir = in_row0 + lr, ch0 = (cbq0 + c) * 8;
            const bool ok = u < G::QITEMS && ir >= 0 && ir < HB;
            const float* src = qf + (((long)n * p.Cb + ch0) * HB + (ok ? ir : 0)) * G::WB + 2 * xp;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
                rq[k][ci] = *reinterpret_cast<const f32x2*>((ok && ch0 + ci < p.Cb) ? src + ci * qcs : qf);
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int u = tid + k * 256;
            const int c = u % 4, px2 = u / 4;
            const int ch0 = a0 + c * 8;
            const bool ok = u < G::PITEMS;
            const float* src = psf + (((long)n * p.Ca + ch0) * G::HS + row0) * WS + 2 * px2;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
                rp[k][ci] = *reinterpret_cast<const f32x2*>((ok && ch0 + ci < p.Ca) ? src + ci * pcs : psf);
        }
    };
    // 2 pixels x 8 channels of fp32 -> [deferred BatchNorm] -> three bf16 planes, two 16-byte units each
    // sc / sh: the item's (scale, shift) in LDS (ctab) - or, with REGC (stride 2), in registers (creg: loaded once per kernel)
    auto split_store = [&](const f32x2 (&r)[8], bool live, int nch, const float* sc, const float* sh, int relu,
                           bool aff, u32x4* dst0, u32x4* dst1, int plane_stride, const float (&creg)[16]) {
        f32x2 vv[8];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            f32x2 v = (live && ci < nch) ? r[ci] : f32x2{0.f, 0.f};
            // (round 4: reading the 8 (scale, shift) pairs as four 16-byte LDS vectors up front made the stride-2 launches 3 % faster
            // and run-to-run NON-deterministic.  Round 5 found why: behind such reads the compiler broadcasts the odd coefficient of a
            // register pair with `v_pk_fma_f32 ... op_sel:[0,1,1]`, a form whose low half is not reliable on this chip -
            // profiles/NOTES.md; tools/isa_opsel_scan.py keeps it out of the build.  Stride 2 now has the coefficients in registers.)
            if constexpr (REGC) {
                if (aff) {
                    f32x2 a{fmaf(v[0], creg[ci], creg[8 + ci]), fmaf(v[1], creg[ci], creg[8 + ci])};
                    if constexpr (AFF == 2) { a[0] = jvae_act(a[0], relu); a[1] = jvae_act(a[1], relu); }
                    else if (relu) { a[0] = fmaxf(a[0], 0.f); a[1] = fmaxf(a[1], 0.f); }
                    v = (live && ci < nch) ? a : f32x2{0.f, 0.f};         // padding rows / missing channels stay exact zeros
                }
            } else if (AFF && aff && live && ci < nch) {
                v[0] = fmaf(v[0], sc[ci], sh[ci]);
                v[1] = fmaf(v[1], sc[ci], sh[ci]);
                if constexpr (AFF == 2) { v[0] = jvae_act(v[0], relu); v[1] = jvae_act(v[1], relu); }
                else if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
            }
            vv[ci] = v;
        }
        u32x4 s[2][3];                              // [pixel][plane]: 8 channels = 4 packed pairs (x3_split2: two values at once)
#pragma unroll
        for (int cp = 0; cp < 4; ++cp)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned hh, mm, ll;
                x3_split2(f32x2{vv[2 * cp][j], vv[2 * cp + 1][j]}, hh, mm, ll);
                s[j][0][cp] = hh; s[j][1][cp] = mm; s[j][2][cp] = ll;
            }
#pragma unroll
        for (int plane = 0; plane < 3; ++plane) {
            dst0[plane * plane_stride] = s[0][plane];
            dst1[plane * plane_stride] = s[1][plane];
        }
    };
    auto lstore = [&](int item) {                  // item: the work item whose data sits in rq / rp
        // (opaque copy of the thread index, split form: the LDS addresses are re-derived per item instead of being hoisted
        // out of the image loop into registers this kernel does not have)
        int tid = threadIdx.x;
        if constexpr (NPL == 3) asm volatile("" : "+v"(tid));
        const int in_row0 = (item % G::TILES) * G::TH * S - p.P;
        if constexpr (NPL == 1) {
#pragma unroll
            for (int k = 0; k < QU8; ++k) {
                const int u = tid + k * 256;
                if (u < G::QUNITS) {
                    const int c = u % G::NCBQ, x = (u / G::NCBQ) % G::WB;
                    const int lr = u / (G::NCBQ * G::WB);
                    const int ir = in_row0 + lr;
                    const bool live = ir >= 0 && ir < HB && cbq0 + c < CBb;      // padding rows / missing blocks: exact zeros
                    u32x4 v = live ? rq8[k] : u32x4{0u, 0u, 0u, 0u};
                    if (AFF && p.aff_q.sc && live) v = aff8(v, &ctab[c * 8], &ctab[NT8 + c * 8], p.aff_q.relu);
                    const int c0 = p.P + x;
                    const int sl = S == 1 ? c0 : (c0 & 1) * G::WPH + (c0 >> 1);
                    const int slot = lr * G::WPS + sl;
                    Qs[slot * G::NCBQ + (SWZ ? c ^ ((sl >> 3 & 1) << 1) : c)] = v;
                }
            }
#pragma unroll
            for (int k = 0; k < PU8; ++k) {
                const int u = tid + k * 256;
                if (u < G::PUNITS) {
                    const int c = u % 4, px = u / 4;
                    const bool live = (a0 >> 3) + c < CBa;
                    u32x4 v = live ? rp8[k] : u32x4{0u, 0u, 0u, 0u};
                    if (AFF && p.aff_p.sc && live)
                        v = aff8(v, &ctab[(G::NCBQ + c) * 8], &ctab[NT8 + (G::NCBQ + c) * 8], p.aff_p.relu);
                    Pt[px * 4 + (SWZ ? c ^ ((px >> 3 & 1) << 1) : c)] = v;
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < QU; ++k) {
            const int u = tid + k * 256;
            if (u < G::QITEMS) {
                const int c = u % G::NCBQ;
                const int xp = (u / G::NCBQ) % W2;
                const int lr = u / (G::NCBQ * W2);
                const int ir = in_row0 + lr;
                const bool live = ir >= 0 && ir < HB;          // padding rows / missing channels stay exact zeros
                const int c0 = p.P + 2 * xp, c1 = c0 + 1;      // padded columns of the two pixels
                const int s0 = S == 1 ? c0 : (c0 & 1) * G::WPH + (c0 >> 1), s1 = S == 1 ? c1 : (c1 & 1) * G::WPH + (c1 >> 1);
                const int z0 = SWZ ? (s0 >> 3 & 1) << 1 : 0, z1 = SWZ ? (s1 >> 3 & 1) << 1 : 0;    // swizzle: keyed on the column
                split_store(rq[k], live, p.Cb - (cbq0 + c) * 8, AFF ? &ctab[c * 8] : ctab, AFF ? &ctab[NT8 + c * 8] : ctab, p.aff_q.relu,
                            p.aff_q.sc != nullptr, &Qs[(lr * G::WPS + s0) * G::NCBQ + (c ^ z0)], &Qs[(lr * G::WPS + s1) * G::NCBQ + (c ^ z1)], G::QS,
                            cr);
            }
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int u = tid + k * 256;
            if (u < G::PITEMS) {
                const int c = u % 4, px2 = u / 4;
                const int zp = SWZ ? (px2 >> 2 & 1) << 1 : 0;      // bit 3 of the pixel index (shared by the two pixels)
                split_store(rp[k], true, p.Ca - (a0 + c * 8), AFF ? &ctab[(G::NCBQ + c) * 8] : ctab, AFF ? &ctab[NT8 + (G::NCBQ + c) * 8] : ctab,
                            p.aff_p.relu, p.aff_p.sc != nullptr, &Pt[(2 * px2) * 4 + (c ^ zp)], &Pt[(2 * px2 + 1) * 4 + (c ^ zp)], G::PS,
                            cr);
            }
        }
    };

    const int item_beg = n_beg * G::TILES, item_end = n_end * G::TILES;
    if (item_beg < item_end) gload(item_beg);
    for (int item = item_beg; item < item_end; ++item) {
        __syncthreads();                           // every wave is past the fragments of the previous item
        lstore(item);
        __syncthreads();
        if (!SH16 && item + 1 < item_end) gload(item + 1);
        if constexpr (SH16) {
            // slots = (K step of 32 pixels, tile, 16-column block): the fragments of the next slot are read before the 12
            // MFMAs (2 row blocks x 6 products) of the current one are issued
            // MODE 0 (32 channels x 1 tap per tile, 25 tiles): 25 taps over 4 waves is 7 / 6 / 6 / 6 - wave 0 kept its SIMD busy for
            // seven tiles while the other three idled through the seventh.  Tap 24 is therefore split by 16x16 BLOCK (round 4): wave w
            // takes block (rb, cb) = (w >> 1, w & 1) of it - 6 full taps + a quarter = 150 MFMAs per K step and wave instead of 168 / 144.
            constexpr bool QS = MODE == 0;
            constexpr int KS = G::TPIX / 32, NSLOT = QS ? (G::NBT - 1) * 2 + 1 : G::NBT * 2;
            bf16x8 a[2][NPL], b[2][NPL];
            const int bq = QS ? (SWZ ? ((wave & 1) ? boff16[G::NBT - 1][0] ^ 32 : boff16[G::NBT - 1][0]) : boff16[G::NBT - 1][0] + (wave & 1) * 32) : 0;
            auto read_a = [&](int ks) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int plane = 0; plane < NPL; ++plane)
                        a[rb][plane] = tr_pair<SWZ ? 1024 : 256>(Pb + plane * G::PS * 16, aoff16 + (rb * 32 ^ aswz) + ks * 32 * 64);
            };
            auto read_b = [&](int ks, int slot, bf16x8 (&d)[NPL]) {
                const int pix0 = ks * 32;
                const int qoff = ((pix0 / WS) * S * G::WPS + (pix0 % WS)) * G::NCBQ * 16;
                // second four pixels of the lane: 16 pixels further = 16 slots (maps of 32 / 64) or one row (maps of 16)
                constexpr int HI16 = (WS >= 32 ? 16 : G::WPS) * G::NCBQ * 16;
#pragma unroll
                for (int plane = 0; plane < NPL; ++plane) {
                    if (QS && slot == NSLOT - 1)
                        d[plane] = tr_pair<SWZ ? HI16 : 64 * G::NCBQ>(Qb + plane * G::QS * 16, bq + qoff);
                    else if constexpr (SWZ)
                        d[plane] = tr_pair<HI16>(Qb + plane * G::QS * 16, ((slot & 1) ? boff16[slot >> 1][0] ^ 32 : boff16[slot >> 1][0]) + qoff);
                    else
                        d[plane] = tr_pair<64 * G::NCBQ>(Qb + plane * G::QS * 16,
                                                         (MODE == 0 ? boff16[slot >> 1][0] + (slot & 1) * 32 : boff16[slot >> 1][slot & 1]) + qoff);
                }
            };
            read_a(0);
            read_b(0, 0, b[0]);
            // the next item's global loads (24 of them, with their address arithmetic) are issued BEHIND the item's first fragment
            // reads, whose LDS round trip they cover (round 4, as the weight staging in conv_x3.hip)
            __builtin_amdgcn_sched_barrier(0);
            if (item + 1 < item_end) gload(item + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int slot = 0; slot < NSLOT; ++slot) {
                    const int cur = (ks * NSLOT + slot) & 1;
                    const int t = slot >> 1, cb = slot & 1;
                    if (slot + 1 < NSLOT) read_b(ks, slot + 1, b[cur ^ 1]);
                    else if (ks + 1 < KS) read_b(ks + 1, 0, b[cur ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // MODE 2 (16 channels x 1 tap per 16-column block): the 25 taps are dealt to the waves block by block - tap
                    // = wave + 4 * slot: 7 / 6 / 6 / 6 blocks - instead of as 13 tiles of two taps (8 / 6 / 6 / 6 with the empty tap 25)
                    constexpr int NPROD = NPL == 3 ? 6 : 1;            // one-plane form: the product itself
                    constexpr int APL[6] = {0, 2, 1, 0, 1, 0}, BPL[6] = {2, 0, 1, 1, 0, 0};
                    if (QS && slot == NSLOT - 1) {                     // the wave's quarter of tap 24 (row block by select: no branch)
                        bf16x8 aq[NPL];
#pragma unroll
                        for (int plane = 0; plane < NPL; ++plane) {            // dword by dword: a select of whole vectors became a
                            const u32x4 x1 = __builtin_bit_cast(u32x4, a[1][plane]), x0 = __builtin_bit_cast(u32x4, a[0][plane]);
                            u32x4 xs;                                          // run-time indexed stack array in the one-plane form
#pragma unroll
                            for (int e = 0; e < 4; ++e) xs[e] = (wave & 2) ? x1[e] : x0[e];
                            aq[plane] = __builtin_bit_cast(bf16x8, xs);
                        }
#pragma unroll
                        for (int m = 0; m < NPROD; ++m)
                            accq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[NPL == 3 ? APL[m] : 0], b[cur][NPL == 3 ? BPL[m] : 0], accq, 0, 0, 0);
                    } else if (QS || (MODE == 2 ? wave + 4 * slot <= 24 : wave + 4 * t < G::NTILE)) {         // wave-uniform
#pragma unroll
                        for (int m = 0; m < NPROD; ++m)
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
                                acc16[t][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb][NPL == 3 ? APL[m] : 0], b[cur][NPL == 3 ? BPL[m] : 0],
                                                                                           acc16[t][rb][cb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (slot + 1 == NSLOT && ks + 1 < KS) read_a(ks + 1);           // the Ps fragments are free again
                }
            }
        } else if constexpr (NPL == 3) {
        // Software-pipelined over the (K step, tile) sequence: the fragments of the NEXT slot are read before the six
        // MFMAs of the current one are issued (hipcc would otherwise place every read right in front of its consumer and
        // wait for it with the matrix pipe idle).  Slots beyond a wave's tiles read a valid stand-in address.
        constexpr int KS = G::TPIX / 16;
        bf16x8 a[2][3], b[2][3];
        auto read_a = [&](int ks, bf16x8 (&d)[3]) {
#pragma unroll
            for (int plane = 0; plane < 3; ++plane) d[plane] = tr_pair<256>(Pb + plane * G::PS * 16, aoff + ks * 16 * 64);
        };
        auto read_b = [&](int ks, int t, bf16x8 (&d)[3]) {
            const int pix0 = ks * 16;
            const int qoff = ((pix0 / WS) * S * G::WPS + (pix0 % WS)) * G::NCBQ * 16;    // compile-time after unrolling
#pragma unroll
            for (int plane = 0; plane < 3; ++plane) d[plane] = tr_pair<64 * G::NCBQ>(Qb + plane * G::QS * 16, boff[t] + qoff);
        };
        read_a(0, a[0]);
        read_b(0, 0, b[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < G::NBT; ++t) {
                const int cur = (ks * G::NBT + t) & 1;
                if (t + 1 < G::NBT) read_b(ks, t + 1, b[cur ^ 1]);
                else if (ks + 1 < KS) { read_a(ks + 1, a[(ks + 1) & 1]); read_b(ks + 1, 0, b[cur ^ 1]); }
                __builtin_amdgcn_sched_barrier(0);
                if (wave + 4 * t < G::NTILE) {                                     // wave-uniform
                    // (Ps plane, Q plane): the three small products first, then the large ones
                    constexpr int APL[6] = {0, 2, 1, 0, 1, 0}, BPL[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][APL[m]], b[cur][BPL[m]], acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
    }

    // slab[g][a][tap][b] (b contiguous over the lanes: coalesced); lane holds column l31, rows a = (r&3) + 8*(r>>2) + 4*half
    float* slab = p.slab + (long)blockIdx.x * p.Ca * (p.Cb * 25);
    if constexpr (SH16) {
        // 16x16 block (rb, cb): lane holds column c16 of the block, rows 4 * g16 + r
        if constexpr (MODE == 0) {                             // the wave's block (rb, cb) = (wave >> 1, wave & 1) of tap 24
            const int b = cbq0 * 8 + (wave & 1) * 16 + c16;
            if (b < p.Cb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + (wave >> 1) * 16 + g16 * 4 + r;
                    if (a < p.Ca) slab[((long)a * 25 + 24) * p.Cb + b] = accq[r];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < (MODE == 0 ? G::NBT - 1 : G::NBT); ++t) {
            const int tile = wave + 4 * t;
            if (MODE != 2 && tile >= G::NTILE) continue;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int j = cb * 16 + c16;
                int b, tap;
                if (MODE == 0) { b = cbq0 * 8 + j; tap = tile; }
                else if (MODE == 2) { b = cbq0 * 8 + c16; tap = wave + 4 * (2 * t + cb); }
                else if (MODE == 3) { b = j & 3; tap = tile * 8 + (j >> 2); }
                else { b = j & 7; tap = tile * 4 + (j >> 3); }
                if (b >= p.Cb || tap > 24) continue;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = a0 + rb * 16 + g16 * 4 + r;
                        if (a < p.Ca) slab[((long)a * 25 + tap) * p.Cb + b] = acc16[t][rb][cb][r];
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < G::NBT; ++t) {
        const int tile = wave + 4 * t;
        if (tile >= G::NTILE) continue;
        int b, tap;
        if (MODE == 0) { b = cbq0 * 8 + l31; tap = tile; }
        else if (MODE == 2) { b = cbq0 * 8 + (l31 & 15); tap = tile * 2 + (l31 >> 4); }
        else if (MODE == 3) { b = l31 & 3; tap = tile * 8 + (l31 >> 2); }
        else { b = l31 & 7; tap = tile * 4 + (l31 >> 3); }
        if (b >= p.Cb || tap > 24) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int a = a0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (a < p.Ca) slab[((long)a * 25 + tap) * p.Cb + b] = acc[t][r];
        }
    }
}

template <int S, int WS, int MODE, int NPL = 3>
int launch_wgx3(const WgX3P& p, hipStream_t st) {
    using G = WgX3Geom<S, WS, MODE, NPL>;
    static_assert(G::LDS_BYTES + 2048 <= 80 * 1024, "two workgroups per CU must fit the 160 KB LDS");
    dim3 grid(p.G, (p.Ca + 31) / 32, (p.Cb + 8 * G::NCBQ - 1) / (8 * G::NCBQ));
    if (MODE == 1 || MODE == 3) grid.z = 1;
    const bool aff = p.aff_p.sc || p.aff_q.sc;
    const bool leaky = (p.aff_p.sc && p.aff_p.relu == JVAE_ACT_LEAKY) || (p.aff_q.sc && p.aff_q.relu == JVAE_ACT_LEAKY);
    // the stride-2 forms keep ONE register set of (scale, shift) for the side that carries a deferred BatchNorm
    if (S == 2 && p.aff_p.sc && p.aff_q.sc) return JVAE_ENOTSUP;
    if constexpr (NPL == 1) {
        if (leaky) return JVAE_ENOTSUP;            // the bf16 one-plane form has ReLU only (cvae.set_compute_dtype refuses 'leaky')
        static bool attr1 = false;
        if (!attr1) {
            const void* fns[2] = {reinterpret_cast<const void*>(&conv5_wgrad_x3_kernel<S, WS, MODE, 0, true, 1>),
                                  reinterpret_cast<const void*>(&conv5_wgrad_x3_kernel<S, WS, MODE, 1, true, 1>)};
            for (const void* f : fns) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
                if (e != hipSuccess) return (int)e;
            }
            attr1 = true;
        }
        if (aff) hipLaunchKernelGGL((conv5_wgrad_x3_kernel<S, WS, MODE, 1, true, 1>), grid, dim3(256), G::LDS_BYTES, st, p);
        else hipLaunchKernelGGL((conv5_wgrad_x3_kernel<S, WS, MODE, 0, true, 1>), grid, dim3(256), G::LDS_BYTES, st, p);
        JVAE_LAUNCH_CHECK();
        return 0;
    } else {
    // (the 32x32x16 MFMA shape of this kernel and its JVAE_WGRAD_SH16 switch left the tree in round 5: the 16x16x32 shape has been
    // the default since round 2 and the other was never part of a test)
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[3] = {reinterpret_cast<const void*>(&conv5_wgrad_x3_kernel<S, WS, MODE, 0, true>),
                              reinterpret_cast<const void*>(&conv5_wgrad_x3_kernel<S, WS, MODE, 1, true>),
                              reinterpret_cast<const void*>(&conv5_wgrad_x3_kernel<S, WS, MODE, 2, true>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    if (leaky) hipLaunchKernelGGL((conv5_wgrad_x3_kernel<S, WS, MODE, 2, true>), grid, dim3(256), G::LDS_BYTES, st, p);
    else if (aff) hipLaunchKernelGGL((conv5_wgrad_x3_kernel<S, WS, MODE, 1, true>), grid, dim3(256), G::LDS_BYTES, st, p);
    else hipLaunchKernelGGL((conv5_wgrad_x3_kernel<S, WS, MODE, 0, true>), grid, dim3(256), G::LDS_BYTES, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
    }
}

// column tiles by the width of the unfolded side: <= 4 channels (the 3-channel image layers): 4 channels x 8 taps (MODE 3; its
// predecessor - MODE 1 for these layers too - and the JVAE_WGRAD_M3 switch left the tree in round 5), <= 8: 8 channels x 4 taps
inline int x3_mode(int S, int Cb) { return Cb <= 4 ? 3 : (Cb <= 8 ? 1 : (S == 2 ? 2 : 0)); }

int slab_count_x3(int N, int Ca, int Cb, int S) {
    const int mode = x3_mode(S, Cb);
    const int gz = (mode == 1 || mode == 3) ? 1 : (mode == 2 ? (Cb + 15) / 16 : (Cb + 31) / 32);
    const int per = ((Ca + 31) / 32) * gz;
    int target = 512 / per;                       // ~512 workgroups in total (2 per CU), equal image counts per slab
    if (target < 1) target = 1;
    int imgs = (N + target - 1) / target;
    if (imgs < 1) imgs = 1;
    return (N + imgs - 1) / imgs;
}

}  // namespace

bool jvae_conv5_wgrad_x3_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P) {
    if (!jvae_conv5_x3_enabled()) return false;           // jvae_conv2d_set_split(0) / JVAE_X3=0: the fp32 matrix-core kernels
    if (S != 1 && S != 2) return false;
    if (HS != WS || HB != WB || WB != WS * S) return false;
    if (WS != 8 && WS != 16 && WS != 32) return false;
    if (P < 0 || P > 4 || Ca < 16 || Cb < 1) return false;
    if (P + WB > (WS - 1) * S + 5) return false;               // the patch row holds P halo columns + the image row
    return true;
}

size_t jvae_conv5_wgrad_x3_ws_floats(int N, int Ca, int Cb, int S) { return (size_t)slab_count_x3(N, Ca, Cb, S) * Ca * Cb * 25; }

// dW (+)= ...; ps / q: fp32 NCHW; swapflip: the caller passed the role-swapped problem (dst = (b*Ca + a)*25 + 24 - tap)
int jvae_conv5_wgrad_x3(const float* ps, const float* q, float* dw, int accumulate, int swapflip,
                        int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                        const InAff* aff_p, const InAff* aff_q) {
    const InAff none{nullptr, nullptr, 0};
    WgX3P p{ps, q, ws, N, Ca, Cb, P, slab_count_x3(N, Ca, Cb, S), aff_p ? *aff_p : none, aff_q ? *aff_q : none};
    int rc = JVAE_ENOTSUP;
    const int mode = x3_mode(S, Cb);
#define WGX3_CASE(S_, WS_, M_) case WS_: rc = launch_wgx3<S_, WS_, M_>(p, st); break;
    if (mode == 3 && S == 1) {
        switch (WS) { WGX3_CASE(1, 8, 3) WGX3_CASE(1, 16, 3) WGX3_CASE(1, 32, 3) }
    } else if (mode == 3) {
        switch (WS) { WGX3_CASE(2, 8, 3) WGX3_CASE(2, 16, 3) WGX3_CASE(2, 32, 3) }
    } else if (mode == 1 && S == 1) {
        switch (WS) { WGX3_CASE(1, 8, 1) WGX3_CASE(1, 16, 1) WGX3_CASE(1, 32, 1) }
    } else if (mode == 1) {
        switch (WS) { WGX3_CASE(2, 8, 1) WGX3_CASE(2, 16, 1) WGX3_CASE(2, 32, 1) }
    } else if (mode == 0) {
        switch (WS) { WGX3_CASE(1, 8, 0) WGX3_CASE(1, 16, 0) WGX3_CASE(1, 32, 0) }
    } else {
        switch (WS) { WGX3_CASE(2, 8, 2) WGX3_CASE(2, 16, 2) WGX3_CASE(2, 32, 2) }
    }
#undef WGX3_CASE
    if (rc) return rc;
    return jvae_wgrad_slab_reduce(ws, dw, p.G, Ca, Cb, accumulate, swapflip, st, 1);
}

// ---- the one-plane form for bf16 "B8" operands (conv_wgrad_b8.hip dispatches here; its own older kernel takes what this one refuses)
bool jvae_conv5_wgrad_b8x_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P) {
    if (S != 1 && S != 2) return false;
    if (HS != WS || HB != WB || WB != WS * S) return false;
    if (WS != 8 && WS != 16 && WS != 32 && !(WS == 64 && S == 1)) return false;
    if (P < 0 || P > 4 || Ca < 16 || Cb < 1) return false;
    if (P + WB > (WS - 1) * S + 5) return false;
    return true;
}

size_t jvae_conv5_wgrad_b8x_ws_floats(int N, int Ca, int Cb, int S) { return (size_t)slab_count_x3(N, Ca, Cb, S) * Ca * Cb * 25; }

int jvae_conv5_wgrad_b8x(const void* ps, const void* q, float* dw, int accumulate, int swapflip,
                         int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                         const InAff* aff_p, const InAff* aff_q) {
    const InAff none{nullptr, nullptr, 0};
    WgX3P p{ps, q, ws, N, Ca, Cb, P, slab_count_x3(N, Ca, Cb, S), aff_p ? *aff_p : none, aff_q ? *aff_q : none};
    int rc = JVAE_ENOTSUP;
    const int mode = x3_mode(S, Cb);
#define WGB8X_CASE(S_, WS_, M_) case WS_: rc = launch_wgx3<S_, WS_, M_, 1>(p, st); break;
    if (mode == 3 && S == 1) {
        switch (WS) { WGB8X_CASE(1, 8, 3) WGB8X_CASE(1, 16, 3) WGB8X_CASE(1, 32, 3) WGB8X_CASE(1, 64, 3) }
    } else if (mode == 3) {
        switch (WS) { WGB8X_CASE(2, 8, 3) WGB8X_CASE(2, 16, 3) WGB8X_CASE(2, 32, 3) }
    } else if (mode == 1 && S == 1) {
        switch (WS) { WGB8X_CASE(1, 8, 1) WGB8X_CASE(1, 16, 1) WGB8X_CASE(1, 32, 1) WGB8X_CASE(1, 64, 1) }
    } else if (mode == 1) {
        switch (WS) { WGB8X_CASE(2, 8, 1) WGB8X_CASE(2, 16, 1) WGB8X_CASE(2, 32, 1) }
    } else if (mode == 0) {
        switch (WS) { WGB8X_CASE(1, 8, 0) WGB8X_CASE(1, 16, 0) WGB8X_CASE(1, 32, 0) WGB8X_CASE(1, 64, 0) }
    } else {
        switch (WS) { WGB8X_CASE(2, 8, 2) WGB8X_CASE(2, 16, 2) WGB8X_CASE(2, 32, 2) }
    }
#undef WGB8X_CASE
    if (rc) return rc;
    return jvae_wgrad_slab_reduce(ws, dw, p.G, Ca, Cb, accumulate, swapflip, st, 1);
}
