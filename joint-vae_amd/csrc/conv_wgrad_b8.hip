// Weight gradient of the 5x5 (transposed) convolutions from bf16 "B8" activations (layout: conv_b8.hip), fp32 result.
//
//   dW[a][b][kh][kw] = sum_{n,u,v} Ps[n][a][u][v] * Q[n][b][u*S + kh - P][v*S + kw - P]
//
// (Ps = tensor on the folded "small" grid, Q = the unfolded "big" one; roles per layer type as in conv_wgrad_mfma.hip.)
//
// The contraction runs over PIXELS, but B8 keeps the channels of a pixel contiguous, i.e. both MFMA operands are stored
// k-major.  gfx950's ds_read_b64_tr_b16 is made for exactly this: per 16 lanes it reads 4 rows (pixels) x 16 columns
// (channels; each lane supplies the address of 4 consecutive channels of one pixel) and returns them column-major, so a
// lane receives 4 consecutive pixels of ITS channel: two such reads are one operand of v_mfma_f32_32x32x16_bf16, and a
// tap shift is a plain 16-byte-aligned unit offset.
//
// A workgroup owns 32 channels `a` x one column group of (b, tap) and loops over its share of the images in tiles of
// TPIX pixels (full rows); its 4 waves split the column tiles of 32:
//   MODE 0 (Cb >= 9):  tile = 32 channels b x 1 tap   -> 25 tiles per 32 channels (wave w: taps w, w+4, ...)
//   MODE 1 (Cb <= 8):  tile = 8 channels b x 4 taps   ->  7 tiles (the 3-channel image side of the first / last layer)
// The A fragment (2 transposed reads per 16 pixels) is shared by all tiles of a wave; every MFMA needs its own B
// fragment (2 transposed reads).  Accumulators stay in registers over the whole image loop; each workgroup writes one
// fp32 slab, reduced in a fixed order by jvae_wgrad_slab_reduce (deterministic; shared with the fp32 path).
#include "common.h"
#include "jvae_internal.h"
#include "conv_b8.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct WgB8P {
    const u32x4* ps;     // B8 units (N, CBa, HS, WS)
    const u32x4* q;      // B8 units (N, CBb, HB, WB)
    float* slab;         // (G, Ca, Cb*25)
    int N, Ca, Cb, CBa, CBb, P, G;
    InAff aff_p, aff_q;  // deferred BatchNorm(+ReLU) of ps / q (whichever is the layer input); CBa*8 / CBb*8 coefficients
};

template <int S, int WS, int MODE>
struct WgB8Geom {
    static constexpr int HS = WS;
    static constexpr int TPIX = (S == 1 && WS >= 16) ? 128 : 64;
    static constexpr int TH = TPIX / WS;
    static constexpr int TILES = HS * WS / TPIX;
    static constexpr int ROWS = (TH - 1) * S + 5;
    static constexpr int WB = WS * S;
    static constexpr int WP0 = (WS - 1) * S + 9, WP1 = WB + 4;
    static constexpr int WP = WP0 > WP1 ? WP0 : WP1;
    static constexpr int CH = ROWS * WP;                       // units per channel block
    static constexpr int NCBQ = MODE == 0 ? 4 : 1;             // channel blocks of Q staged per item
    static constexpr int QS = NCBQ * CH;
    static constexpr int PS = 4 * TPIX;
    static constexpr int NTILE = MODE == 0 ? 25 : 7;
    static constexpr int NBT = (NTILE + 3) / 4;                // per wave
    static constexpr int LDS_BYTES = (QS + PS) * 16;
};

// HI = byte distance of 4 pixels (64 in a dense tile, 64*S in the strided patch)
template <int HI>
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* base, int off) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + HI));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int S, int WS, int MODE, bool AFF>
__global__ __launch_bounds__(256, 2) void conv5_wgrad_b8_kernel(WgB8P p) {
    using G = WgB8Geom<S, WS, MODE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Qs = reinterpret_cast<u32x4*>(lds_raw);
    u32x4* Pt = Qs + G::QS;
    const unsigned char* Qb = lds_raw;
    const unsigned char* Pb = lds_raw + G::QS * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int a0 = blockIdx.y * 32;
    const int cbq0 = blockIdx.z * 4;                           // first channel block of Q (MODE 0)

    for (int i = tid; i < G::QS; i += 256) Qs[i] = u32x4{0u, 0u, 0u, 0u};
    __shared__ __attribute__((aligned(16))) float ctab[AFF ? 2 * 8 * (G::NCBQ + 4) : 4];     // (scale, shift) of this workgroup's q / ps channel blocks
    if (AFF && tid < 8 * (G::NCBQ + 4)) {
        constexpr int NQ = 8 * G::NCBQ;
        const bool isq = tid < NQ;
        const InAff& a = isq ? p.aff_q : p.aff_p;
        const int ch = isq ? cbq0 * 8 + tid : a0 + (tid - NQ);
        const bool ok = a.sc && ch < (isq ? p.CBb : p.CBa) * 8;
        ctab[tid] = ok ? a.sc[ch] : 0.f;
        ctab[8 * (G::NCBQ + 4) + tid] = ok ? a.sh[ch] : 0.f;
    }

    // transposed-read roles of this lane: row (pixel) q of the 4x16 block, channel quad pp; cg = 16-column group
    const int cg = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int pl = 8 * half + q4;                              // pixel of this lane inside a 16-pixel K step
    const int aoff = ((cg * 2 + (pp >> 1)) * G::TPIX + pl) * 16 + (pp & 1) * 8;
    const int lane_pix = (pl / WS) * S * G::WP + (pl % WS) * S;          // WS = 8: the K step spans two rows
    static_assert(WS >= 8, "a lane's 8 pixels must lie in one row");

    int boff[G::NBT];
#pragma unroll
    for (int t = 0; t < G::NBT; ++t) {
        const int tile = wave + 4 * t;
        int tap, cbl, sub;
        if (MODE == 0) { tap = tile; cbl = cg * 2 + (pp >> 1); sub = pp & 1; }
        else { tap = tile * 4 + cg * 2 + (pp >> 1); cbl = 0; sub = pp & 1; }
        if (tap > 24) tap = 24;                                // unused slots: any valid address
        boff[t] = (cbl * G::CH + lane_pix + (tap / 5) * G::WP + (tap % 5) + 4 - p.P) * 16 + sub * 8;
    }

    f32x16 acc[G::NBT];
#pragma unroll
    for (int t = 0; t < G::NBT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const ImageRange ir = image_range(p.N, p.G, blockIdx.x);   // trailing workgroups may be empty
    const int n_beg = ir.nb, n_end = ir.ne;
    constexpr int QUNITS = G::NCBQ * G::ROWS * G::WB;
    constexpr int PUNITS = G::PS;
    constexpr int QU = (QUNITS + 255) / 256, PU = (PUNITS + 255) / 256;
    constexpr int HB = G::HS * S;
    u32x4 rq[QU], rp[PU];
    auto gload = [&](int item) {
        const int n = item / G::TILES, tile = item % G::TILES;
        const int row0 = tile * G::TH;
        const int in_row0 = row0 * S - p.P;
#pragma unroll
        for (int k = 0; k < QU; ++k) {
            const int u = tid + k * 256;
            const int x = u % G::WB;
            const int t = u / G::WB;
            const int lr = t % G::ROWS, c = t / G::ROWS;
            const int ir = in_row0 + lr, cb = cbq0 + c;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (u < QUNITS && ir >= 0 && ir < HB && cb < p.CBb)
                v = p.q[(((long)n * p.CBb + cb) * HB + ir) * G::WB + x];
            rq[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int u = tid + k * 256;
            const int px = u % G::TPIX, c = u / G::TPIX;
            const int cb = (a0 >> 3) + c;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (u < PUNITS && cb < p.CBa)
                v = p.ps[(((long)n * p.CBa + cb) * G::HS + row0) * WS + px];
            rp[k] = v;
        }
    };
    auto lstore = [&](int item) {                  // item: the work item whose data sits in rq / rp
        constexpr int NT8 = 8 * (G::NCBQ + 4);
        const int in_row0 = (item % G::TILES) * G::TH * S - p.P;
#pragma unroll
        for (int k = 0; k < QU; ++k) {
            const int u = tid + k * 256;
            if (u < QUNITS) {
                const int x = u % G::WB;
                const int t = u / G::WB;
                const int lr = t % G::ROWS, c = t / G::ROWS;
                u32x4 v = rq[k];
                if (AFF && p.aff_q.sc) {           // padding rows / missing channel blocks stay exact zeros
                    const int ir = in_row0 + lr;
                    if (ir >= 0 && ir < HB && cbq0 + c < p.CBb) v = aff8(v, &ctab[c * 8], &ctab[NT8 + c * 8], p.aff_q.relu);
                }
                Qs[c * G::CH + lr * G::WP + 4 + x] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int u = tid + k * 256;
            if (u < PUNITS) {
                u32x4 v = rp[k];
                if (AFF && p.aff_p.sc) {
                    const int c = u / G::TPIX;
                    if ((a0 >> 3) + c < p.CBa) v = aff8(v, &ctab[(G::NCBQ + c) * 8], &ctab[NT8 + (G::NCBQ + c) * 8], p.aff_p.relu);
                }
                Pt[u] = v;
            }
        }
    };

    const int item_beg = n_beg * G::TILES, item_end = n_end * G::TILES;
    if (item_beg < item_end) gload(item_beg);
    for (int item = item_beg; item < item_end; ++item) {
        __syncthreads();
        lstore(item);
        __syncthreads();
        if (item + 1 < item_end) gload(item + 1);
#pragma unroll
        for (int ks = 0; ks < G::TPIX / 16; ++ks) {
            const int pix0 = ks * 16;
            const int qoff = ((pix0 / WS) * S * G::WP + (pix0 % WS) * S) * 16;     // compile-time after unrolling
            const bf16x8 a = tr_pair<64>(Pb, aoff + pix0 * 16);
#pragma unroll
            for (int t = 0; t < G::NBT; ++t) {
                if (wave + 4 * t < G::NTILE) {                                     // wave-uniform
                    const bf16x8 b = tr_pair<64 * S>(Qb, boff[t] + qoff);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // slab[g][a][b*25 + tap]: lane holds column l31, rows a = (r&3) + 8*(r>>2) + 4*half
    float* slab = p.slab + (long)blockIdx.x * p.Ca * (p.Cb * 25);
#pragma unroll
    for (int t = 0; t < G::NBT; ++t) {
        const int tile = wave + 4 * t;
        if (tile >= G::NTILE) continue;
        int b, tap;
        if (MODE == 0) { b = cbq0 * 8 + l31; tap = tile; }
        else { b = l31 & 7; tap = tile * 4 + (l31 >> 3); }
        if (b >= p.Cb || tap > 24) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int a = a0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (a < p.Ca) slab[(long)a * (p.Cb * 25) + b * 25 + tap] = acc[t][r];
        }
    }
}

template <int S, int WS, int MODE>
int launch_wgb8(const WgB8P& p, hipStream_t st) {
    using G = WgB8Geom<S, WS, MODE>;
    static_assert(G::LDS_BYTES <= 64 * 1024, "LDS budget");
    dim3 grid(p.G, (p.Ca + 31) / 32, MODE == 0 ? (p.Cb + 31) / 32 : 1);
    if (p.aff_p.sc || p.aff_q.sc) hipLaunchKernelGGL((conv5_wgrad_b8_kernel<S, WS, MODE, true>), grid, dim3(256), G::LDS_BYTES, st, p);
    else hipLaunchKernelGGL((conv5_wgrad_b8_kernel<S, WS, MODE, false>), grid, dim3(256), G::LDS_BYTES, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int slab_count(int N, int Ca, int Cb) {
    const int per = ((Ca + 31) / 32) * (Cb <= 8 ? 1 : (Cb + 31) / 32);
    int target = 512 / per;
    if (target < 1) target = 1;
    int imgs = (N + target - 1) / target;
    if (imgs < 1) imgs = 1;
    return (N + imgs - 1) / imgs;
}

}  // namespace

bool jvae_conv5_wgrad_b8_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P) {
    if (S != 1 && S != 2) return false;
    if (HS != WS || HB != WB || WB != WS * S) return false;
    if (WS != 8 && WS != 16 && WS != 32 && WS != 64) return false;
    if (S == 2 && WS == 64) return false;
    if (P < 0 || P > 4 || Ca < 1 || Cb < 1) return false;
    return true;
}

size_t jvae_conv5_wgrad_b8_ws_floats(int N, int Ca, int Cb) {
    const size_t a = (size_t)slab_count(N, Ca, Cb) * Ca * Cb * 25;
    const size_t b1 = jvae_conv5_wgrad_b8x_ws_floats(N, Ca, Cb, 1), b2 = jvae_conv5_wgrad_b8x_ws_floats(N, Ca, Cb, 2);
    const size_t b = b1 > b2 ? b1 : b2;
    return a > b ? a : b;
}

// dW (+)= ...; ps / q: B8 tensors; swapflip: the caller passed the role-swapped problem (dst = (b*Ca + a)*25 + 24 - tap)
int jvae_conv5_wgrad_b8(const void* ps, const void* q, float* dw, int accumulate, int swapflip,
                        int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                        const InAff* aff_p, const InAff* aff_q) {
    // the LDS image and read-ahead pipeline of the split-bf16 kernel, one plane (conv_wgrad_x3.hip)
    if (jvae_conv5_wgrad_b8x_ok(Ca, WS, WS, Cb, WS * S, WS * S, S, P))
        return jvae_conv5_wgrad_b8x(ps, q, dw, accumulate, swapflip, N, Ca, WS, Cb, S, P, ws, st, aff_p, aff_q);
    const InAff none{nullptr, nullptr, 0};
    WgB8P p{(const u32x4*)ps, (const u32x4*)q, ws, N, Ca, Cb, (Ca + 7) / 8, (Cb + 7) / 8, P, slab_count(N, Ca, Cb),
            aff_p ? *aff_p : none, aff_q ? *aff_q : none};
    int rc = JVAE_ENOTSUP;
#define WG_CASE(S_, WS_)                                                              \
    case WS_: rc = Cb <= 8 ? launch_wgb8<S_, WS_, 1>(p, st) : launch_wgb8<S_, WS_, 0>(p, st); break;
    if (S == 1) {
        switch (WS) { WG_CASE(1, 8) WG_CASE(1, 16) WG_CASE(1, 32) WG_CASE(1, 64) }
    } else {
        switch (WS) { WG_CASE(2, 8) WG_CASE(2, 16) WG_CASE(2, 32) }
    }
#undef WG_CASE
    if (rc) return rc;
    return jvae_wgrad_slab_reduce(ws, dw, p.G, Ca, Cb, accumulate, swapflip, st);
}
