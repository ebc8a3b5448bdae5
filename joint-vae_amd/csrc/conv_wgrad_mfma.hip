// Weight gradient of the 5x5 (transposed) convolutions on the fp32 matrix cores.
//
//   dW[a][b][kh][kw] = sum_{n,u,v} Ps[n][a][u][v] * Q[n][b][u*S + kh - P][v*S + kw - P]
//
// Ps = the tensor on the folded ("small") grid, Q = the unfolded ("big") one:
//   Conv2d          : a = cout, Ps = dy ; b = cin,  Q = x    -> dW in the layer's own layout (Cout,Cin,5,5)
//   ConvTranspose2d : a = cin,  Ps = x  ; b = cout, Q = dy   -> (Cin,Cout,5,5), again the layer's layout.
// (Reference: autograd of nn.Conv2d / nn.ConvTranspose2d, module/vae_layers/conv.py:186-196.)
//
// MFMA mapping (v_mfma_f32_32x32x2_f32): rows i = 32 channels `a`, columns j = 32 consecutive (b, tap)
// pairs, K = 2 pixels per instruction.  A workgroup owns one 32-row a-tile x CB channels of b (CB*25 columns,
// split over its 4 waves) and loops over its share of the images in tiles of TPIX pixels (full rows); the
// 25 taps of a column differ only by a constant LDS offset into the staged Q patch (with zero halo), so the
// B fragment read is one ds_read_b32 at lane_offset + immediate.  The Ps tile is stored with an odd row
// pitch (conflict-free A reads).  Accumulators live in registers across the whole image loop; each
// workgroup writes one slab, and a second kernel reduces the slabs in a fixed order (deterministic, no
// float atomics) into dW (optionally accumulating, optionally transposing / flipping for the swapped form).
#include <stdlib.h>
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"

namespace {

struct WgP {
    const float* ps;     // (N, Ca, HS, WS)
    const float* q;      // (N, Cb, HB, WB)
    float* slab;         // (G, Ca, Cb*25)
    int N, Ca, Cb, P, G;
    InAff aff_p, aff_q;  // deferred BatchNorm(+ReLU) of ps / q (whichever is the layer's input; sc == nullptr: none)
};

template <int S, int WS, int CB>
struct WgGeom {
    static constexpr int HS = WS;
    static constexpr int TPIX = (WS >= 16 && !(S == 2 && WS >= 32)) ? 128 : 64;   // pixels per tile (full rows)
    static constexpr int TH = TPIX / WS;
    static constexpr int TILES = HS * WS / TPIX;              // tiles per image
    static constexpr int ROWS = (TH - 1) * S + 5;
    static constexpr int WB = WS * S;
    static constexpr int WP0 = (WS - 1) * S + 9, WP1 = WB + 4;
    static constexpr int WP = (((WP0 > WP1 ? WP0 : WP1) + 3) / 4) * 4;
    static constexpr int CH = ROWS * WP;
    static constexpr int QS = CB * CH;
    static constexpr int PPITCH = TPIX + 1;
    static constexpr int PSZ = 32 * PPITCH;
    static constexpr int COLS = CB * 25;
    static constexpr int NTILE = (COLS + 31) / 32;            // 32-column tiles of the workgroup
    static constexpr int NBT = (NTILE + 3) / 4;               // per wave
};

template <int S, int WS, int CB, bool PF, int AFF>      // AFF: 0 plain operands, 1 deferred BatchNorm (+ReLU by flag), 2 ... with a leaky ReLU on a side
__global__ __launch_bounds__(256, 2) void conv5_wgrad_kernel(WgP p) {
    using G = WgGeom<S, WS, CB>;
    __shared__ __attribute__((aligned(16))) float lds[G::QS + G::PSZ];
    __shared__ float coef[AFF ? 2 * (CB + 32) : 1];           // deferred-BatchNorm (scale, shift) of this workgroup's channels
    float* Qs = lds;
    float* Pt = lds + G::QS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int a0 = blockIdx.y * 32;
    const int b0 = blockIdx.z * CB;
    const int cb_here = min(CB, p.Cb - b0);                   // channels of b actually present
    const int cols_here = cb_here * 25;

    for (int i = tid; i < G::QS / 4; i += 256) reinterpret_cast<f32x4*>(Qs)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (AFF && tid < CB + 32) {                                // identity where no transform was given
        const bool isq = tid < CB;
        const InAff& a = isq ? p.aff_q : p.aff_p;
        const int ch = isq ? b0 + tid : a0 + (tid - CB);
        const bool ok = a.sc && ch < (isq ? p.Cb : p.Ca);
        coef[2 * tid] = ok ? a.sc[ch] : 1.f;
        coef[2 * tid + 1] = ok ? a.sh[ch] : 0.f;
    }

    // this lane's columns: LDS offset of (b_local, tap) relative to a pixel's patch origin
    int boff[G::NBT];
    bool bvalid[G::NBT];
#pragma unroll
    for (int t = 0; t < G::NBT; ++t) {
        const int col = (wave * G::NBT + t) * 32 + l31;
        bvalid[t] = col < cols_here;
        const int cc = bvalid[t] ? col : 0;
        const int bl = cc / 25, tap = cc % 25;
        boff[t] = bl * G::CH + (tap / 5) * G::WP + (tap % 5) + 4 - p.P + half * S;
    }
    const int aoff = l31 * G::PPITCH + half;

    f32x16 acc[G::NBT];
#pragma unroll
    for (int t = 0; t < G::NBT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // images of this workgroup
    const ImageRange ir = image_range(p.N, p.G, blockIdx.x);   // trailing workgroups may be empty
    const int n_beg = ir.nb, n_end = ir.ne;
    constexpr int W4 = G::WB / 4;
    constexpr int QUNITS = CB * G::ROWS * W4;
    constexpr int PUNITS = 32 * G::TPIX / 4;
    const int HB = G::HS * S;

    // (image, row-tile) work items of this workgroup, software-pipelined: the global loads of item i+1 are in
    // flight (registers) under the MFMAs of item i.
    constexpr int QU = (QUNITS + 255) / 256, PU = (PUNITS + 255) / 256;
    f32x4 rq[QU], rp[PU];
    auto gload = [&](int item) {
        const int n = item / G::TILES, tile = item % G::TILES;
        const int row0 = tile * G::TH;
        const int in_row0 = row0 * S - p.P;
#pragma unroll
        for (int k = 0; k < QU; ++k) {
            const int u = tid + k * 256;
            const int x4 = u % W4;
            const int t = u / W4;
            const int lr = t % G::ROWS, c = t / G::ROWS;
            const int ir = in_row0 + lr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < QUNITS && ir >= 0 && ir < HB && c < cb_here)
                v = *reinterpret_cast<const f32x4*>(p.q + (((long)n * p.Cb + b0 + c) * HB + ir) * G::WB + x4 * 4);
            rq[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int u = tid + k * 256;
            const int p4 = u % (G::TPIX / 4), a = u / (G::TPIX / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < PUNITS && a0 + a < p.Ca)
                v = *reinterpret_cast<const f32x4*>(p.ps + (((long)n * p.Ca + a0 + a) * G::HS + row0) * WS + p4 * 4);
            rp[k] = v;
        }
    };
    // `item`: the work item whose data sits in rq / rp (needed to tell padding rows, which must stay zero, from data)
    auto lstore = [&](int item) {
        const int in_row0 = (item % G::TILES) * G::TH * S - p.P;
#pragma unroll
        for (int k = 0; k < QU; ++k) {
            const int u = tid + k * 256;
            if (u < QUNITS) {
                const int x4 = u % W4;
                const int t = u / W4;
                const int lr = t % G::ROWS, c = t / G::ROWS;
                f32x4 v = rq[k];
                if (AFF && p.aff_q.sc) {
                    const int ir = in_row0 + lr;
                    if (ir >= 0 && ir < HB && c < cb_here) v = AFF == 2 ? aff4_kind(v, coef[2 * c], coef[2 * c + 1], p.aff_q.relu) : aff4(v, coef[2 * c], coef[2 * c + 1], p.aff_q.relu);
                }
                *reinterpret_cast<f32x4*>(&Qs[c * G::CH + lr * G::WP + 4 + x4 * 4]) = v;
            }
        }
#pragma unroll
        for (int k = 0; k < PU; ++k) {
            const int u = tid + k * 256;
            if (u < PUNITS) {
                const int p4 = u % (G::TPIX / 4), a = u / (G::TPIX / 4);
                f32x4 v = rp[k];
                if (AFF && p.aff_p.sc && a0 + a < p.Ca) v = AFF == 2 ? aff4_kind(v, coef[2 * (CB + a)], coef[2 * (CB + a) + 1], p.aff_p.relu) : aff4(v, coef[2 * (CB + a)], coef[2 * (CB + a) + 1], p.aff_p.relu);
                float* d = &Pt[a * G::PPITCH + p4 * 4];
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        }
    };

    const int item_beg = n_beg * G::TILES, item_end = n_end * G::TILES;
    if (PF && item_beg < item_end) gload(item_beg);
    for (int item = item_beg; item < item_end; ++item) {
        __syncthreads();
        if (!PF) gload(item);                    // wide (CB = 32) variant: no registers to spare for a prefetch
        lstore(item);
        __syncthreads();
        if (PF && item + 1 < item_end) gload(item + 1);
#pragma unroll
        for (int ks = 0; ks < G::TPIX / 2; ++ks) {
            const int pix = 2 * ks;                                   // + half (folded into aoff / boff)
            const int qoff = (pix / WS) * S * G::WP + (pix % WS) * S; // compile-time after unrolling
            const float a = Pt[aoff + pix];
#pragma unroll
            for (int t = 0; t < G::NBT; ++t) {
                if ((wave * G::NBT + t) < G::NTILE) {                 // wave-uniform
                    const float b = Qs[boff[t] + qoff];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // slab[g][a][b*25 + tap]: lane holds column (b,tap), rows a = (r&3) + 8*(r>>2) + 4*half
    float* slab = p.slab + (long)blockIdx.x * p.Ca * (p.Cb * 25);
#pragma unroll
    for (int t = 0; t < G::NBT; ++t) {
        if (!bvalid[t]) continue;
        const int col = (wave * G::NBT + t) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int a = a0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (a < p.Ca) slab[(long)a * (p.Cb * 25) + b0 * 25 + col] = acc[t][r];
        }
    }
}

// dw[dst(i)] (+)= sum_g slab[g][i] in a fixed order (4 interleaved partial sums per output, then LDS).
// swapflip: slab is (a, b, tap') of the role-swapped problem -> dst = (b*Ca + a)*25 + 24 - tap'.
// tapmajor: the slab is (a, tap, b) instead of (a, b, tap) (conv_wgrad_x3.hip: lanes = channels b, coalesced slab stores).
__device__ __forceinline__ int wgrad_dst(int i, int Ca, int Cb, int swapflip, int tapmajor) {
    if (!swapflip && !tapmajor) return i;
    int tap, b;
    const int a = i / (25 * Cb);
    if (tapmajor) { b = i % Cb; tap = (i / Cb) % 25; }
    else { tap = i % 25; b = (i / 25) % Cb; }
    return swapflip ? (b * Ca + a) * 25 + 24 - tap : (a * Cb + b) * 25 + tap;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           int G, int Ca, int Cb, int accumulate, int swapflip, int tapmajor) {
    __shared__ float part[4][64];
    const int total = Ca * Cb * 25;
    const int ix = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + ix;
    float s = 0.f;
    if (i < total)
        for (int g = gy; g < G; g += 4) s += slab[(long)g * total + i];
    part[gy][ix] = s;
    __syncthreads();
    if (gy != 0 || i >= total) return;
    s = (part[0][ix] + part[1][ix]) + (part[2][ix] + part[3][ix]);
    const int dst = wgrad_dst(i, Ca, Cb, swapflip, tapmajor);
    dw[dst] = accumulate ? dw[dst] + s : s;
}

// The same fold with 16-byte accesses (Ca*Cb*25 a multiple of 4): 16 outputs-of-4 x 16 slab lanes per workgroup, every
// thread adds its G/16 slabs in order, the 16 partial sums are folded in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const f32x4* __restrict__ slab, float* __restrict__ dw,
                                                            int G, int Ca, int Cb, int accumulate, int swapflip, int tapmajor) {
    __shared__ f32x4 part[16][16];
    const int total4 = Ca * Cb * 25 / 4;
    const int ix = threadIdx.x & 15, gy = threadIdx.x >> 4;
    const int i4 = blockIdx.x * 16 + ix;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i4 < total4)
        for (int g = gy; g < G; g += 16) s += slab[(long)g * total4 + i4];
    part[gy][ix] = s;
    __syncthreads();
    if (gy != 0 || i4 >= total4) return;
    f32x4 t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (part[4 * k][ix] + part[4 * k + 1][ix]) + (part[4 * k + 2][ix] + part[4 * k + 3][ix]);
    s = (t[0] + t[1]) + (t[2] + t[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int dst = wgrad_dst(i4 * 4 + e, Ca, Cb, swapflip, tapmajor);
        dw[dst] = accumulate ? dw[dst] + s[e] : s[e];
    }
}

template <int S, int WS, int CB, bool PF = true>
int launch_wg(const WgP& p, hipStream_t st) {
    using G = WgGeom<S, WS, CB>;
    static_assert((G::QS + G::PSZ) * 4 + 8 * (CB + 32) <= 64 * 1024, "static LDS budget");
    dim3 grid(p.G, (p.Ca + 31) / 32, (p.Cb + CB - 1) / CB);
    // two instantiations: the deferred-BatchNorm transform costs registers / LDS only where it is used
    const bool leaky = (p.aff_p.sc && p.aff_p.relu == JVAE_ACT_LEAKY) || (p.aff_q.sc && p.aff_q.relu == JVAE_ACT_LEAKY);
    if (leaky) hipLaunchKernelGGL((conv5_wgrad_kernel<S, WS, CB, PF, 2>), grid, dim3(256), 0, st, p);
    else if (p.aff_p.sc || p.aff_q.sc) hipLaunchKernelGGL((conv5_wgrad_kernel<S, WS, CB, PF, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv5_wgrad_kernel<S, WS, CB, PF, 0>), grid, dim3(256), 0, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

inline bool wide_ok(int S, int WS, int Cb) {          // the 32-channel (25 tiles / 28 slots) variant
    return S == 1 && (WS == 16 || WS == 32) && Cb % 32 == 0;
}
inline int pick_cb(int S, int WS, int Cb) {
    if (Cb <= 4) return 4;                      // 3-channel tensors (image side of the first / last layer)
    return wide_ok(S, WS, Cb) ? 32 : 16;
}

}  // namespace

// dw (+)= sum over the G slabs (a, b, tap) in a fixed order; shared with the bf16 path (conv_wgrad_b8.hip)
int jvae_wgrad_slab_reduce(const float* slab, float* dw, int G, int Ca, int Cb, int accumulate, int swapflip, hipStream_t st,
                           int tapmajor) {
    const int total = Ca * Cb * 25;
    if (total % 4 == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(cdiv(total / 4, 16)), dim3(256), 0, st, (const f32x4*)slab, dw, G, Ca, Cb,
                           accumulate, swapflip, tapmajor);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, st, slab, dw, G, Ca, Cb, accumulate, swapflip, tapmajor);
    JVAE_LAUNCH_CHECK();
    return 0;
}

bool jvae_conv5_wgrad_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P) {
    if (S != 1 && S != 2) return false;
    if (HS != WS || HB != WB || WB != WS * S) return false;
    if (WS != 8 && WS != 16 && WS != 32 && WS != 64) return false;
    if (P < 0 || P > 4 || Ca < 1 || Cb < 1) return false;
    return true;
}

static int slab_count(int N, int Ca, int Cb, int S, int WS) {
    // ~512 workgroups in total (2 per CU; fatter slabs amortise the zero-fill / slab write / reduce), every slab
    // with the same number of images (no idle tail)
    const int cb = pick_cb(S, WS, Cb);
    const int per = ((Ca + 31) / 32) * ((Cb + cb - 1) / cb);
    int target = 512 / per;
    if (target < 1) target = 1;
    int imgs = (N + target - 1) / target;         // images per slab
    if (imgs < 1) imgs = 1;
    return (N + imgs - 1) / imgs;
}

size_t jvae_conv5_wgrad_ws_floats(int N, int Ca, int Cb, int S, int WS) {
    const size_t f32 = (size_t)slab_count(N, Ca, Cb, S, WS) * Ca * Cb * 25;
    const size_t x3 = jvae_conv5_wgrad_x3_ok(Ca, WS, WS, Cb, WS * S, WS * S, S, 2) ? jvae_conv5_wgrad_x3_ws_floats(N, Ca, Cb, S) : 0;
    return f32 > x3 ? f32 : x3;
}

// dW (+)= ... ; swapflip: the caller passed the role-swapped problem (see wgrad_reduce_kernel).
int jvae_conv5_wgrad(const float* ps, const float* q, float* dw, int accumulate, int swapflip,
                     int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                     const InAff* aff_p, const InAff* aff_q) {
    if (jvae_conv5_wgrad_x3_ok(Ca, WS, WS, Cb, WS * S, WS * S, S, P))      // split-bf16 arithmetic (conv_wgrad_x3.hip)
        return jvae_conv5_wgrad_x3(ps, q, dw, accumulate, swapflip, N, Ca, WS, Cb, S, P, ws, st, aff_p, aff_q);
    const InAff none{nullptr, nullptr, 0};
    WgP p{ps, q, ws, N, Ca, Cb, P, slab_count(N, Ca, Cb, S, WS), aff_p ? *aff_p : none, aff_q ? *aff_q : none};
    int rc = JVAE_ENOTSUP;
    if (Cb <= 4) {
        if (S == 1) {
            switch (WS) {
                case 8: rc = launch_wg<1, 8, 4>(p, st); break;
                case 16: rc = launch_wg<1, 16, 4>(p, st); break;
                case 32: rc = launch_wg<1, 32, 4>(p, st); break;
                case 64: rc = launch_wg<1, 64, 4>(p, st); break;
            }
        } else {
            switch (WS) {
                case 8: rc = launch_wg<2, 8, 4>(p, st); break;
                case 16: rc = launch_wg<2, 16, 4>(p, st); break;
                case 32: rc = launch_wg<2, 32, 4>(p, st); break;
                case 64: rc = launch_wg<2, 64, 4>(p, st); break;
            }
        }
    } else if (wide_ok(S, WS, Cb)) {
        rc = WS == 16 ? launch_wg<1, 16, 32, false>(p, st) : launch_wg<1, 32, 32, false>(p, st);
    } else if (S == 1) {
        switch (WS) {
            case 8: rc = launch_wg<1, 8, 16>(p, st); break;
            case 16: rc = launch_wg<1, 16, 16>(p, st); break;
            case 32: rc = launch_wg<1, 32, 16>(p, st); break;
            case 64: rc = launch_wg<1, 64, 16>(p, st); break;
        }
    } else {
        switch (WS) {
            case 8: rc = launch_wg<2, 8, 16>(p, st); break;
            case 16: rc = launch_wg<2, 16, 16>(p, st); break;
            case 32: rc = launch_wg<2, 32, 16>(p, st); break;
            case 64: rc = launch_wg<2, 64, 16>(p, st); break;
        }
    }
    if (rc) return rc;
    return jvae_wgrad_slab_reduce(ws, dw, p.G, Ca, Cb, accumulate, swapflip, st);
}
