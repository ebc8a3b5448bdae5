// Implicit-GEMM 5x5 convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) for NCHW tensors.
//
//   out[n][o][y][x] = bias[o] + sum_{c,kh,kw} in[n][c][y*S + kh - P][x*S + kw - P] * Wp[c][kh*5+kw][o]
//
// One kernel family ("forward-type") serves, through a weight re-pack (pack_kernel below):
//   Conv2d forward (S = 1, 2), ConvTranspose2d stride-1 forward (flipped taps, P' = 4 - P),
//   Conv2d stride-1 dgrad (swapped + flipped), ConvTranspose2d dgrad (a plain convolution of dy, S = 1, 2).
// Reference ops: nn.Conv2d / nn.ConvTranspose2d 5x5 layers of conv32 / deconv32 (conv-models.ini:19,25).
//
// Mapping to CDNA4:
//  * A workgroup (256 threads = 4 waves) owns PIX = MT*128 consecutive output pixels (full image rows, one or
//    several images) x NT*32 output channels.  The input patch of those pixels (with halo) for CC input
//    channels and the CC*25*(NT*32) weight slice live in LDS; the halo / out-of-image cells are zeroed ONCE,
//    per channel chunk only the interior is re-staged with aligned 16-byte global loads (NCHW rows).
//  * MFMA operands: A = weights (rows i = output channel), B = input patch (cols j = pixel), K = 2 input
//    channels of the same tap per instruction, so both fragment reads are one ds_read_b32 with an
//    immediate offset; 32 consecutive pixels per half-wave are consecutive LDS dwords (conflict-free for
//    32-wide rows, 2-way for narrower ones).
//  * The accumulator tile has the pixel on the lane and the channel in the register index, so every store
//    instruction writes 128 contiguous bytes of one NCHW plane.
//  * Each wave: MT x NT tiles of 32x32 (64 accumulator VGPRs); per 4 MFMAs (256 matrix-pipe cycles) it
//    issues MT + NT LDS reads: the kernel is matrix-pipe bound, co-resident workgroups hide the staging.
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"
#include "pack_elems.h"

namespace {

struct FwdP {
    const float* in;     // (N, Cin, H, W)
    const float* wp;     // packed (Cin, 25, Cout)
    const float* bias;   // (Cout) or null
    float* out;          // (N, Cout, OH, OW)
    int N, Cin, H, W, Cout, P;   // Cout: padded to a multiple of 32 (packed weights), CoutReal: channels of `out`
    int CoutReal;
    float* stats;        // optional (CoutReal, gridDim.x, 2): per-workgroup sum / sum of squares of (out - bias)
    InAff aff;           // deferred BatchNorm(+ReLU) of the input (sc == nullptr: none)
};

template <int S, int OW, int MT, int NT, int CC>
struct FwdGeom {
    static constexpr int OH = OW;
    static constexpr int PIX = MT * 128;                       // output pixels per workgroup
    static constexpr int OHW = OH * OW;
    static constexpr int NIMG = PIX >= OHW ? PIX / OHW : 1;    // images per workgroup
    static constexpr int TH = PIX >= OHW ? OH : PIX / OW;      // output rows per image in the tile
    static constexpr int ROWS = (TH - 1) * S + 5;              // input rows in the patch
    static constexpr int WIN = OW * S;                         // input width
    static constexpr int WP0 = (OW - 1) * S + 9;               // worst case P = 0 (lds col = x*S + kw - P + 4)
    static constexpr int WP1 = WIN + 4;
    static constexpr int WP = (((WP0 > WP1 ? WP0 : WP1) + 3) / 4) * 4;
    static constexpr int CH = ROWS * WP;                       // floats per channel per image
    static constexpr int XS = NIMG * CC * CH;                  // input patch floats
    static constexpr int WCOLS = NT * 32;
    static constexpr int WS = CC * 25 * WCOLS;
};

template <int S, int OW, int MT, int NT, int CC, int AFF>      // AFF: 0 plain input, 1 deferred BatchNorm (+ReLU by flag), 2 ... + leaky ReLU
__global__ __launch_bounds__(256, 2) void conv5_fwd_kernel(FwdP p) {
    using G = FwdGeom<S, OW, MT, NT, CC>;
    __shared__ __attribute__((aligned(16))) float lds[G::XS + G::WS];
    float* Xs = lds;
    float* Ws = lds + G::XS;
    __shared__ float bias_s[G::WCOLS];                        // this workgroup's bias values, fetched at the start (conv_x3.hip)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int TILES_PER_IMG = G::OHW >= G::PIX ? G::OHW / G::PIX : 1;
    const int img0 = (G::OHW >= G::PIX) ? (int)(blockIdx.x / TILES_PER_IMG) : (int)blockIdx.x * G::NIMG;
    const int row0 = (G::OHW >= G::PIX) ? (int)(blockIdx.x % TILES_PER_IMG) * G::TH : 0;
    const int o0 = blockIdx.y * G::WCOLS;
    if (tid < G::WCOLS) bias_s[tid] = (p.bias && o0 + tid < p.CoutReal) ? p.bias[o0 + tid] : 0.f;

    // ---- zero the whole patch once: halo columns / out-of-image rows / missing images stay zero for all chunks
    for (int i = tid; i < G::XS / 4; i += 256) reinterpret_cast<f32x4*>(Xs)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- this lane's pixels (B operand): LDS offset of tap (0,0), channel 0
    int pixoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pix = (wave * MT + mt) * 32 + l31;          // pixel inside the workgroup tile
        const int im = pix / (G::TH * OW), rem = pix % (G::TH * OW);
        const int r = rem / OW, c = rem % OW;
        pixoff[mt] = im * (CC * G::CH) + (r * S) * G::WP + c * S + 4 - p.P + half * G::CH;
    }

    f32x16 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int in_row0 = row0 * S - p.P;                       // input row of patch row 0
    constexpr int W4 = G::WIN / 4;
    constexpr int XUNITS = G::NIMG * CC * G::ROWS * W4;       // interior float4s per chunk
    constexpr int WUNITS = G::WS / 4;

    // Software pipeline: the global loads of chunk c+1 are issued before the MFMAs of chunk c and land in
    // registers; they are written to LDS after the barrier that retires chunk c's fragment reads.
    constexpr int XU = (XUNITS + 255) / 256, WU = (WUNITS + 255) / 256;
    f32x4 rx[XU], rw[WU];
    float rsc[AFF ? XU : 1], rsh[AFF ? XU : 1];   // deferred-BatchNorm coefficients of the prefetched units (applied in
                                                  // lstore, i.e. after the MFMAs that hide the global-load latency)
    auto gload = [&](int c0) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            const int x4 = u % W4;
            int t = u / W4;
            const int lr = t % G::ROWS; t /= G::ROWS;
            const int c = t % CC, im = t / CC;
            const int ir = in_row0 + lr, n = img0 + im, ch = c0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            float sc = 0.f, sh = 0.f;                        // padding cells: 0*0 + 0 stays an exact zero
            if (u < XUNITS && ir >= 0 && ir < p.H && n < p.N && ch < p.Cin) {
                v = *reinterpret_cast<const f32x4*>(p.in + (((long)n * p.Cin + ch) * p.H + ir) * p.W + x4 * 4);
                if (AFF) { sc = p.aff.sc[ch]; sh = p.aff.sh[ch]; }
            }
            rx[k] = v;
            if (AFF) { rsc[k] = sc; rsh[k] = sh; }
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * 256;
            const int col4 = u % (G::WCOLS / 4), kr = u / (G::WCOLS / 4);     // kr = c*25 + tap
            const int ch = c0 + kr / 25;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < WUNITS && ch < p.Cin)
                v = *reinterpret_cast<const f32x4*>(p.wp + ((long)(c0 * 25 + kr)) * p.Cout + o0 + col4 * 4);
            rw[k] = v;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            if (u < XUNITS) {
                const int x4 = u % W4;
                int t = u / W4;
                const int lr = t % G::ROWS; t /= G::ROWS;
                const int c = t % CC, im = t / CC;
                *reinterpret_cast<f32x4*>(&Xs[(im * CC + c) * G::CH + lr * G::WP + 4 + x4 * 4]) =
                    AFF == 2 ? aff4_leaky(rx[k], rsc[AFF ? k : 0], rsh[AFF ? k : 0]) : (AFF ? aff4(rx[k], rsc[AFF ? k : 0], rsh[AFF ? k : 0], p.aff.relu) : rx[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * 256;
            if (u < WUNITS) reinterpret_cast<f32x4*>(Ws)[u] = rw[k];
        }
    };

    gload(0);
    for (int c0 = 0; c0 < p.Cin; c0 += CC) {
        __syncthreads();                                      // previous chunk fully consumed (and zero fill done)
        lstore();
        __syncthreads();
        if (c0 + CC < p.Cin) gload(c0 + CC);
        // operand fragments of step j+1 are read from LDS before the MFMAs of step j are issued (explicit one-deep
        // software pipeline: the LDS latency hides under MT*NT MFMAs instead of stalling in front of them)
        constexpr int STEPS = (CC / 2) * 25;
        float a[2][NT], b[2][MT];
        auto frag = [&](int j, float (&fa)[NT], float (&fb)[MT]) {
            const int cp = j / 25, tap = j % 25;
            const int kh = tap / 5, kw = tap % 5;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fa[nt] = Ws[((cp * 2 + half) * 25 + tap) * G::WCOLS + nt * 32 + l31];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fb[mt] = Xs[pixoff[mt] + (cp * 2) * G::CH + kh * G::WP + kw];
        };
        frag(0, a[0], b[0]);
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            if (j + 1 < STEPS) frag(j + 1, a[(j + 1) & 1], b[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);                 // keep the reads of step j+1 ahead of the MFMAs of step j
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][nt], b[j & 1][mt], acc[nt][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- optional BatchNorm statistics of this workgroup's tile (pixels of missing images contribute exact zeros)
    if (p.stats) {
        __syncthreads();                                      // LDS is free now
        float* red = lds;                                     // [4 waves][NT*32][2]
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

    // ---- epilogue: D[i = channel][j = pixel]; lane holds pixel j = l31, rows i = (r&3) + 8*(r>>2) + 4*half
    auto store_tile = [&](const float (&bv)[NT][16], bool biased) {
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
                for (int r = 0; r < 16; ++r) {
                    const int o = o0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (o >= p.CoutReal) continue;
                    p.out[(((long)n * p.CoutReal + o) * G::OH + oy) * OW + ox] = biased ? acc[nt][mt][r] + bv[nt][r] : acc[nt][mt][r];
                }
        }
    };
    float bv[NT][16];
    if (p.bias) {           // the lane's bias values in one batch of loads, not one dependent load in front of every store
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bv[nt][r] = bias_s[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
            }
        store_tile(bv, true);
    } else {
        store_tile(bv, false);
    }
}

// Wp[c][tap][o] (o < OP, zero for o >= O) from a PyTorch-layout weight.
// swap: source is [c][o][tap] (else [o][c][tap]); flip: tap -> 24 - tap
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                   int C, int O, int OP, int swap, int flip) {
    const int total = C * 25 * OP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        jvae_pack_f32_elem(w, wp, i, C, O, swap, flip);
}

thread_local int g_last_splits = 0;     // grid.x of the last forward-type launch (host side, per call: read right after launching)

template <int S, int OW, int MT, int NT, int CC>
int launch_fwd(const FwdP& p, hipStream_t st) {
    // two instantiations: the deferred-BatchNorm input transform costs registers only where it is used
    using G = FwdGeom<S, OW, MT, NT, CC>;
    static_assert((G::XS + G::WS) * 4 <= 160 * 1024, "LDS budget");
    const long pixels = (long)p.N * G::OHW;
    dim3 grid((unsigned)((pixels + G::PIX - 1) / G::PIX), (unsigned)(p.Cout / G::WCOLS));
    if (G::OHW < G::PIX) grid.x = (unsigned)((p.N + G::NIMG - 1) / G::NIMG);
    g_last_splits = (int)grid.x;
    if (p.aff.sc && p.aff.relu == JVAE_ACT_LEAKY) hipLaunchKernelGGL((conv5_fwd_kernel<S, OW, MT, NT, CC, 2>), grid, dim3(256), 0, st, p);
    else if (p.aff.sc) hipLaunchKernelGGL((conv5_fwd_kernel<S, OW, MT, NT, CC, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv5_fwd_kernel<S, OW, MT, NT, CC, 0>), grid, dim3(256), 0, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// Tile choice: the largest tile that still gives the chip >= ~3 workgroups per CU (256 CUs); small feature maps
// (8x8, 16x16 at batch 512) fall back to 256- or 128-pixel tiles x 32 channels.
template <int S, int OW>
int launch_fwd_ow(const FwdP& p, hipStream_t st) {
    const long pixels = (long)p.N * OW * OW;
    auto wgs = [&](int mt, int nt) { return ((pixels + mt * 128 - 1) / (mt * 128)) * (p.Cout / (nt * 32)); };
    constexpr long ENOUGH = 768;
    if (p.Cin <= 4) return launch_fwd<S, OW, 4, 1, 4>(p, st);      // 3-channel inputs: one chunk of 2 channel pairs
    if (p.Cout % 64 == 0 && wgs(2, 2) >= ENOUGH) return launch_fwd<S, OW, 2, 2, (S == 1 ? 8 : 4)>(p, st);
    if (wgs(4, 1) >= ENOUGH) return launch_fwd<S, OW, 4, 1, (S == 1 ? 8 : 4)>(p, st);
    if (wgs(2, 1) >= ENOUGH) return launch_fwd<S, OW, 2, 1, (S == 1 ? 8 : 4)>(p, st);
    return launch_fwd<S, OW, 1, 1, (S == 1 ? 8 : 4)>(p, st);
}

}  // namespace

// Is the forward-type fast kernel applicable?  in: (N,Cin,H,W) -> out: (N,Cout,OH,OW), 5x5, stride S, padding P.
bool jvae_conv5_fwd_ok(int Cin, int H, int W, int Cout, int OH, int OW, int S, int P) {
    if (S != 1 && S != 2) return false;
    if (OH != OW || H != W) return false;
    if (OW != 8 && OW != 16 && OW != 32 && OW != 64) return false;
    if (W != OW * S) return false;                 // "same"-style geometry: 5x5, P = 2 (or its transposed mirror)
    if (P < 0 || P > 4) return false;
    if ((OW - 1) * S + 4 - P >= W + 4) return false;   // right-most tap must stay inside the zero halo
    if (Cout < 1 || Cin < 1) return false;       // Cout is padded to a multiple of 32 inside
    return true;
}

// workspace of the weight re-pack in floats: the fp32 operand layout or the three bf16 planes of conv_x3.hip
size_t jvae_conv5_pack_floats(int Cin, int Cout) {
    const size_t a = (size_t)Cin * 25 * ((Cout + 31) / 32 * 32), b = (jvae_conv5_x3_pack_bytes(Cin, Cout) + 3) / 4;
    return a > b ? a : b;
}

int jvae_conv5_pack(const float* w, float* wp, int C, int O, int swap, int flip, hipStream_t st) {
    const int OP = (O + 31) / 32 * 32;
    const int total = C * 25 * OP;
    hipLaunchKernelGGL(pack_kernel, dim3(cdiv(total, 256) > 512 ? 512 : cdiv(total, 256)), dim3(256), 0, st,
                       w, wp, C, O, OP, swap, flip);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// ws must hold Cin*25*Cout floats (the packed weights).  swap / flip: see pack_kernel.
// Upper bound of the number of per-channel partials a stats-producing launch writes (smallest tile: 128 pixels).
int jvae_conv5_fwd_max_splits(int N, int OW) { return (int)(((long)N * OW * OW + 127) / 128) + 1; }

int jvae_conv5_fwd(const float* in, const float* w, int swap, int flip, const float* bias, float* out,
                   int N, int Cin, int H, int W, int Cout, int OW, int S, int P, float* ws, hipStream_t st,
                   float* stats, int* nsplit, const InAff* aff) {
    if (jvae_conv5_x3_ok(Cin, H, W, Cout, OW, OW, S, P))      // stride-1 layers with >= 16 input channels: conv_x3.hip
        return jvae_conv5_x3_fwd(in, w, swap, flip, bias, out, N, Cin, H, W, Cout, OW, S, P, ws, st, stats, nsplit, aff);
    // <= 4 input channels: vector ALUs (conv_smallco.hip) - for the DGRAD role only (no bias, no BatchNorm sums: the image head's
    // dgrad, 68 -> 56 us).  The first layer's FORWARD gains 3 us there (47 -> 44) and stays on this kernel: another summation order
    // moves its outputs by 1e-7, which at the small-batch goldens flips single ReLU units further up (b2_n8_vib: one unit of
    // features.13, global gradient norm 3e-4 off instead of 2e-6; tests/diagnostics/vib_grad_diag.py).
    if (!aff && jvae_conv5_smallci_ok(Cin, H, W, Cout, OW, S, P, !bias && !stats))
        return jvae_conv5_smallci(in, w, swap, flip, bias, out, N, Cin, W, Cout, ws, st, stats, nsplit);
    {   // packed weights: the step's cache slot (refreshed once per step, pack_cache.hip) or this call's workspace
        bool fresh = true;
        float* slot = (float*)jvae_pack_cache_get(JVAE_PACK_F32, w, Cin, Cout, swap, flip, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            int rc = jvae_conv5_pack(w, ws, Cin, Cout, swap, flip, st);
            if (rc) return rc;
        }
    }
    FwdP p{in, ws, bias, out, N, Cin, H, W, (Cout + 31) / 32 * 32, P, Cout, stats, aff ? *aff : InAff{nullptr, nullptr, 0}};
    struct Fin { int* n; ~Fin() { if (n) *n = g_last_splits; } } fin{nsplit};
    if (S == 1) {
        switch (OW) {
            case 8: return launch_fwd_ow<1, 8>(p, st);
            case 16: return launch_fwd_ow<1, 16>(p, st);
            case 32: return launch_fwd_ow<1, 32>(p, st);
            case 64: return launch_fwd_ow<1, 64>(p, st);
        }
    } else {
        switch (OW) {
            case 8: return launch_fwd_ow<2, 8>(p, st);
            case 16: return launch_fwd_ow<2, 16>(p, st);
            case 32: return launch_fwd_ow<2, 32>(p, st);
            case 64: return launch_fwd_ow<2, 64>(p, st);
        }
    }
    return JVAE_ENOTSUP;
}
