// Stride-2 5x5 transposed convolution (padding 2, output_padding 1: H -> 2H) on the fp32 matrix cores,
// by the 4-phase sub-pixel decomposition: no multiply-by-zero work, no scatter.
//
//   big[n][o][2a + r][2b + q] = bias[o] + sum_c sum_{kh = r (mod 2), kw = q (mod 2)}
//                               small[n][c][a + (r + 2 - kh)/2][b + (q + 2 - kw)/2] * Wp[c][kh*5 + kw][o]
//
// i.e. output phase (r,q) is a stride-1 correlation of the SMALL tensor with the 3x3 / 3x2 / 2x3 / 2x2 taps of
// that parity (9 + 6 + 6 + 4 = 25).  Serves ConvTranspose2d(5, stride 2, padding 2, output_padding 1) forward
// (imager.6 / imager.12 of deconv32) and the dgrad of Conv2d(5, stride 2, padding 2) (features.3 / features.9
// of conv32); both weight layouts read as [c][o][tap] (pack with swap = 1, flip = 0).
//
// Mapping: a wave owns 32 consecutive small-grid pixels (MFMA columns j) x NT*32 output channels (rows i) and
// keeps the 4 phases in 4 accumulator sets; per input-channel pair it issues 9 patch reads (the 3x3
// neighbourhood, shared by all phases), 25*NT weight reads and 25*NT MFMAs.  The two q-phases of a row are
// stored together as 8-byte pairs, so each store instruction still writes contiguous memory.
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"

namespace {

struct T2P {
    const float* in;     // small (N, C, HS, WS)
    const float* wp;     // packed (C, 25, O)
    const float* bias;   // (O) or null
    float* out;          // big (N, O, 2HS, 2WS)
    int N, C, O;
    float* stats;        // optional (O, gridDim.x, 2): per-workgroup sum / sum of squares of (out - bias)
    InAff aff;           // deferred BatchNorm(+ReLU) of the input
};

template <int WS, int NT, int CC>
struct T2Geom {
    static constexpr int HS = WS;
    static constexpr int PIX = 128;                              // small-grid pixels per workgroup (4 waves x 32)
    static constexpr int HSWS = HS * WS;
    static constexpr int NIMG = PIX >= HSWS ? PIX / HSWS : 1;
    static constexpr int TH = PIX >= HSWS ? HS : PIX / WS;
    static constexpr int ROWS = TH + 2;
    static constexpr int WP = WS + 8;                            // data at col 4, one halo column each side used
    static constexpr int CH = ROWS * WP;
    static constexpr int XS = NIMG * CC * CH;
    static constexpr int WCOLS = NT * 32;
    static constexpr int WSZ = CC * 25 * WCOLS;
};

// NT = 1: 64 accumulator registers; <= 128 VGPRs keeps 4 workgroups per CU resident (measured: 134 VGPRs = 3 per CU
// costs 15 %)
template <int WS, int NT, int CC, int AFF>      // AFF: 0 plain input, 1 deferred BatchNorm (+ReLU by flag), 2 ... + leaky ReLU
__global__ __launch_bounds__(256, NT == 1 ? 4 : 2) void convt2_kernel(T2P p) {
    using G = T2Geom<WS, NT, CC>;
    __shared__ __attribute__((aligned(16))) float lds[G::XS + G::WSZ];
    float* Xs = lds;
    float* Ws = lds + G::XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int TILES_PER_IMG = G::HSWS >= G::PIX ? G::HSWS / G::PIX : 1;
    const int img0 = (G::HSWS >= G::PIX) ? (int)(blockIdx.x / TILES_PER_IMG) : (int)blockIdx.x * G::NIMG;
    const int row0 = (G::HSWS >= G::PIX) ? (int)(blockIdx.x % TILES_PER_IMG) * G::TH : 0;
    const int o0 = blockIdx.y * G::WCOLS;

    for (int i = tid; i < G::XS / 4; i += 256) reinterpret_cast<f32x4*>(Xs)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int pix = wave * 32 + l31;
    const int im = pix / (G::TH * WS), rem = pix % (G::TH * WS);
    const int pr = rem / WS, pc = rem % WS;
    const int pixoff = im * (CC * G::CH) + (pr + 1) * G::WP + pc + 4 + half * G::CH;   // centre of the 3x3

    f32x16 acc[2][2][NT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][q][t][e] = 0.f;

    constexpr int W4 = WS / 4;
    constexpr int XUNITS = G::NIMG * CC * G::ROWS * W4;
    constexpr int WUNITS = G::WSZ / 4;

    constexpr int XU = (XUNITS + 255) / 256, WU = (WUNITS + 255) / 256;
    f32x4 rx[XU], rw[WU];
    float rsc[AFF ? XU : 1], rsh[AFF ? XU : 1];   // deferred-BatchNorm coefficients, applied when the units go to LDS
    auto gload = [&](int c0) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            const int x4 = u % W4;
            int t = u / W4;
            const int lr = t % G::ROWS; t /= G::ROWS;
            const int c = t % CC, i2 = t / CC;
            const int ir = row0 - 1 + lr, n = img0 + i2, ch = c0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            float sc = 0.f, sh = 0.f;
            if (u < XUNITS && ir >= 0 && ir < G::HS && n < p.N && ch < p.C) {
                v = *reinterpret_cast<const f32x4*>(p.in + (((long)n * p.C + ch) * G::HS + ir) * WS + x4 * 4);
                if (AFF) { sc = p.aff.sc[ch]; sh = p.aff.sh[ch]; }
            }
            rx[k] = v;
            if (AFF) { rsc[k] = sc; rsh[k] = sh; }
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * 256;
            const int col4 = u % (G::WCOLS / 4), kr = u / (G::WCOLS / 4);
            const int ch = c0 + kr / 25;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < WUNITS && ch < p.C)
                v = *reinterpret_cast<const f32x4*>(p.wp + ((long)(c0 * 25 + kr)) * p.O + o0 + col4 * 4);
            rw[k] = v;
        }
    };
    gload(0);
    for (int c0 = 0; c0 < p.C; c0 += CC) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            if (u < XUNITS) {
                const int x4 = u % W4;
                int t = u / W4;
                const int lr = t % G::ROWS; t /= G::ROWS;
                const int c = t % CC, i2 = t / CC;
                *reinterpret_cast<f32x4*>(&Xs[(i2 * CC + c) * G::CH + lr * G::WP + 4 + x4 * 4]) =
                    AFF == 2 ? aff4_leaky(rx[k], rsc[AFF ? k : 0], rsh[AFF ? k : 0]) : (AFF ? aff4(rx[k], rsc[AFF ? k : 0], rsh[AFF ? k : 0], p.aff.relu) : rx[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * 256;
            if (u < WUNITS) reinterpret_cast<f32x4*>(Ws)[u] = rw[k];
        }
        __syncthreads();
        if (c0 + CC < p.C) gload(c0 + CC);
        // one-deep software pipeline: the weight fragment of step j+1 (and, mid-way through a channel pair, the 3x3
        // neighbourhood of the next pair) is read from LDS before the MFMA of step j is issued
        constexpr int STEPS = (CC / 2) * 25;
        float nb[2][3][3], wa[2][NT];
        auto load_nb = [&](int cp, float (&d)[3][3]) {
#pragma unroll
            for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
                for (int dw = -1; dw <= 1; ++dw)
                    d[dh + 1][dw + 1] = Xs[pixoff + (cp * 2) * G::CH + dh * G::WP + dw];
        };
        auto load_w = [&](int j, float (&d)[NT]) {
            const int cp = j / 25, tap = j % 25;
#pragma unroll
            for (int t = 0; t < NT; ++t) d[t] = Ws[((cp * 2 + half) * 25 + tap) * G::WCOLS + t * 32 + l31];
        };
        load_nb(0, nb[0]);
        load_w(0, wa[0]);
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int cp = j / 25, tap = j % 25;
            const int kh = tap / 5, kw = tap % 5;
            if (j + 1 < STEPS) load_w(j + 1, wa[(j + 1) & 1]);
            if (tap == 12 && cp + 1 < CC / 2) load_nb(cp + 1, nb[(cp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int r = kh & 1, q = kw & 1;
            const int dh = (r + 2 - kh) / 2, dw = (q + 2 - kw) / 2;
            const float b = nb[cp & 1][dh + 1][dw + 1];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[r][q][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j & 1][t], b, acc[r][q][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (p.stats) {
        __syncthreads();
        float* red = lds;                                     // [4 waves][NT*32][2]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float sv[32];                                     // [sum | sum of squares][register row]
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { const float v = acc[r][q][t][e]; s1 += v; s2 += v * v; }
                sv[e] = s1;
                sv[16 + e] = s2;
            }
            // lane l31 receives the half-wave total of sv[l31]
            const float tot = half_wave_reduce32(sv);
            const int e = l31 & 15, ch = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            red[(wave * G::WCOLS + ch) * 2 + (l31 >> 4)] = tot;
        }
        __syncthreads();
        if (tid < G::WCOLS) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[(w * G::WCOLS + tid) * 2]; s2 += red[(w * G::WCOLS + tid) * 2 + 1]; }
            float* dst = p.stats + ((long)(o0 + tid) * gridDim.x + blockIdx.x) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }
    const int n = img0 + im;
    if (n >= p.N) return;
    const int a_ = row0 + pr;
    constexpr int HB = 2 * G::HS, WB = 2 * WS;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int o = o0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            const float bv = p.bias ? p.bias[o] : 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float2 v = make_float2(acc[r][0][t][e] + bv, acc[r][1][t][e] + bv);
                *reinterpret_cast<float2*>(p.out + (((long)n * p.O + o) * HB + 2 * a_ + r) * WB + 2 * pc) = v;
            }
        }
}

static thread_local int g_t2_splits = 0;

template <int WS, int NT>
int launch_t2(const T2P& p, hipStream_t st) {
    using G = T2Geom<WS, NT, 4>;
    static_assert((G::XS + G::WSZ) * 4 <= 64 * 1024, "static LDS budget");
    dim3 grid(G::HSWS >= G::PIX ? (unsigned)((long)p.N * G::HSWS / G::PIX) : (unsigned)((p.N + G::NIMG - 1) / G::NIMG),
              (unsigned)(p.O / G::WCOLS));
    g_t2_splits = (int)grid.x;
    if (p.aff.sc && p.aff.relu == JVAE_ACT_LEAKY) hipLaunchKernelGGL((convt2_kernel<WS, NT, 4, 2>), grid, dim3(256), 0, st, p);
    else if (p.aff.sc) hipLaunchKernelGGL((convt2_kernel<WS, NT, 4, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((convt2_kernel<WS, NT, 4, 0>), grid, dim3(256), 0, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// small (C, HS, WS) --ConvT 5x5 s2 p2 op1--> big (O, 2HS, 2WS)
bool jvae_convt2_ok(int C, int HS, int WS, int O, int HB, int WB, int KH, int KW, int S, int P) {
    if (KH != 5 || KW != 5 || S != 2 || P != 2) return false;
    if (HS != WS || HB != 2 * HS || WB != 2 * WS) return false;
    if (WS != 8 && WS != 16 && WS != 32) return false;
    return O % 32 == 0 && C >= 1;
}

int jvae_convt2(const float* in, const float* wpacked, const float* bias, float* out, int N, int C, int WS, int O,
                hipStream_t st, float* stats, int* nsplit, const InAff* aff) {
    T2P p{in, wpacked, bias, out, N, C, O, stats, aff ? *aff : InAff{nullptr, nullptr, 0}};
    struct Fin { int* n; ~Fin() { if (n) *n = g_t2_splits; } } fin{nsplit};
    const bool two = false;      // NT = 2 needs 128 accumulator registers (1 wave/SIMD): one 32-channel tile per wave instead
    switch (WS) {
        case 8: return two ? launch_t2<8, 2>(p, st) : launch_t2<8, 1>(p, st);
        case 16: return two ? launch_t2<16, 2>(p, st) : launch_t2<16, 1>(p, st);
        case 32: return two ? launch_t2<32, 2>(p, st) : launch_t2<32, 1>(p, st);
    }
    return JVAE_ENOTSUP;
}
