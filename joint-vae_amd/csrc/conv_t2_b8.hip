// bf16 (B8 layout, see conv_b8.hip) stride-2 5x5 transposed convolution (padding 2, output_padding 1: H -> 2H) by the
// 4-phase sub-pixel decomposition of conv_t2_mfma.hip:
//
//   big[n][o][2a + r][2b + q] = bias[o] + sum_c sum_{kh = r (mod 2), kw = q (mod 2)}
//                               small[n][c][a + (r + 2 - kh)/2][b + (q + 2 - kw)/2] * W[c][o][kh][kw]
//
// Serves ConvTranspose2d(5, stride 2, padding 2, output_padding 1) forward and the dgrad of Conv2d(5, stride 2,
// padding 2); both read the weight as [c][o][tap] (b8 weight pack with swap = 1, flip = 0).
//
// A wave owns 32 consecutive small-grid pixels (MFMA columns) x 32 output channels (rows) and keeps the 4 phases in 4
// accumulator sets; per 16 input channels it issues 9 patch reads (the 3x3 neighbourhood, one ds_read_b128 each,
// shared by the phases), 25 weight reads and 25 v_mfma_f32_32x32x16_bf16.
#include "common.h"
#include "jvae_internal.h"
#include "conv_b8.h"
#include "pack_elems.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct T2B8P {
    const u32x4* in;     // small, B8 units (N, CBin, HS, WS)
    const u32x4* wp;     // packed weight units (KB, 25, 2, OP)
    const float* bias;   // (O) or null
    u32x2* out;          // big, B8 half units (N, CBout, 2HS, 2WS, 2)
    int N, CBin, OP, O, CBout;
    float* stats;        // optional (O, gridDim.x, 2)
    InAff aff;           // deferred BatchNorm(+ReLU) of the input
};

template <int WS, int NW>
struct T2B8Geom {
    static constexpr int HS = WS;
    static constexpr int PIX = NW * 32;                          // small-grid pixels per workgroup
    static constexpr int HSWS = HS * WS;
    static constexpr int NIMG = PIX >= HSWS ? PIX / HSWS : 1;
    static constexpr int TH = PIX >= HSWS ? HS : PIX / WS;
    static constexpr int ROWS = TH + 2;
    static constexpr int WP = WS + 2;                            // one halo column each side
    static constexpr int CH = ROWS * WP;
    static constexpr int XS = NIMG * 2 * CH;                     // units of one K step
    static constexpr int WSZ = 25 * 2 * 32;
    static constexpr int LDS_BYTES = (XS + WSZ) * 16;
};

template <int WS, int NW, bool AFF>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void convt2_b8_kernel(T2B8P p) {
    using G = T2B8Geom<WS, NW>;
    constexpr int NT_ = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Xs = reinterpret_cast<u32x4*>(lds_raw);
    u32x4* Ws = Xs + G::XS;
    __shared__ __attribute__((aligned(16))) float ctab[AFF ? 2 * 256 : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (AFF)
        for (int i = tid; i < p.CBin * 8; i += NW * 64) { ctab[i] = p.aff.sc[i]; ctab[256 + i] = p.aff.sh[i]; }
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int TILES_PER_IMG = G::HSWS >= G::PIX ? G::HSWS / G::PIX : 1;
    const int img0 = (G::HSWS >= G::PIX) ? (int)(blockIdx.x / TILES_PER_IMG) : (int)blockIdx.x * G::NIMG;
    const int row0 = (G::HSWS >= G::PIX) ? (int)(blockIdx.x % TILES_PER_IMG) * G::TH : 0;
    const int o0 = blockIdx.y * 32;

    for (int i = tid; i < G::XS; i += NT_) Xs[i] = u32x4{0u, 0u, 0u, 0u};

    const int pix = wave * 32 + l31;
    const int im = pix / (G::TH * WS), rem = pix % (G::TH * WS);
    const int pr = rem / WS, pc = rem % WS;
    const int pixoff = im * (2 * G::CH) + half * G::CH + (pr + 1) * G::WP + pc + 1;       // centre of the 3x3

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][q][e] = 0.f;

    constexpr int XUNITS = G::NIMG * 2 * G::ROWS * WS;
    constexpr int XU = (XUNITS + NT_ - 1) / NT_, WU = (G::WSZ + NT_ - 1) / NT_;
    u32x4 rx[XU], rw[WU];
    const int KB = (p.CBin + 1) / 2;
    auto gload = [&](int kb) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * NT_;
            const int x = u % WS;
            int t = u / WS;
            const int lr = t % G::ROWS; t /= G::ROWS;
            const int h = t % 2, i2 = t / 2;
            const int ir = row0 - 1 + lr, n = img0 + i2, cb = kb * 2 + h;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (u < XUNITS && ir >= 0 && ir < G::HS && n < p.N && cb < p.CBin)
                v = p.in[(((long)n * p.CBin + cb) * G::HS + ir) * WS + x];
            rx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * NT_;
            const int col = u % 32, th = u / 32;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (u < G::WSZ) v = p.wp[((long)kb * 50 + th) * p.OP + o0 + col];
            rw[k] = v;
        }
    };
    gload(0);
    for (int kb = 0; kb < KB; ++kb) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * NT_;
            if (u < XUNITS) {
                const int x = u % WS;
                int t = u / WS;
                const int lr = t % G::ROWS; t /= G::ROWS;
                const int h = t % 2, i2 = t / 2;
                u32x4 v = rx[k];
                if (AFF) {
                    const int ir = row0 - 1 + lr, n = img0 + i2, cb = kb * 2 + h;
                    if (ir >= 0 && ir < G::HS && n < p.N && cb < p.CBin) v = aff8(v, &ctab[cb * 8], &ctab[256 + cb * 8], p.aff.relu);
                }
                Xs[(i2 * 2 + h) * G::CH + lr * G::WP + 1 + x] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < WU; ++k) {
            const int u = tid + k * NT_;
            if (u < G::WSZ) Ws[u] = rw[k];
        }
        __syncthreads();
        if (kb + 1 < KB) gload(kb + 1);
        bf16x8 nb[3][3];
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw)
                nb[dh + 1][dw + 1] = __builtin_bit_cast(bf16x8, Xs[pixoff + dh * G::WP + dw]);
        // weight fragments two taps ahead of the MFMA that consumes them (32-cycle MFMAs: an LDS round trip is ~2 of them)
        u32x4 wa[3];
        wa[0] = Ws[(0 * 2 + half) * 32 + l31];
        wa[1] = Ws[(1 * 2 + half) * 32 + l31];
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            const int kh = tap / 5, kw = tap % 5;
            if (tap + 2 < 25) wa[(tap + 2) % 3] = Ws[((tap + 2) * 2 + half) * 32 + l31];
            __builtin_amdgcn_sched_barrier(0);
            const int r = kh & 1, q = kw & 1;
            const int dh = (r + 2 - kh) / 2, dw = (q + 2 - kw) / 2;
            acc[r][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[tap % 3]), nb[dh + 1][dw + 1],
                                                                acc[r][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (p.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds_raw);         // [NW waves][32][2]
        float sv[32];                                           // [sum | sum of squares][register row]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q) { const float v = acc[r][q][e]; s1 += v; s2 += v * v; }
            sv[e] = s1;
            sv[16 + e] = s2;
        }
        {   // lane l31 receives the half-wave total of sv[l31]
            const float tot = half_wave_reduce32(sv);
            const int e = l31 & 15, ch = (e & 3) + 8 * (e >> 2) + 4 * half;
            red[(wave * 32 + ch) * 2 + (l31 >> 4)] = tot;
        }
        __syncthreads();
        if (tid < 32 && o0 + tid < p.O) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { s1 += red[(w * 32 + tid) * 2]; s2 += red[(w * 32 + tid) * 2 + 1]; }
            float* dst = p.stats + ((long)(o0 + tid) * gridDim.x + blockIdx.x) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }
    const int n = img0 + im;
    if (n >= p.N) return;
    const int a_ = row0 + pr;
    constexpr int HB = 2 * G::HS, WB = 2 * WS;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int ob = o0 + 8 * rg + 4 * half;
        const int cb = ob >> 3;
        if (cb >= p.CBout) continue;
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (p.bias && ob + e < p.O) ? p.bias[ob + e] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bf16x4 v = {(__bf16)(acc[r][q][rg * 4 + 0] + bv[0]), (__bf16)(acc[r][q][rg * 4 + 1] + bv[1]),
                                  (__bf16)(acc[r][q][rg * 4 + 2] + bv[2]), (__bf16)(acc[r][q][rg * 4 + 3] + bv[3])};
                p.out[((((long)n * p.CBout + cb) * HB + 2 * a_ + r) * WB + 2 * pc + q) * 2 + half] =
                    __builtin_bit_cast(u32x2, v);
            }
    }
}

thread_local int g_t2b8_splits = 0;

template <int WS, int NW>
int launch_t2b8(const T2B8P& p, hipStream_t st) {
    using G = T2B8Geom<WS, NW>;
    static_assert(G::LDS_BYTES <= 64 * 1024, "LDS budget");
    dim3 grid(G::HSWS >= G::PIX ? (unsigned)((long)p.N * G::HSWS / G::PIX) : (unsigned)((p.N + G::NIMG - 1) / G::NIMG),
              (unsigned)(p.OP / 32));
    g_t2b8_splits = (int)grid.x;
    if (p.aff.sc) {
        if (p.CBin * 8 > 256) return JVAE_ENOTSUP;
        hipLaunchKernelGGL((convt2_b8_kernel<WS, NW, true>), grid, dim3(NW * 64), G::LDS_BYTES, st, p);
    } else {
        hipLaunchKernelGGL((convt2_b8_kernel<WS, NW, false>), grid, dim3(NW * 64), G::LDS_BYTES, st, p);
    }
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool jvae_convt2_b8_ok(int C, int HS, int WS, int O, int HB, int WB, int KH, int KW, int S, int P) {
    if (KH != 5 || KW != 5 || S != 2 || P != 2) return false;
    if (HS != WS || HB != 2 * HS || WB != 2 * WS) return false;
    if (WS != 4 && WS != 8 && WS != 16 && WS != 32) return false;
    return O >= 1 && C >= 1;
}

// small (N, ceil(C/8), WS, WS, 8) --ConvT 5x5 s2 p2 op1--> big (N, ceil(O/8), 2WS, 2WS, 8); ws: packed weights
int jvae_convt2_b8(const void* in, const float* w, const float* bias, void* out, int N, int C, int WS, int O,
                   void* ws, hipStream_t st, float* stats, int* nsplit, const InAff* aff) {
    {
        bool fresh = true;
        void* slot = jvae_pack_cache_get(JVAE_PACK_B8, w, C, O, 1, 0, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            int rc = jvae_conv5_b8_wpack(w, ws, C, O, 1, 0, st);
            if (rc) return rc;
        }
    }
    T2B8P p{(const u32x4*)in, (const u32x4*)ws, bias, (u32x2*)out, N, (C + 7) / 8, (O + 31) / 32 * 32, O, (O + 7) / 8, stats,
            aff ? *aff : InAff{nullptr, nullptr, 0}};
    struct Fin { int* n; ~Fin() { if (n) *n = g_t2b8_splits; } } fin{nsplit};
    switch (WS) {
        case 4: return launch_t2b8<4, 4>(p, st);         // 4x4 -> 8x8 (deconv32+): 8 images per workgroup
        case 8: return launch_t2b8<8, 4>(p, st);
        case 16: return launch_t2b8<16, 4>(p, st);
        case 32: return launch_t2b8<32, 4>(p, st);
    }
    return JVAE_ENOTSUP;
}
