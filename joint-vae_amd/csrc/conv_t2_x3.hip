// Stride-2 5x5 transposed convolution (padding 2, output_padding 1: H -> 2H), 4-phase sub-pixel form, on the bf16 matrix
// cores with exact 3-way operand splitting: the arithmetic of conv_x3.hip applied to the operator of conv_t2_mfma.hip.
//
//   big[n][o][2a + r][2b + q] = bias[o] + sum_c sum_{kh = r (mod 2), kw = q (mod 2)}
//                               small[n][c][a + (r + 2 - kh)/2][b + (q + 2 - kw)/2] * W[c][o][kh*5 + kw]
//
// Serves ConvTranspose2d(5, stride 2, padding 2, output_padding 1) forward (imager.6 / imager.12 of deconv32) and the
// dgrad of Conv2d(5, stride 2, padding 2) (features.3 / features.9 of conv32), fp32 NCHW in and out.
//
// Mapping: workgroup = 4 waves = 128 small-grid pixels x 32 output channels, K step = 16 input channels.  The small
// tensor's patch (one halo row / column) is split into three bf16 planes of 16-byte units exactly as in conv_x3.hip;
// weights are staged per KERNEL ROW (5 taps): a row kh fixes the output row phase r = kh & 1 and the patch row
// (r + 2 - kh)/2, its five taps alternate between the two column phases.  Four accumulator sets (one per phase), six
// bf16 MFMAs per tap; the two column phases of an output row are stored as 8-byte pairs.
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"
#include "conv_x3.h"
#include "pack_elems.h"

namespace {

typedef x3_bf16x8 bf16x8;
typedef x3_u32x4 u32x4;
typedef x3_f32x2 f32x2;

struct T2X3P {
    const float* in;     // small (N, C, HS, WS) fp32
    const u32x4* wp;     // split weights, layout of x3_wpack_kernel: (KB*5, 30, OP) units
    const float* bias;   // (O) or null
    float* out;          // big (N, O, 2HS, 2WS)
    int N, C, O;
    float* stats;        // optional (O, gridDim.x, 2)
    InAff aff;           // deferred BatchNorm(+ReLU) of the input
};

template <int WS>
struct T2X3Geom {
    static constexpr int HS = WS;
    static constexpr int PIX = 128;
    static constexpr int HSWS = HS * WS;
    static constexpr int NIMG = PIX >= HSWS ? PIX / HSWS : 1;
    static constexpr int TH = PIX >= HSWS ? HS : PIX / WS;
    static constexpr int ROWS = TH + 2;
    static constexpr int WP = WS + 2;                          // units per patch row: data at column 1
    static constexpr int CH = ROWS * WP;
    static constexpr int XS = NIMG * 2 * CH;                   // patch units of one plane (16 channels)
    static constexpr int WGS = 3 * 5 * 2 * 32;                 // weight units of one kernel row
    static constexpr int LDS_BYTES = (3 * XS + 2 * WGS) * 16;
};

template <int WS, bool AFF>
__global__ __launch_bounds__(256, 2) void convt2_x3_kernel(T2X3P p) {
    using G = T2X3Geom<WS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* Xs = reinterpret_cast<u32x4*>(lds_raw);            // [3 planes][XS]
    u32x4* Ws = Xs + 3 * G::XS;                                // [2 buffers][WGS]
    __shared__ float ctab[AFF ? 2 * 256 : 1];
    __shared__ float bias_s[32];                               // this workgroup's bias values, fetched at the start (conv_x3.hip)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int TILES_PER_IMG = G::HSWS >= G::PIX ? G::HSWS / G::PIX : 1;
    const int img0 = (G::HSWS >= G::PIX) ? (int)(blockIdx.x / TILES_PER_IMG) : (int)blockIdx.x * G::NIMG;
    const int row0 = (G::HSWS >= G::PIX) ? (int)(blockIdx.x % TILES_PER_IMG) * G::TH : 0;
    const int o0 = blockIdx.y * 32;
    const int KB = (p.C + 15) / 16;
    const int OP = p.O;                                        // multiple of 32 (jvae_convt2_ok)
    if (tid < 32) bias_s[tid] = p.bias ? p.bias[o0 + tid] : 0.f;   // visible after the first barrier of the K loop

    if (AFF)
        for (int i = tid; i < KB * 16; i += 256) {
            const bool ok = i < p.C;
            ctab[i] = ok ? p.aff.sc[i] : 0.f;
            ctab[256 + i] = ok ? p.aff.sh[i] : 0.f;
        }
    for (int i = tid; i < 3 * G::XS; i += 256) Xs[i] = u32x4{0u, 0u, 0u, 0u};

    const int pix = wave * 32 + l31;
    const int im = pix / (G::TH * WS), rem = pix % (G::TH * WS);
    const int pr = rem / WS, pc = rem % WS;
    const int pixoff = im * (2 * G::CH) + half * G::CH + (pr + 1) * G::WP + pc + 1;     // centre of the 3x3

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][q][e] = 0.f;

    constexpr int W2 = WS / 2;
    constexpr int XPAIRS = G::NIMG * 2 * G::ROWS * W2;
    constexpr int XU = (XPAIRS + 255) / 256, WU = (G::WGS + 255) / 256;
    f32x2 rx[XU][8];
    u32x4 rw[WU];
    const float* xsrc[XU];
    const u32x4* wsrc[WU];
    const long cstride = (long)G::HS * WS;
#pragma unroll
    for (int k = 0; k < XU; ++k) {
        const int u = tid + k * 256;
        const int xp = u % W2;
        int t = u / W2;
        const int lr = t % G::ROWS; t /= G::ROWS;
        const int h = t % 2, i2 = t / 2;
        const int ir = row0 - 1 + lr, n = img0 + i2;
        const bool ok = u < XPAIRS && ir >= 0 && ir < G::HS && n < p.N;
        xsrc[k] = p.in + (((long)(ok ? n : 0) * p.C + h * 8) * G::HS + (ok ? ir : 0)) * WS + 2 * xp;
    }
#pragma unroll
    for (int k = 0; k < WU; ++k) {
        const int u = min(tid + k * 256, G::WGS - 1);
        wsrc[k] = p.wp + (long)(u / 32) * OP + o0 + u % 32;
    }
    auto gloadX = [&](int kb) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int h = ((tid + k * 256) / (W2 * G::ROWS)) % 2;
            const float* src = xsrc[k] + (long)kb * 16 * cstride;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const bool okc = kb * 16 + h * 8 + ci < p.C;
                rx[k][ci] = *reinterpret_cast<const f32x2*>(okc ? src + ci * cstride : xsrc[k]);
            }
        }
    };
    auto gloadW = [&](int g) {
#pragma unroll
        for (int k = 0; k < WU; ++k) rw[k] = wsrc[k][(long)g * 30 * OP];
    };
    auto lstoreX = [&](int kb) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            if (u < XPAIRS) {
                const int xp = u % W2;
                int t = u / W2;
                const int lr = t % G::ROWS; t /= G::ROWS;
                const int h = t % 2, i2 = t / 2;
                const int ir = row0 - 1 + lr, n = img0 + i2;
                const bool live = ir >= 0 && ir < G::HS && n < p.N;
                f32x2 vv[8];
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    f32x2 v = (live && kb * 16 + h * 8 + ci < p.C) ? rx[k][ci] : f32x2{0.f, 0.f};
                    if (AFF && live) {
                        const int ch = kb * 16 + h * 8 + ci;
                        const float sc = ctab[ch], sh = ctab[256 + ch];
                        v[0] = fmaf(v[0], sc, sh);
                        v[1] = fmaf(v[1], sc, sh);
                        if (p.aff.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
                    }
                    vv[ci] = v;
                }
                u32x4 s[2][3];                      // [pixel][plane]: 8 channels = 4 packed pairs (x3_split2: two values at once)
#pragma unroll
                for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        unsigned hh, mm, ll;
                        x3_split2(f32x2{vv[2 * cp][j], vv[2 * cp + 1][j]}, hh, mm, ll);
                        s[j][0][cp] = hh; s[j][1][cp] = mm; s[j][2][cp] = ll;
                    }
                const int base = (i2 * 2 + h) * G::CH + lr * G::WP + 1 + 2 * xp;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Xs[pl * G::XS + base + j] = s[j][pl];
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
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    const int NG = KB * 5;
    gloadX(0);
    gloadW(0);
    __syncthreads();
    lstoreX(0);
    lstoreW(0);
    __builtin_amdgcn_sched_barrier(0);
    if (NG > 1) gloadW(1);
    lds_barrier();
    int g = 0;
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int kh = 0; kh < 5; ++kh, ++g) {                  // unrolled: the accumulator set depends on kh
            if (g + 1 < NG) lstoreW((g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG) gloadW(g + 2);
            if (kh == 3 && kb + 1 < KB) gloadX(kb + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int rr = kh & 1;
            const int dh = (rr + 2 - kh) / 2;
            const u32x4* Wb = Ws + (g & 1) * G::WGS + half * 32 + l31;
            u32x4 fa[2][3], fb[2][3];
            auto frag = [&](int kw, u32x4 (&a)[3], u32x4 (&b)[3]) {
                const int q = kw & 1, dw = (q + 2 - kw) / 2;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[pl] = Wb[(pl * 5 + kw) * 64];
                    b[pl] = Xs[pl * G::XS + pixoff + dh * G::WP + dw];
                }
            };
            frag(0, fa[0], fb[0]);
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                if (kw + 1 < 5) frag(kw + 1, fa[(kw + 1) & 1], fb[(kw + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                // (weight plane, input plane), smallest partial products first
                constexpr int WPL[6] = {0, 2, 1, 0, 1, 0}, XPL[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[rr][kw & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, fa[kw & 1][WPL[t]]), __builtin_bit_cast(bf16x8, fb[kw & 1][XPL[t]]),
                        acc[rr][kw & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier();
        }
        if (kb + 1 < KB) {                                     // K step change: the patch is fully consumed
            lstoreX(kb + 1);
            lds_barrier();
        }
    }

    if (p.stats) {
        float* red = reinterpret_cast<float*>(lds_raw);       // [4 waves][32][2]; the loop ended with a barrier
        float sv[32];                                         // [sum | sum of squares][register row]
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
        if (tid < 32) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[(w * 32 + tid) * 2]; s2 += red[(w * 32 + tid) * 2 + 1]; }
            float* dst = p.stats + ((long)(o0 + tid) * gridDim.x + blockIdx.x) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }
    const int n = img0 + im;
    if (n >= p.N) return;
    const int a_ = row0 + pr;
    constexpr int HB = 2 * G::HS, WB = 2 * WS;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = o0 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const float bv = bias_s[o - o0];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float2 v = make_float2(acc[r][0][e] + bv, acc[r][1][e] + bv);
            *reinterpret_cast<float2*>(p.out + (((long)n * p.O + o) * HB + 2 * a_ + r) * WB + 2 * pc) = v;
        }
    }
}

static thread_local int g_t2x3_splits = 0;

template <int WS>
int launch_t2x3(const T2X3P& p, hipStream_t st) {
    using G = T2X3Geom<WS>;
    static_assert(G::LDS_BYTES + 2048 <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2_x3_kernel<WS, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2_x3_kernel<WS, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(G::HSWS >= G::PIX ? (unsigned)((long)p.N * G::HSWS / G::PIX) : (unsigned)((p.N + G::NIMG - 1) / G::NIMG),
              (unsigned)(p.O / 32));
    g_t2x3_splits = (int)grid.x;
    if (p.aff.sc) hipLaunchKernelGGL((convt2_x3_kernel<WS, true>), grid, dim3(256), G::LDS_BYTES, st, p);
    else hipLaunchKernelGGL((convt2_x3_kernel<WS, false>), grid, dim3(256), G::LDS_BYTES, st, p);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// taken over from conv_t2_mfma.hip when the split-bf16 mode is on and there is at least one full K step of channels
bool jvae_convt2_x3_ok(int C, int WS, int O) {
    if (!jvae_conv5_x3_enabled()) return false;
    return C >= 16 && C <= 256 && O % 32 == 0 && (WS == 8 || WS == 16 || WS == 32);
}

// w: the layer's weight read as [c][o][tap] (ConvTranspose2d layout / Conv2d dgrad); ws: jvae_conv5_x3_pack_bytes(C, O)
int jvae_convt2_x3(const float* in, const float* w, const float* bias, float* out, int N, int C, int WS, int O, float* ws,
                   hipStream_t st, float* stats, int* nsplit, const InAff* aff) {
    {
        bool fresh = true;
        float* slot = (float*)jvae_pack_cache_get(JVAE_PACK_X3, w, C, O, 1, 0, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            int rc = jvae_conv5_x3_wpack(w, ws, C, O, 1, 0, st);
            if (rc) return rc;
        }
    }
    T2X3P p{in, (const u32x4*)ws, bias, out, N, C, O, stats, aff ? *aff : InAff{nullptr, nullptr, 0}};
    struct Fin { int* n; ~Fin() { if (n) *n = g_t2x3_splits; } } fin{nsplit};
    switch (WS) {
        case 8: return launch_t2x3<8>(p, st);
        case 16: return launch_t2x3<16>(p, st);
        case 32: return launch_t2x3<32>(p, st);
    }
    return JVAE_ENOTSUP;
}
