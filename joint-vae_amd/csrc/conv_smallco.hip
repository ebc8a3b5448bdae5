// 5x5 stride-1 convolution with a handful of OUTPUT channels (the `!3x5+2` image head of every upsampler in
// conv-models.ini: 32 -> 3 channels on the full-resolution grid), on the vector ALUs.
//
// With 3 output channels a 32-wide MFMA row/column tile would be 90 % padding, so this layer runs as a
// register-blocked direct convolution instead: one thread owns 4 horizontally adjacent output pixels x CO
// channels (4*CO accumulators); per (input channel, kernel row) it reads 12 patch values with three aligned
// ds_read_b128, takes its 5*CO weights as scalar operands and issues 20*CO FMAs.  A 256-thread workgroup
// covers 1024 pixels (32 rows of a 32-wide image, 16 rows of a 64-wide one); input channels are staged 4 at a
// time, next chunk prefetched into registers under the FMAs.
#include <stdlib.h>
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"
#include "pack_elems.h"

namespace {

struct SmP {
    const float* in;     // (N, Cin, H, W)
    const float* w;      // (CO, Cin, 5, 5)   PyTorch Conv2d layout
    const float* bias;   // (CO) or null
    float* out;          // (N, CO, H, W)
    int N, Cin, P;
    InAff aff;           // deferred BatchNorm(+ReLU) of the input
};

// Round 4 form (the first form - weights as wave-uniform LDS broadcasts - left the tree in round 5).  What that one was bound by: per
// (input channel, kernel row) a wave issued 3 + 5 ds_read_b128 - 3 for its 12 patch values, 5 for the 5 x CO weights, read as
// wave-uniform LDS broadcasts - for 60 FMAs: 4 SIMDs x 8 reads x 4 LDS cycles = 128 LDS cycles per 120 FMA cycles.  The LDS,
// not the vector ALU, set the pace (107 us for 5.03 GFLOP at N = 1024: 47 TFLOP/s of 157).  Here the weights never touch the
// LDS: they are wave-uniform, so the compiler fetches them with scalar loads straight into SGPRs (s_load through the scalar
// cache; the 9.6 KB of the layer stay resident) and every FMA takes its weight as the instruction's scalar operand.  The LDS
// then serves 3 reads per 60 FMAs.  CC = 4 input channels per stage (23 KB): four workgroups per CU, so the 1024 workgroups
// of the 32 x 32 head (one image each) are all resident at once - the first form ran 768 at a time, a second round for the
// last third.
template <int OW, int CO, int CC, bool LEAKY = false>
__global__ __launch_bounds__(256, 4) void conv5_smallco2_kernel(SmP p) {
    constexpr int OH = OW;
    constexpr int TH = 1024 / OW;                  // rows per workgroup
    constexpr int ROWS = TH + 4;
    constexpr int WP = OW + 8;                     // data at col 4 (16-byte aligned), halo 2 each side (P <= 4)
    constexpr int CH = ROWS * WP;
    constexpr int XS = CC * CH;
    __shared__ __attribute__((aligned(16))) float Xs[XS];
    __shared__ float ctab[2 * 256];                // (scale, shift) of the deferred BatchNorm, all input channels (host: Cin <= 256)

    const int tid = threadIdx.x;
    constexpr int TPI = OH / TH;                   // tiles per image
    const int n = blockIdx.x / TPI, row0 = (blockIdx.x % TPI) * TH;
    constexpr int XQ = OW / 4;
    const int r = tid / XQ, xq = tid % XQ;         // this thread: row r, pixels 4*xq .. 4*xq+3

    constexpr int W4 = OW / 4;
    constexpr int XUNITS = CC * ROWS * W4;
    constexpr int XU = (XUNITS + 255) / 256;
    f32x4 rx[XU];
    const int in_row0 = row0 - p.P;
    if (p.aff.sc)
        for (int i = tid; i < p.Cin; i += 256) { ctab[i] = p.aff.sc[i]; ctab[256 + i] = p.aff.sh[i]; }
    auto gload = [&](int c0) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            const int x4 = u % W4;
            const int t = u / W4;
            const int lr = t % ROWS, c = t / ROWS;
            const int ir = in_row0 + lr, ch = c0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < XUNITS && ir >= 0 && ir < OH && ch < p.Cin)
                v = *reinterpret_cast<const f32x4*>(p.in + (((long)n * p.Cin + ch) * OH + ir) * OW + x4 * 4);
            rx[k] = v;
        }
    };
    gload(0);
    // halo columns: zeroed once, never written again (rows are rewritten per stage, zeros outside the image)
    for (int i = tid; i < CC * ROWS * 2; i += 256) {
        f32x4* rowp = reinterpret_cast<f32x4*>(&Xs[(i >> 1) * WP]);
        rowp[(i & 1) ? (WP / 4 - 1) : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    float acc[CO][4];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;

    // constant address space: a wave-uniform load from it is a scalar load (s_load_dword*) whatever else the kernel stores
    typedef const __attribute__((address_space(4))) float* const_f32_p;
    const const_f32_p wq = (const_f32_p)(unsigned long long)p.w;
    for (int c0 = 0; c0 < p.Cin; c0 += CC) {
        __syncthreads();                           // every wave is past the previous stage's reads
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int u = tid + k * 256;
            if (u < XUNITS) {
                const int x4 = u % W4;
                const int t = u / W4;
                const int lr = t % ROWS, c = t / ROWS;
                const int ir = in_row0 + lr, ch = c0 + c;
                f32x4 v = rx[k];                   // padding rows / missing channels hold zeros and stay zeros
                if (p.aff.sc && ir >= 0 && ir < OH && ch < p.Cin) v = LEAKY ? aff4_leaky(v, ctab[ch], ctab[256 + ch]) : aff4(v, ctab[ch], ctab[256 + ch], p.aff.relu);
                *reinterpret_cast<f32x4*>(&Xs[c * CH + lr * WP + 4 + x4 * 4]) = v;
            }
        }
        __syncthreads();
        if (c0 + CC < p.Cin) gload(c0 + CC);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const int ch = min(c0 + c, p.Cin - 1);                 // channels beyond Cin: zero patch, any valid weight
#pragma unroll
            for (int kh = 0; kh < 5; ++kh) {
                // 12 consecutive patch values starting at lds col 4*xq (input col 4*xq - 4)
                const float* row = &Xs[c * CH + (r + kh) * WP + 4 * xq];
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 4);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(row + 8);
                const float in12[12] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
                float wv[CO][5];                                   // wave-uniform: scalar loads, SGPR operands
#pragma unroll
                for (int o = 0; o < CO; ++o)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw) wv[o][kw] = wq[((long)o * p.Cin + ch) * 25 + kh * 5 + kw];
#pragma unroll
                for (int kw = 0; kw < 5; ++kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = in12[j + kw + 4 - 2];   // lds col = x + kw - P + 4 relative to 4*xq, P == 2
#pragma unroll
                        for (int o = 0; o < CO; ++o) acc[o][j] = fmaf(wv[o][kw], xv, acc[o][j]);
                    }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const float b = p.bias ? p.bias[o] : 0.f;
        f32x4 v = {acc[o][0] + b, acc[o][1] + b, acc[o][2] + b, acc[o][3] + b};
        *reinterpret_cast<f32x4*>(p.out + (((long)n * CO + o) * OH + row0 + r) * OW + 4 * xq) = v;
    }
}


// ---- the mirror case: a handful of INPUT channels, many output channels (round 4) ---------------------------------------------------
// First layer of conv32 (3 -> 32, forward with BatchNorm sums) and the dgrad of the image head (3 -> 32 with swapped / flipped
// weights).  K = 3 x 25 = 75 is a poor fit for the matrix cores - the fp32 MFMA kernel pads it to 100 and reached 46-66 TFLOP/s - so
// this is the same register-blocked direct convolution as above with the roles of the channel counts exchanged: one thread owns 4
// adjacent pixels x 8 output channels (32 accumulators), a 256-thread workgroup 1024 pixels x 8 channels (blockIdx.y = the
// channel group: the 3-channel patch is small enough to be staged by each of the four groups), weights are wave-uniform scalar
// operands (s_load from the raw weight tensor, whatever its layout: swap / flip are index arithmetic).  Per (input channel, kernel
// row): 3 ds_read_b128, 40 scalar loads, 160 FMAs.
struct SciP {
    const float* in;     // (N, CI, H, W)
    const float* wp;     // packed weights [o / 8][c][tap][o % 8] (pack_elems.h JVAE_PACK_SCI; swap / flip resolved by the pack)
    const float* bias;   // (Cout) or null
    float* out;          // (N, Cout, H, W)
    float* stats;        // optional (Cout, gridDim.x, 2): sums of (y - bias), (y - bias)^2 over this workgroup's pixels
    int N, Cout;
};

__global__ __launch_bounds__(256) void sci_wpack_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int O, long total,
                                                        int swap, int flip) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        jvae_pack_sci_elem(w, wp, i, C, O, swap, flip);
}

template <int OW, int CI>
__global__ __launch_bounds__(256, 4) void conv5_smallci_kernel(SciP p) {
    constexpr int OH = OW, OC = 8, GPW = 2;
    constexpr int TH = 1024 / OW;                  // rows per workgroup
    constexpr int ROWS = TH + 4;
    constexpr int WP = OW + 8;                     // data at col 4 (16-byte aligned), halo 2 each side
    constexpr int CH = ROWS * WP;
    __shared__ __attribute__((aligned(16))) float Xs[CI * CH];
    __shared__ float red[4][2 * OC];

    const int tid = threadIdx.x;
    constexpr int TPI = OH / TH;
    const int n = blockIdx.x / TPI, row0 = (blockIdx.x % TPI) * TH;
    constexpr int XQ = OW / 4;
    const int r = tid / XQ, xq = tid % XQ;

    // stage the patch (zeros outside the image: rows AND halo columns)
    constexpr int UNITS = CI * ROWS * (WP / 4);
    for (int u = tid; u < UNITS; u += 256) {
        const int x4 = u % (WP / 4);
        const int t = u / (WP / 4);
        const int lr = t % ROWS, c = t / ROWS;
        const int ir = row0 - 2 + lr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (x4 >= 1 && x4 <= XQ && ir >= 0 && ir < OH)
            v = *reinterpret_cast<const f32x4*>(p.in + (((long)n * CI + c) * OH + ir) * OW + (x4 - 1) * 4);
        *reinterpret_cast<f32x4*>(&Xs[c * CH + lr * WP + x4 * 4]) = v;
    }
    __syncthreads();

    // GPW channel groups per workgroup, one after the other on the same staged patch (halves the per-workgroup fixed cost)
#pragma unroll 1
    for (int gg = 0; gg < GPW; ++gg) {
    const int o0 = (blockIdx.y * GPW + gg) * OC;
    if (o0 >= p.Cout) break;                       // uniform
    float acc[OC][4];
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;

    typedef const __attribute__((address_space(4))) float* const_f32_p;      // uniform loads from it are scalar loads
    const const_f32_p wg = (const_f32_p)(unsigned long long)p.wp + (long)(blockIdx.y * GPW + gg) * (CI * 25 * OC);
    // NOT unrolled over (channel, kernel row): unrolled, the compiler hoists all 600 scalar loads to the top and spills SGPRs
#pragma unroll 1
    for (int c = 0; c < CI; ++c) {
#pragma unroll 1
        for (int kh = 0; kh < 5; ++kh) {
            const float* row = &Xs[c * CH + (r + kh) * WP + 4 * xq];
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 4);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(row + 8);
            const float in12[12] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
            float wv[5][OC];                                   // 40 consecutive floats: five s_load_dwordx8
#pragma unroll
            for (int kw = 0; kw < 5; ++kw)
#pragma unroll
                for (int o = 0; o < OC; ++o) wv[kw][o] = wg[((c * 5 + kh) * 5 + kw) * OC + o];
#pragma unroll
            for (int kw = 0; kw < 5; ++kw)
#pragma unroll
                for (int o = 0; o < OC; ++o)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(wv[kw][o], in12[j + kw + 2], acc[o][j]);
        }
    }
#pragma unroll
    for (int o = 0; o < OC; ++o) {
        if (o0 + o >= p.Cout) continue;
        const float b = p.bias ? p.bias[o0 + o] : 0.f;
        f32x4 v = {acc[o][0] + b, acc[o][1] + b, acc[o][2] + b, acc[o][3] + b};
        *reinterpret_cast<f32x4*>(p.out + (((long)n * p.Cout + o0 + o) * OH + row0 + r) * OW + 4 * xq) = v;
    }
    if (p.stats) {          // BatchNorm partial sums of this workgroup's tile (pivot = bias), fixed order
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int o = 0; o < OC; ++o) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += acc[o][j]; s2 += acc[o][j] * acc[o][j]; }
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) { red[wave][2 * o] = s1; red[wave][2 * o + 1] = s2; }
        }
        __syncthreads();
        if (tid < 2 * OC && o0 + (tid >> 1) < p.Cout) {
            const float t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            p.stats[((long)(o0 + (tid >> 1)) * gridDim.x + blockIdx.x) * 2 + (tid & 1)] = t;
        }
    }
        if (p.stats) __syncthreads();              // `red` is reused by the next group
    }
}

template <int OW>
int launch_sm(const SmP& p, int CO, hipStream_t st) {
    dim3 grid((unsigned)(p.N * (OW * OW / 1024)));
    const bool leaky = p.aff.sc && p.aff.relu == JVAE_ACT_LEAKY;
#define SM_CASE(CO_) \
    case CO_: \
        if (leaky) hipLaunchKernelGGL((conv5_smallco2_kernel<OW, CO_, 4, true>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((conv5_smallco2_kernel<OW, CO_, 4, false>), grid, dim3(256), 0, st, p); \
        break;
    switch (CO) {
        SM_CASE(1) SM_CASE(2) SM_CASE(3) SM_CASE(4)
        default: return JVAE_ENOTSUP;
    }
#undef SM_CASE
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool jvae_conv5_smallco_ok(int Cin, int H, int W, int Cout, int KH, int KW, int S, int P) {
    return KH == 5 && KW == 5 && S == 1 && P == 2 && H == W && (W == 32 || W == 64) && Cout >= 1 && Cout <= 4 && Cin >= 1;
}

int jvae_conv5_smallco(const float* in, const float* w, const float* bias, float* out, int N, int Cin, int W, int Cout,
                       hipStream_t st, const InAff* aff) {
    SmP p{in, w, bias, out, N, Cin, 2, aff ? *aff : InAff{nullptr, nullptr, 0}};
    if (p.aff.sc && Cin > 256) return JVAE_ENOTSUP;          // coefficient table of the deferred BatchNorm
    if (W == 32) return launch_sm<32>(p, Cout, st);
    if (W == 64) return launch_sm<64>(p, Cout, st);
    return JVAE_ENOTSUP;
}

// Forward-type 5x5 stride-1 'same' convolution with <= 4 input channels (conv_mfma.hip hands these over): swap / flip as there.
bool jvae_conv5_smallci_ok(int Cin, int H, int W, int Cout, int OW, int S, int P, bool dgrad_role) {
    // the dgrad role only (the image head's 3 -> 32 dgrad): for the first layer's FORWARD the kernel's summation order moves a
    // ReLU unit of a small-batch golden to the other branch for 3 us (profiles/NOTES.md, round 4); the JVAE_SMALLCI switch is gone
    if (!dgrad_role) return false;
    return Cin >= 1 && Cin <= 4 && Cout >= 8 && S == 1 && P == 2 && H == W && OW == W && (W == 32 || W == 64);
}

int jvae_conv5_smallci(const float* in, const float* w, int swap, int flip, const float* bias, float* out,
                       int N, int Cin, int W, int Cout, float* ws, hipStream_t st, float* stats, int* nsplit) {
    {   // packed weights: the step's cache slot (refreshed once per step, pack_cache.hip) or this call's workspace
        bool fresh = true;
        float* slot = (float*)jvae_pack_cache_get(JVAE_PACK_SCI, w, Cin, Cout, swap, flip, &fresh);
        if (slot) ws = slot;
        if (!slot || !fresh) {
            const long total = jvae_pack_elems(JVAE_PACK_SCI, Cin, Cout);
            hipLaunchKernelGGL(sci_wpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, ws, Cin, Cout, total, swap, flip);
            JVAE_LAUNCH_CHECK();
        }
    }
    SciP p{in, ws, bias, out, stats, N, Cout};
    dim3 grid((unsigned)(N * (W * W / 1024)), (unsigned)((Cout + 15) / 16));      // 2 channel groups of 8 per workgroup
    if (nsplit) *nsplit = stats ? (int)grid.x : 0;
#define SCI_CASE(W_, C_) hipLaunchKernelGGL((conv5_smallci_kernel<W_, C_>), grid, dim3(256), 0, st, p); break;
    if (W == 32) {
        switch (Cin) { case 1: SCI_CASE(32, 1) case 2: SCI_CASE(32, 2) case 3: SCI_CASE(32, 3) case 4: SCI_CASE(32, 4) }
    } else {
        switch (Cin) { case 1: SCI_CASE(64, 1) case 2: SCI_CASE(64, 2) case 3: SCI_CASE(64, 3) case 4: SCI_CASE(64, 4) }
    }
#undef SCI_CASE
    JVAE_LAUNCH_CHECK();
    return 0;
}
