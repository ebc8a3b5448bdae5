// Generic (any kernel size / stride / padding) convolution and transposed convolution for NCHW fp32:
// unfold ("im2col") of the big-side tensor + the MFMA GEMM of gemm.hip, all three directions.
//
// Reference ops replaced: nn.Conv2d / nn.ConvTranspose2d built by build_de_conv_layers
// (module/vae_layers/conv.py:186-196) and their autograd backward.
//
// Vocabulary (jvae_internal.h ConvGeom): the BIG side is the tensor that gets unfolded (conv input,
// transposed-conv output), the SMALL side lives on the folded grid (conv output, transposed-conv input);
// big position = small position * S + tap - P for both layer kinds, and both weight layouts are
// row-major [Cs][Cb*KH*KW].  Three primitives cover six directions:
//     fold_fwd   : Ys = W . unfold(Xb)        conv forward          | transposed-conv dgrad
//     fold_bwd   : Xb = fold(W^T . Ys)        conv dgrad            | transposed-conv forward
//     fold_wgrad : dW = Ys . unfold(Xb)^T     conv wgrad            | transposed-conv wgrad
// The hot 5x5 layers are taken over by the implicit LDS-patch kernels in conv_mfma.hip; this file stays
// the path for every other geometry (7x7 / 8x8 / 3x3 / 4x4 heads of conv32, deconv32, conv32+, ...).
#include "common.h"
#include "jvae_internal.h"

namespace {

// col(n,k,q) = Xb[n][cb][hs*S+kh-P][ws*S+kw-P]  (k = (cb,kh,kw), q = (hs,ws)), zero outside the image.
// Output address = n*sn + k*sk + q*sq so that both col layouts ([n][k][q] and [q][n][k]) are served.
// `kfast`: consecutive threads walk k (for the [q][n][k] layout) instead of q.
__global__ __launch_bounds__(256) void unfold_kernel(const float* __restrict__ xb, float* __restrict__ col,
                                                     ConvGeom g, int n0, int nimg, long sn, long sk, long sq,
                                                     int kfast) {
    const int Kd = g.Cb * g.KH * g.KW, Ps = g.Hs * g.Ws;
    const long total = (long)nimg * Kd * Ps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n, k, q;
        if (kfast) { k = (int)(i % Kd); long r = i / Kd; n = (int)(r % nimg); q = (int)(r / nimg); }
        else       { q = (int)(i % Ps); long r = i / Ps; k = (int)(r % Kd);  n = (int)(r / Kd); }
        const int kw = k % g.KW, kh = (k / g.KW) % g.KH, cb = k / (g.KW * g.KH);
        const int hs = q / g.Ws, ws = q % g.Ws;
        const int hb = hs * g.S + kh - g.P, wb = ws * g.S + kw - g.P;
        float v = 0.f;
        if (hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb)
            v = xb[(((long)(n0 + n) * g.Cb + cb) * g.Hb + hb) * g.Wb + wb];
        col[(long)n * sn + (long)k * sk + (long)q * sq] = v;
    }
}

// The same unfold with 32-bit index arithmetic and 16-byte stores (the generic kernel above spends its time in 64-bit
// divisions: 0.7 TB/s on a 105 MB buffer).  Four consecutive elements of the fastest output dimension per thread:
// KFAST: col[q][n][k], 4 consecutive k (Kd % 4 == 0); else col[n][k][q], 4 consecutive q of one output row (Ws % 4 == 0).
template <bool KFAST>
__global__ __launch_bounds__(256) void unfold4_kernel(const float* __restrict__ xb, float* __restrict__ col,
                                                      ConvGeom g, int n0, int nimg) {
    const unsigned Kd = g.Cb * g.KH * g.KW, Ps = g.Hs * g.Ws, KK = g.KH * g.KW;
    const unsigned inner4 = (KFAST ? Kd : Ps) / 4;
    const unsigned total = (KFAST ? Ps * (unsigned)nimg : (unsigned)nimg * Kd) * inner4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned row = i / inner4, e4 = (i - row * inner4) * 4;
        unsigned n, k, q;
        if (KFAST) { q = row / nimg; n = row - q * nimg; k = e4; }
        else       { n = row / Kd;   k = row - n * Kd;   q = e4; }
        unsigned cb = k / KK, t = k - cb * KK;
        unsigned kh = t / g.KW, kw = t - kh * g.KW;
        const unsigned hs = q / g.Ws, ws = q - hs * g.Ws;
        const float* src = xb + (long)(n0 + n) * g.Cb * g.Hb * g.Wb;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int hb = (int)(hs * g.S + kh) - g.P;
            const int wb = (int)((KFAST ? ws : ws + e) * g.S + kw) - g.P;
            v[e] = (hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb) ? src[((long)cb * g.Hb + hb) * g.Wb + wb] : 0.f;
            if (KFAST && ++kw == (unsigned)g.KW) { kw = 0; if (++kh == (unsigned)g.KH) { kh = 0; ++cb; } }
        }
        *reinterpret_cast<f32x4*>(col + (long)i * 4) = v;        // both layouts are dense in (row, inner) order
    }
}

// Xb[n][cb][hb][wb] (=|+=) bias[cb] + sum over taps of col(n,(cb,kh,kw),(hs,ws)) with hs*S+kh-P == hb.
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ col, float* __restrict__ xb,
                                                   const float* __restrict__ bias, ConvGeom g, int n0, int nimg,
                                                   long sn, long sk, long sq) {
    // 32-bit index arithmetic (the host launches this kernel only for fewer than 2^31 elements: a 64-bit div / mod chain per
    // element is ~300 vector instructions in front of a handful of loads)
    const unsigned total = (unsigned)nimg * g.Cb * g.Hb * g.Wb;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        unsigned r = i / (unsigned)g.Wb;
        const int wb = (int)(i - r * g.Wb);
        unsigned r2 = r / (unsigned)g.Hb;
        const int hb = (int)(r - r2 * g.Hb);
        const int n = (int)(r2 / (unsigned)g.Cb);
        const int cb = (int)(r2 - (unsigned)n * g.Cb);
        float acc = bias ? bias[cb] : 0.f;
        if (g.Hs * g.Ws < g.KH * g.KW) {
            // few folded positions (e.g. 2x2): walk them instead of the taps
            for (int hs = 0; hs < g.Hs; ++hs) {
                const int kh = hb + g.P - hs * g.S;
                if (kh < 0 || kh >= g.KH) continue;
                for (int ws = 0; ws < g.Ws; ++ws) {
                    const int kw = wb + g.P - ws * g.S;
                    if (kw < 0 || kw >= g.KW) continue;
                    const int k = (cb * g.KH + kh) * g.KW + kw;
                    acc += col[(long)n * sn + (long)k * sk + (long)(hs * g.Ws + ws) * sq];
                }
            }
            xb[(((long)(n0 + n) * g.Cb + cb) * g.Hb + hb) * g.Wb + wb] = acc;
            continue;
        }
        for (int kh = 0; kh < g.KH; ++kh) {
            const int th = hb + g.P - kh;
            if (th < 0 || th % g.S) continue;
            const int hs = th / g.S;
            if (hs >= g.Hs) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int tw = wb + g.P - kw;
                if (tw < 0 || tw % g.S) continue;
                const int ws = tw / g.S;
                if (ws >= g.Ws) continue;
                const int k = (cb * g.KH + kh) * g.KW + kw;
                acc += col[(long)n * sn + (long)k * sk + (long)(hs * g.Ws + ws) * sq];
            }
        }
        xb[(((long)(n0 + n) * g.Cb + cb) * g.Hb + hb) * g.Wb + wb] = acc;
    }
}

// partial[c][sp] = sum_{n in split sp, q} t[n][c][q]   (bias gradients; reference: autograd of the conv bias add).
// grid (C, nsplit): each block reduces a slice of images with 16-byte loads; channel_fold_kernel adds the slices in order.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ t, float* __restrict__ partial,
                                                          int N, int C, int P, int nsplit) {
    __shared__ float red[17];
    const int c = blockIdx.x, sp = blockIdx.y;
    const ImageRange ir = image_range(N, nsplit, sp);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    float s = 0.f;
    if ((P & 3) == 0) {
        const int P4 = P >> 2;
        const long cnt = (long)(ne - nb) * P4;
        for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long n = nb + i / P4, q = i % P4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(t + (n * C + c) * (long)P + q * 4);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        const long cnt = (long)(ne - nb) * P;
        for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long n = nb + i / P, q = i % P;
            s += t[(n * C + c) * (long)P + q];
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[(long)c * nsplit + sp] = s;
}

__global__ void channel_fold_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int nsplit,
                                    int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(long)c * nsplit + k];
    out[c] = (accumulate ? out[c] : 0.f) + s;
}

// images per fold_kernel launch: the kernel's element index is 32-bit
inline int fold_slab_images(const ConvGeom& g) {
    const long per = (long)g.Cb * g.Hb * g.Wb;
    long n = ((1L << 31) - 1) / (per > 0 ? per : 1);
    if (const char* e = getenv("JVAE_FOLD_SLAB")) { const long v = atol(e); if (v > 0 && v < n) n = v; }   // tests: force the slab path
    return (int)(n < 1 ? 1 : (n > g.N ? (g.N > 0 ? g.N : 1) : n));
}

inline int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

inline int grid_for4(long total4) {
    long b = (total4 + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

// dense col buffer of `nimg` images in layout [q][n][k] (kfast) or [n][k][q]
int launch_unfold(const float* xb, float* col, const ConvGeom& g, int n0, int nimg, bool kfast, hipStream_t st) {
    const long Kd = (long)g.Cb * g.KH * g.KW, Ps = (long)g.Hs * g.Ws, total = (long)nimg * Kd * Ps;
    const bool vec = total < (1L << 31) && (kfast ? Kd % 4 == 0 : g.Ws % 4 == 0) && ((uintptr_t)col & 15) == 0;
    if (vec && kfast)
        hipLaunchKernelGGL(unfold4_kernel<true>, dim3(grid_for4(total / 4)), dim3(256), 0, st, xb, col, g, n0, nimg);
    else if (vec)
        hipLaunchKernelGGL(unfold4_kernel<false>, dim3(grid_for4(total / 4)), dim3(256), 0, st, xb, col, g, n0, nimg);
    else if (kfast)
        hipLaunchKernelGGL(unfold_kernel, dim3(grid_for(total)), dim3(256), 0, st, xb, col, g, n0, nimg, Kd, 1L, (long)nimg * Kd, 1);
    else
        hipLaunchKernelGGL(unfold_kernel, dim3(grid_for(total)), dim3(256), 0, st, xb, col, g, n0, nimg, Kd * Ps, Ps, 1L, 0);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// Small folded grids (E4 of conv32: 2x2) make per-image products degenerate: batch over positions instead.
// (also the 6x6 / 5x5 grids of conv32+ / deconv32+: 36 / 25 positions)
inline bool pixel_batched(const ConvGeom& g) { return g.Hs * g.Ws <= 48; }

// pixel-batched wgrad as ONE product over (position, image) with S deterministic K slices
inline bool wgrad_joint(const ConvGeom& g) { return pixel_batched(g) && g.Hs * g.Ws > 1; }
inline int wgrad_slices(const ConvGeom& g) {
    const long Kd = (long)g.Cb * g.KH * g.KW, K = (long)g.Hs * g.Ws * g.N;
    const long tiles = (long)cdiv(g.Cs, 64) * cdiv(Kd, 64);
    for (int S = 16; S > 1; S >>= 1)
        if (tiles * S <= 2048 && K % S == 0 && K / S >= 256) return S;
    return 1;
}

// position-batched forward: K pieces when the (image, channel, position) tiles alone cannot fill the chip
inline int fwd_slices(const ConvGeom& g) {
    if (!pixel_batched(g)) return 1;
    const long Kd = (long)g.Cb * g.KH * g.KW;
    const long tiles = (long)cdiv(g.N, 64) * cdiv(g.Cs, 64) * g.Hs * g.Ws;
    if (tiles >= 512 || Kd < 1024) return 1;
    // one residency round of FOUR workgroups per CU: the phases of a K step (loads, split + LDS stores, fragment reads + MFMAs) run
    // one after the other inside a workgroup, so it takes several of them per SIMD to keep the matrix pipe fed (round 4, with the
    // coalesced partial-product layout: features.12 of conv32 51.4 -> 50.1 us against the two-per-CU choice of round 3)
    int s = (int)(1024 / tiles);
    return s > 8 ? 8 : (s < 1 ? 1 : s);
}

// Yt[q][n][cs] = Ys[n][cs][q]
__global__ __launch_bounds__(256) void small_transpose_kernel(const float* __restrict__ ys, float* __restrict__ yt,
                                                              int N, int Cs, int Ps) {
    const long total = (long)N * Cs * Ps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cs = (int)(i % Cs);
        const long r = i / Cs;
        const int n = (int)(r % N), q = (int)(r / N);
        yt[i] = ys[((long)n * Cs + cs) * Ps + q];
    }
}

inline long col_floats_per_image(const ConvGeom& g) { return (long)g.Cb * g.KH * g.KW * g.Hs * g.Ws; }

}  // namespace

size_t jvae_conv_generic_ws(const ConvGeom& g) {
    // bounded: at most ~1 GiB of unfold buffer, at least one image
    const long per = col_floats_per_image(g) * 4;
    long imgs = (1L << 30) / (per > 0 ? per : 1);
    if (imgs < 1) imgs = 1;
    if (imgs > g.N) imgs = g.N;
    if (pixel_batched(g)) imgs = g.N;            // [q][n][k] layout is not chunked (tiny grids only)
    size_t bytes = (size_t)(imgs * per);
    size_t extra = 0;
    if (wgrad_joint(g))                          // + transposed Ys + the K-slice partial products
        extra = 4 * ((size_t)g.N * g.Cs * g.Hs * g.Ws + (size_t)wgrad_slices(g) * g.Cs * g.Cb * g.KH * g.KW);
    const size_t fwd_extra = fwd_slices(g) > 1 ? 4 * (size_t)fwd_slices(g) * g.N * g.Cs * g.Hs * g.Ws : 0;
    return bytes + (extra > fwd_extra ? extra : fwd_extra);
}

static int chunk_images(const ConvGeom& g, size_t ws_bytes) {
    const long per = col_floats_per_image(g) * 4;
    long imgs = (long)(ws_bytes / (size_t)(per > 0 ? per : 1));
    if (imgs > g.N) imgs = g.N;
    return (int)imgs;
}

int jvae_fold_fwd(const ConvGeom& g, const float* xb, const float* w, const float* bias, float* ys,
                  float* ws, size_t ws_bytes, hipStream_t st) {
    const int Kd = g.Cb * g.KH * g.KW, Ps = g.Hs * g.Ws;
    if (pixel_batched(g)) {
        if (ws_bytes < (size_t)col_floats_per_image(g) * 4 * g.N) return JVAE_EWORKSPACE;
        // col[q][n][k];  Y_q[n][cs] = col_q[n][:] . W[cs][:]
        { const int rc = launch_unfold(xb, ws, g, 0, g.N, true, st); if (rc) return rc; }
        const int want = fwd_slices(g);
        if (want > 1) {
            // long K, few tiles: K pieces stored side by side (copies of the output layout), folded in a fixed order
            const long colf = (long)col_floats_per_image(g) * g.N, outf = (long)g.N * g.Cs * Ps;
            if (ws_bytes < 4 * (size_t)(colf + want * outf)) return JVAE_EWORKSPACE;
            float* part = ws + colf;
            int S = 0;
            // partial products as [slice][q][n][cs] (rows of cs: coalesced stores), folded into ys[n][cs][q]
            int rc = jvae_gemm_launch_part(g.N, g.Cs, Kd, Ps, ws, Kd, 1, (long)g.N * Kd, w, 1, Kd, 0,
                                           part, g.Cs, 1, (long)g.N * g.Cs, outf, want, &S, st);
            if (rc) return rc;
            return jvae_splitk_fold_qn(part, bias, ys, S, g.N, g.Cs, Ps, st);
        }
        return jvae_gemm_launch(g.N, g.Cs, Kd, Ps, ws, Kd, 1, (long)g.N * Kd, w, 1, Kd, 0,
                                ys, (long)g.Cs * Ps, Ps, 1, bias, bias ? 1 : 0, 0, 1, st);
    }
    const int chunk = chunk_images(g, ws_bytes);
    if (chunk < 1) return JVAE_EWORKSPACE;
    for (int n0 = 0; n0 < g.N; n0 += chunk) {
        const int ni = (g.N - n0 < chunk) ? g.N - n0 : chunk;
        { const int rc = launch_unfold(xb, ws, g, n0, ni, false, st); if (rc) return rc; }
        int rc = jvae_gemm_launch(g.Cs, Ps, Kd, ni, w, Kd, 1, 0, ws, Ps, 1, (long)Kd * Ps,
                                  ys + (long)n0 * g.Cs * Ps, Ps, 1, (long)g.Cs * Ps, bias, bias ? 2 : 0, 0, 1, st);
        if (rc) return rc;
    }
    return 0;
}

int jvae_fold_bwd(const ConvGeom& g, const float* ys, const float* w, const float* bias, float* xb,
                  float* ws, size_t ws_bytes, hipStream_t st) {
    const int Kd = g.Cb * g.KH * g.KW, Ps = g.Hs * g.Ws;
    if (pixel_batched(g)) {
        if (ws_bytes < (size_t)col_floats_per_image(g) * 4 * g.N) return JVAE_EWORKSPACE;
        // dcol_q[n][k] = sum_cs Ys[n][cs][q] W[cs][k].  Ys is strided by the positions (element stride Ps along cs): transposed
        // to Yt[q][n][cs] first (a 5 us copy) the product loads 16-byte groups instead of gathering scalars (59 -> 35 us for
        // features.12 of conv32, profiles/r03_step_trace_no_overlap.txt)
        const long colf = (long)col_floats_per_image(g) * g.N, ytf = (long)g.N * g.Cs * Ps;
        int rc;
        if (Ps > 1 && ws_bytes >= 4 * (size_t)(colf + ytf)) {
            float* yt = ws + colf;
            hipLaunchKernelGGL(small_transpose_kernel, dim3(grid_for(ytf)), dim3(256), 0, st, ys, yt, g.N, g.Cs, Ps);
            JVAE_LAUNCH_CHECK();
            rc = jvae_gemm_launch(g.N, Kd, g.Cs, Ps, yt, g.Cs, 1, (long)g.N * g.Cs, w, Kd, 1, 0,
                                  ws, Kd, 1, (long)g.N * Kd, nullptr, 0, 0, 1, st);
        } else {
            rc = jvae_gemm_launch(g.N, Kd, g.Cs, Ps, ys, (long)g.Cs * Ps, Ps, 1, w, Kd, 1, 0,
                                  ws, Kd, 1, (long)g.N * Kd, nullptr, 0, 0, 1, st);
        }
        if (rc) return rc;
        // fold_kernel indexes with 32 bits: slabs of images below 2^31 output elements (one launch for every real size)
        const int slab = fold_slab_images(g);
        for (int n0 = 0; n0 < g.N; n0 += slab) {
            const int ni = (g.N - n0 < slab) ? g.N - n0 : slab;
            hipLaunchKernelGGL(fold_kernel, dim3(grid_for((long)ni * g.Cb * g.Hb * g.Wb)), dim3(256), 0, st,
                               ws + (long)n0 * Kd, xb, bias, g, n0, ni, (long)Kd, 1L, (long)g.N * Kd);
            JVAE_LAUNCH_CHECK();
        }
        return 0;
    }
    const int chunk = chunk_images(g, ws_bytes);
    if (chunk < 1) return JVAE_EWORKSPACE;
    for (int n0 = 0; n0 < g.N; n0 += chunk) {
        const int ni = (g.N - n0 < chunk) ? g.N - n0 : chunk;
        // dcol_n[k][q] = sum_cs W[cs][k] Ys_n[cs][q]
        int rc = jvae_gemm_launch(Kd, Ps, g.Cs, ni, w, 1, Kd, 0, ys + (long)n0 * g.Cs * Ps, Ps, 1, (long)g.Cs * Ps,
                                  ws, Ps, 1, (long)Kd * Ps, nullptr, 0, 0, 1, st);
        if (rc) return rc;
        const int slab = fold_slab_images(g);
        for (int m0 = 0; m0 < ni; m0 += slab) {
            const int mi = (ni - m0 < slab) ? ni - m0 : slab;
            hipLaunchKernelGGL(fold_kernel, dim3(grid_for((long)mi * g.Cb * g.Hb * g.Wb)), dim3(256), 0, st,
                               ws + (long)m0 * Kd * Ps, xb, bias, g, n0 + m0, mi, (long)Kd * Ps, (long)Ps, 1L);
            JVAE_LAUNCH_CHECK();
        }
    }
    return 0;
}

// dW[Cs][Kd] (+)= sum_n Ys_n . unfold(Xb)_n^T.  dW must already hold the value to accumulate onto
// (the caller zeroes it for a plain gradient): partial products are added with float atomics.
int jvae_fold_wgrad(const ConvGeom& g, const float* xb, const float* ys, float* dw,
                    float* ws, size_t ws_bytes, hipStream_t st) {
    const int Kd = g.Cb * g.KH * g.KW, Ps = g.Hs * g.Ws;
    if (pixel_batched(g)) {
        if (ws_bytes < (size_t)col_floats_per_image(g) * 4 * g.N) return JVAE_EWORKSPACE;
        { const int rc = launch_unfold(xb, ws, g, 0, g.N, true, st); if (rc) return rc; }
        if (wgrad_joint(g)) {
            // dW[cs][k] += sum_{(q,n)} Yt[(q,n)][cs] col[(q,n)][k]: one product, K = Ps*N sliced over the batch
            // dimension into S partial results, folded in a fixed order
            const int S = wgrad_slices(g);
            const long colf = (long)col_floats_per_image(g) * g.N, ytf = (long)g.N * g.Cs * Ps;
            if (ws_bytes < 4 * (size_t)(colf + ytf + (long)S * g.Cs * Kd)) return JVAE_EWORKSPACE;
            float* yt = ws + colf;
            float* part = yt + ytf;
            hipLaunchKernelGGL(small_transpose_kernel, dim3(grid_for(ytf)), dim3(256), 0, st, ys, yt, g.N, g.Cs, Ps);
            JVAE_LAUNCH_CHECK();
            const long Ks = (long)Ps * g.N / S;
            int rc = jvae_gemm_launch(g.Cs, Kd, (int)Ks, S, yt, 1, g.Cs, Ks * g.Cs, ws, Kd, 1, Ks * Kd,
                                      part, Kd, 1, (long)g.Cs * Kd, nullptr, 0, 0, 1, st);
            if (rc) return rc;
            return jvae_splitk_fold(part, nullptr, dw, S, (long)g.Cs * Kd, Kd, 0, 1, st);
        }
        // dW[cs][k] += sum_q sum_n Ys[n][cs][q] col_q[n][k]: one accumulating product per folded position, in order
        // (deterministic: no float atomics)
        for (int q = 0; q < Ps; ++q) {
            int rc = jvae_gemm_launch(g.Cs, Kd, g.N, 1, ys + q, Ps, (long)g.Cs * Ps, 0, ws + (long)q * g.N * Kd, Kd, 1, 0,
                                      dw, Kd, 1, 0, nullptr, 0, 1, 1, st);
            if (rc) return rc;
        }
        return 0;
    }
    const int chunk = chunk_images(g, ws_bytes);
    if (chunk < 1) return JVAE_EWORKSPACE;
    for (int n0 = 0; n0 < g.N; n0 += chunk) {
        const int ni = (g.N - n0 < chunk) ? g.N - n0 : chunk;
        { const int rc = launch_unfold(xb, ws, g, n0, ni, false, st); if (rc) return rc; }
        int rc = jvae_gemm_launch(g.Cs, Kd, Ps, ni, ys + (long)n0 * g.Cs * Ps, Ps, 1, (long)g.Cs * Ps,
                                  ws, 1, Ps, (long)Kd * Ps, dw, Kd, 1, 0, nullptr, 0, 4, 1, st);
        if (rc) return rc;
    }
    return 0;
}

size_t jvae_channel_sum_ws_bytes(int C) { return sizeof(float) * (size_t)(C > 0 ? C : 1) * 64; }

// out[c] (+)= sum_{n,q} t[n][c][q], deterministic (no float atomics); ws: jvae_channel_sum_ws_bytes(C)
int jvae_channel_sum(const float* t, float* out, int N, int C, int P, int accumulate, float* ws, size_t ws_bytes,
                     hipStream_t st) {
    if (C <= 0) return 0;
    if (N <= 0) {
        if (!accumulate) {
            hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)C, st);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    }
    if (!ws || ws_bytes < jvae_channel_sum_ws_bytes(C)) return JVAE_EWORKSPACE;
    long work = (long)N * P;
    int ns = (int)(work / 4096);
    const int cap = (1024 + C - 1) / C;
    if (ns > cap) ns = cap;
    if (ns > 64) ns = 64;
    if (ns > N) ns = N;
    if (ns < 1) ns = 1;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, ns), dim3(256), 0, st, t, ws, N, C, P, ns);
    JVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(channel_fold_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, (const float*)ws, out, C, ns, accumulate);
    JVAE_LAUNCH_CHECK();
    return 0;
}
