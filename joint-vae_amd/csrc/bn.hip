// Training-mode BatchNorm2d (+ fused ReLU) for NCHW fp32: forward, backward, running-stat update.
// HBM-bound: every pass streams the activation once with 16-byte accesses.
//
// Reference: nn.BatchNorm2d(eps 1e-5, momentum 0.1, affine) inserted after every (de)conv by
// build_de_conv_layers (module/vae_layers/conv.py:214-220) followed by the activation (ReLU inplace).
// Forward statistics are the biased batch variance; running_var receives the unbiased one.
//
// Numerics: per-channel sums are taken on data shifted by the channel's first element (robust against
// |mean| >> std), per-block partials are combined in fp64 by a one-block finalize kernel.
#include <stdlib.h>
#include "common.h"
#include "jvae_internal.h"

namespace {

constexpr int MAX_SPLIT = 64;

// y = fmaf(x, scale, shift): one definition so that backward re-derives the forward's ReLU mask bit-exactly
__device__ __forceinline__ void bn_coef(float g, float b, float mean, float invstd, float* sc, float* sh) {
    *sc = g * invstd;
    *sh = b - mean * (g * invstd);
}

// float offset of the i-th float4 of channel c inside the images [nb, ne): 32-bit arithmetic, a shift when the plane size is a
// power of two.  (The loops used 64-bit i / P4 and i % P4 per 16 bytes: ~100 vector instructions per load, which is what made
// these HBM-bound kernels crawl beside the matrix-core kernels of the other stream - they compete for the same issue slots.)
struct Plane4Idx {
    unsigned p4; int sh; long stride, base;
    __device__ __forceinline__ long operator()(unsigned i) const {
        const unsigned n = sh >= 0 ? i >> sh : i / p4;
        return base + (long)n * stride + (long)(i - n * p4) * 4;
    }
};
__device__ __forceinline__ Plane4Idx plane4_idx(int nb, int C, int c, int P) {
    Plane4Idx u;
    u.p4 = (unsigned)(P >> 2);
    u.sh = (u.p4 & (u.p4 - 1)) == 0 ? __ffs((int)u.p4) - 1 : -1;
    u.stride = (long)C * P;
    u.base = ((long)nb * C + c) * P;
    return u;
}

// partial[c][s] = (sum(x-p), sum((x-p)^2)) over the images of split s;  p = x[0][c][0]
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                       int N, int C, int P, int nsplit, const float* __restrict__ pv) {
    __shared__ float red[17];
    const int c = blockIdx.x, s = blockIdx.y;
    const float pivot = pv ? pv[c] : x[(long)c * P];
    const ImageRange ir = image_range(N, nsplit, s);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    float s1 = 0.f, s2 = 0.f;
    if ((P & 3) == 0) {
        const unsigned cnt = (unsigned)(ne - nb) * (unsigned)(P >> 2);
        const Plane4Idx pi = plane4_idx(nb, C, c, P);
#pragma unroll 2
        for (unsigned i = threadIdx.x; i < cnt; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + pi(i));
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[j] - pivot; s1 += d; s2 += d * d; }
        }
    } else {
        const long cnt = (long)(ne - nb) * P;
        for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long n = nb + i / P, q = i % P;
            const float d = x[(n * C + c) * (long)P + q] - pivot;
            s1 += d; s2 += d * d;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        partial[((long)c * nsplit + s) * 2 + 0] = s1;
        partial[((long)c * nsplit + s) * 2 + 1] = s2;
    }
}

// Forward apply, one block per (channel, image chunk).  The block first folds the channel's partial sums
// (or the running statistics in eval mode) into y = fmaf(x, scale, shift); chunk 0 also publishes mean / invstd
// and updates the running statistics, so no separate finalize launch is needed.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* running_mean, float* running_var, long long* num_batches_tracked,
                                                       float* save_mean, float* save_invstd, float* __restrict__ y,
                                                       int N, int C, int P, int nsplit, int nchunk, float momentum, float eps,
                                                       int training, int relu, int ext_pivot, const float* __restrict__ pivot, int count_mult) {
    __shared__ float cs[2];
    __shared__ double dred[2][4];
    const int c = blockIdx.x, j = blockIdx.y;
    double s1 = 0., s2 = 0.;
    if (training) {                              // fold the channel's partial sums with the whole block (fp64)
        for (int s = threadIdx.x; s < nsplit; s += blockDim.x) {
            s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
            s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if ((threadIdx.x & 63) == 0) { dred[0][threadIdx.x >> 6] = s1; dred[1][threadIdx.x >> 6] = s2; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        float mean, invstd;
        if (training) {
            s1 = dred[0][0] + dred[0][1] + dred[0][2] + dred[0][3];
            s2 = dred[1][0] + dred[1][1] + dred[1][2] + dred[1][3];
            const double n = (double)N * P * count_mult;          // count_mult = ranks of a synchronised BatchNorm
            const double dm = s1 / n;
            double var = s2 / n - dm * dm;
            if (var < 0.) var = 0.;
            const double pv = ext_pivot ? (pivot ? (double)pivot[c] : 0.) : (double)x[(long)c * P];
            mean = (float)(pv + dm);
            invstd = (float)(1.0 / sqrt(var + (double)eps));
            if (j == 0) {
                save_mean[c] = mean;
                save_invstd[c] = invstd;
                if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                if (running_var) {
                    const float unbiased = (float)(n > 1. ? var * n / (n - 1.) : var);
                    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
                }
                if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
            }
        } else {
            mean = running_mean[c];
            invstd = rsqrtf(running_var[c] + eps);
        }
        bn_coef(g, b, mean, invstd, &cs[0], &cs[1]);
    }
    __syncthreads();
    const float sc = cs[0], sh = cs[1];
    const ImageRange ir = image_range(N, nchunk, j);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    if ((P & 3) == 0) {
        const unsigned cnt = (unsigned)(ne - nb) * (unsigned)(P >> 2);
        const Plane4Idx pi = plane4_idx(nb, C, c, P);
#pragma unroll 2
        for (unsigned i = threadIdx.x; i < cnt; i += 256) {
            const long off = pi(i);
            f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = fmaf(v[e], sc, sh); v[e] = jvae_act(t, relu); }
            *reinterpret_cast<f32x4*>(y + off) = v;
        }
    } else {
        const long cnt = (long)(ne - nb) * P;
        for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long n = nb + i / P, q = i % P;
            const long off = (n * C + c) * (long)P + q;
            const float t = fmaf(x[off], sc, sh);
            y[off] = jvae_act(t, relu);
        }
    }
}

// One block per channel: folds the partial sums (fp64, fixed order), publishes mean / invstd, updates the running
// statistics and writes the coefficients of y = fmaf(x, scale, shift) - the BatchNorm itself is then applied by the
// consuming convolution while it stages its input (InAff), so the normalised tensor never exists in HBM.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* running_mean, float* running_var, long long* num_batches_tracked,
                                                         float* save_mean, float* save_invstd,
                                                         float* __restrict__ scale, float* __restrict__ shift,
                                                         int N, int C, int P, int nsplit, float momentum, float eps,
                                                         int training, int ext_pivot, const float* __restrict__ pivot) {
    __shared__ double dred[2][4];
    const int c = blockIdx.x, l = threadIdx.x;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float mean, invstd;
    if (training) {
        double s1 = 0., s2 = 0.;
        for (int s = l; s < nsplit; s += 256) {
            s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
            s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if ((l & 63) == 0) { dred[0][l >> 6] = s1; dred[1][l >> 6] = s2; }
        __syncthreads();
        s1 = (dred[0][0] + dred[0][1]) + (dred[0][2] + dred[0][3]);
        s2 = (dred[1][0] + dred[1][1]) + (dred[1][2] + dred[1][3]);
        const double n = (double)N * P;
        const double dm = s1 / n;
        double var = s2 / n - dm * dm;
        if (var < 0.) var = 0.;
        const double pv = ext_pivot ? (pivot ? (double)pivot[c] : 0.) : (double)x[(long)c * P];
        mean = (float)(pv + dm);
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (l == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            if (running_var) {
                const float unbiased = (float)(n > 1. ? var * n / (n - 1.) : var);
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
            }
            if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
        }
    } else {
        mean = running_mean[c];
        invstd = rsqrtf(running_var[c] + eps);
    }
    if (l == 0) bn_coef(g, b, mean, invstd, &scale[c], &shift[c]);
}

// partial[c][s] = (sum g, sum g*xhat) with g = dy * [y > 0 if relu], xhat = (x - mean)*invstd
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ partial, int N, int C, int P, int nsplit,
                                                            int relu, int rev) {
    __shared__ float red[17];
    const int c = blockIdx.x, s = rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;    // rev: walk the image parts downwards
    const float neg = relu == JVAE_ACT_LEAKY ? JVAE_LEAKY_SLOPE : 0.f;       // gradient factor where the pre-activation is <= 0
    const float mu = mean[c], is = invstd[c];
    const float g_ = gamma ? gamma[c] : 1.f, b_ = beta ? beta[c] : 0.f;
    float sc, sh;
    bn_coef(g_, b_, mu, is, &sc, &sh);
    const ImageRange ir = image_range(N, nsplit, s);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    float s1 = 0.f, s2 = 0.f;
    if ((P & 3) == 0) {
        const unsigned cnt = (unsigned)(ne - nb) * (unsigned)(P >> 2);
        const Plane4Idx pi = plane4_idx(nb, C, c, P);
#pragma unroll 2
        for (unsigned i = threadIdx.x; i < cnt; i += 256) {
            const long off = pi(i);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(dy + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = gv[e];
                if (relu && !(fmaf(xv[e], sc, sh) > 0.f)) g *= neg;
                s1 += g; s2 += g * ((xv[e] - mu) * is);
            }
        }
    } else {
        const long cnt = (long)(ne - nb) * P;
        for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long n = nb + i / P, q = i % P;
            const long idx = (n * C + c) * (long)P + q;
            const float xv = x[idx];
            float g = dy[idx];
            if (relu && !(fmaf(xv, sc, sh) > 0.f)) g *= neg;
            s1 += g; s2 += g * ((xv - mu) * is);
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        partial[((long)c * nsplit + s) * 2 + 0] = s1;
        partial[((long)c * nsplit + s) * 2 + 1] = s2;
    }
}

// dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); one block per (channel, image chunk); the block folds the
// partial sums itself and chunk 0 writes (or accumulates) dgamma / dbeta.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ partial, float* __restrict__ dx,
                                                           float* dgamma, float* dbeta, int accumulate,
                                                           int N, int C, int P, int nsplit, int nchunk, int relu,
                                                           const float* __restrict__ gsums, int count_mult, int rev) {
    __shared__ float ms[2];
    const int c = blockIdx.x, j = rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    const float neg = relu == JVAE_ACT_LEAKY ? JVAE_LEAKY_SLOPE : 0.f;
    if (threadIdx.x == 0) {
        double s1 = 0., s2 = 0.;
        for (int s = 0; s < nsplit; ++s) {
            s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
            s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
        }
        const double M = (double)N * P * count_mult;
        // synchronised BatchNorm: the means over ALL ranks come from gsums; dgamma / dbeta stay the LOCAL sums
        ms[0] = (float)((gsums ? (double)gsums[2 * c] : s1) / M);
        ms[1] = (float)((gsums ? (double)gsums[2 * c + 1] : s2) / M);
        if (j == 0) {
            if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
            if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
        }
    }
    __syncthreads();
    const float m1 = ms[0], m2 = ms[1];
    const float mu = mean[c], is = invstd[c];
    const float g_ = gamma ? gamma[c] : 1.f, b_ = beta ? beta[c] : 0.f;
    float sc, sh;
    bn_coef(g_, b_, mu, is, &sc, &sh);
    const float k = g_ * is;
    const ImageRange ir = image_range(N, nchunk, j);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    if ((P & 3) == 0) {
        const unsigned cnt = (unsigned)(ne - nb) * (unsigned)(P >> 2);
        const Plane4Idx pi = plane4_idx(nb, C, c, P);
#pragma unroll 2
        for (unsigned i = threadIdx.x; i < cnt; i += 256) {
            const long off = pi(i);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
            f32x4 gv = *reinterpret_cast<const f32x4*>(dy + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = gv[e];
                if (relu && !(fmaf(xv[e], sc, sh) > 0.f)) g *= neg;
                gv[e] = k * (g - m1 - ((xv[e] - mu) * is) * m2);
            }
            *reinterpret_cast<f32x4*>(dx + off) = gv;
        }
    } else {
        const long cnt = (long)(ne - nb) * P;
        for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long n = nb + i / P, q = i % P;
            const long idx = (n * C + c) * (long)P + q;
            const float xv = x[idx];
            float g = dy[idx];
            if (relu && !(fmaf(xv, sc, sh) > 0.f)) g *= neg;
            dx[idx] = k * (g - m1 - ((xv - mu) * is) * m2);
        }
    }
}

// sums[c] = sum over splits of partial[c][s] (fp64 accumulation, fixed order)
__global__ void bn_fold_kernel(const float* __restrict__ partial, float* __restrict__ sums, int C, int nsplit) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0., s2 = 0.;
    for (int s = 0; s < nsplit; ++s) {
        s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
        s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
    }
    sums[2 * c] = (float)s1;
    sums[2 * c + 1] = (float)s2;
}

inline int pick_split(int N, int C, int P) {
    long work = (long)N * P;
    int s = (int)(work / 8192);
    if (s < 1) s = 1;
    int cap = (2048 + C - 1) / C;          // ~2048 blocks in flight is plenty
    if (s > cap) s = cap;
    if (s > MAX_SPLIT) s = MAX_SPLIT;
    if (s > N) s = N;
    return s < 1 ? 1 : s;
}

// image chunks per channel for the apply kernels: ~16K elements per block, <= 64 chunks
inline int pick_chunk(int N, int C, int P) {
    long work = (long)N * P;
    int s = (int)(work / 16384);
    if (s < 1) s = 1;
    int cap = (4096 + C - 1) / C;
    if (s > cap) s = cap;
    if (s > N) s = N;
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" {

// Host-only (no GPU call): the launch plan of the fp32 BatchNorm kernels for a (N, C, P) tensor - `nsplit` image parts for the
// two reduction kernels, `nchunk` for the two apply kernels - and the image range [nb, ne) that part j of `parts` receives
// (image_range(), the one definition every kernel uses).  tests/test_abi_and_host.py sweeps every batch size with these.
int jvae_bn_plan(int N, int C, int P, int* nsplit, int* nchunk) {
    if (N <= 0 || C <= 0 || P <= 0 || !nsplit || !nchunk) return JVAE_EINVAL;
    *nsplit = pick_split(N, C, P);
    *nchunk = pick_chunk(N, C, P);
    return 0;
}
int jvae_image_range(int N, int parts, int j, int* nb, int* ne) {
    if (N < 0 || parts <= 0 || j < 0 || j >= parts || !nb || !ne) return JVAE_EINVAL;
    const ImageRange r = image_range(N, parts, j);
    *nb = r.nb; *ne = r.ne;
    return 0;
}

// workspace: 2*C*MAX_SPLIT partials + 2*C coefficients
size_t jvae_bn_workspace_bytes(int C) { return sizeof(float) * ((size_t)2 * C * MAX_SPLIT + (size_t)2 * C); }

static int bn_fwd_impl(const float* x, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked,
                       float* y, float* save_mean, float* save_invstd,
                       int N, int C, int P, float momentum, float eps, int training, int relu,
                       const float* ext_stats, int ext_nsplit, const float* ext_pivot, int count_mult,
                       void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || N < 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes(C) || !ws) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const float* partial = (const float*)ws;
    int ns = 1;
    const bool ext = training && ext_stats && ext_nsplit > 0;
    if (ext) {
        partial = ext_stats;
        ns = ext_nsplit;
        if (!save_mean || !save_invstd) return JVAE_EINVAL;
    } else if (training) {
        if (!save_mean || !save_invstd) return JVAE_EINVAL;
        ns = pick_split(N, C, P);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns), dim3(256), 0, st, x, (float*)ws, N, C, P, ns, (const float*)nullptr);
        JVAE_LAUNCH_CHECK();
    } else if (!running_mean || !running_var) {
        return JVAE_EINVAL;
    }
    const int nc = pick_chunk(N, C, P);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(C, nc), dim3(256), 0, st, x, partial, gamma, beta, running_mean, running_var,
                       num_batches_tracked, save_mean, save_invstd, y, N, C, P, ns, nc, momentum, eps, training, jvae_act_kind(relu),
                       ext ? 1 : 0, ext_pivot, count_mult > 0 ? count_mult : 1);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_bn_fwd_f32(const float* x, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* y, float* save_mean, float* save_invstd,
                    int N, int C, int P, float momentum, float eps, int training, int relu,
                    void* ws, size_t ws_bytes, void* stream) {
    return bn_fwd_impl(x, gamma, beta, running_mean, running_var, num_batches_tracked, y, save_mean, save_invstd,
                       N, C, P, momentum, eps, training, relu, nullptr, 0, nullptr, 1, ws, ws_bytes, stream);
}

// Same, with the batch statistics supplied by the producing convolution (jvae_conv2d_fwd_stats_f32):
// ext_stats (C, ext_nsplit, 2) = per-workgroup (sum, sum of squares) of (x - ext_pivot[c]); ext_pivot = the conv bias
// (NULL = 0).  ext_nsplit == 0 falls back to the statistics kernel.
int jvae_bn_fwd_ext_f32(const float* x, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked,
                        float* y, float* save_mean, float* save_invstd,
                        int N, int C, int P, float momentum, float eps, int training, int relu,
                        const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                        void* ws, size_t ws_bytes, void* stream) {
    return bn_fwd_impl(x, gamma, beta, running_mean, running_var, num_batches_tracked, y, save_mean, save_invstd,
                       N, C, P, momentum, eps, training, relu, ext_stats, ext_nsplit, ext_pivot, 1, ws, ws_bytes, stream);
}

// Statistics + coefficients only (see bn_finalize_kernel): scale / shift (C floats each) are what the consuming
// convolution applies to x while loading it (jvae_conv2d_fwd_aff_f32 / jvae_conv2d_wgrad_aff_f32).  The backward pass is
// the ordinary jvae_bn_bwd_f32 on (dy = gradient w.r.t. the normalised activation, x).
int jvae_bn_finalize_f32(const float* x, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked,
                         float* save_mean, float* save_invstd, float* scale, float* shift,
                         int N, int C, int P, float momentum, float eps, int training,
                         const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                         void* ws, size_t ws_bytes, void* stream) {
    if (!x || !scale || !shift || N <= 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const float* partial = (const float*)ws;
    int ns = 1;
    const bool ext = training && ext_stats && ext_nsplit > 0;
    if (training && (!save_mean || !save_invstd)) return JVAE_EINVAL;
    if (ext) {
        partial = ext_stats;
        ns = ext_nsplit;
    } else if (training) {
        ns = pick_split(N, C, P);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns), dim3(256), 0, st, x, (float*)ws, N, C, P, ns, (const float*)nullptr);
        JVAE_LAUNCH_CHECK();
    } else if (!running_mean || !running_var) {
        return JVAE_EINVAL;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, st, x, partial, gamma, beta, running_mean, running_var,
                       num_batches_tracked, save_mean, save_invstd, scale, shift, N, C, P, ns, momentum, eps, training,
                       ext ? 1 : 0, ext_pivot);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_bn_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd,
                    float* dx, float* dgamma, float* dbeta, int accumulate,
                    int N, int C, int P, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !dx || N < 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes(C) || !ws) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)ws;
    const int ns = pick_split(N, C, P);
    // Traversal order against the 256 MB Infinity Cache: the reduction walks the image parts DOWNWARDS - it starts on the part of dy
    // that the producing dgrad kernel wrote last - and the apply pass walks them upwards, i.e. starts on what the reduction
    // read last: 113-118 us instead of 120-124 for the 134 MB activation (two 268 MB sweeps), about 1 % of the step.
    // (order bit 0: reduce downwards, bit 1: apply downwards; 0 and 3 measured alike, profiles/NOTES.md round 4)
    const int order = 1;
    relu = jvae_act_kind(relu);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, ns), dim3(256), 0, st, dy, x, gamma, beta, save_mean, save_invstd,
                       partial, N, C, P, ns, relu, order & 1);
    JVAE_LAUNCH_CHECK();
    const int nc = pick_chunk(N, C, P);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(C, nc), dim3(256), 0, st, dy, x, gamma, beta, save_mean, save_invstd,
                       partial, dx, dgamma, dbeta, accumulate, N, C, P, ns, nc, relu, (const float*)nullptr, 1, (order >> 1) & 1);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// ---- synchronised BatchNorm (data-parallel ranks share the batch statistics) --------------------------
// The host all-reduces the (C,2) sums between the two calls of each direction.

// sums[c] = (sum(x - pivot[c]), sum((x - pivot[c])^2)) over this rank's batch; pivot: (C) identical on every rank
int jvae_bn_sums_f32(const float* x, const float* pivot, float* sums, int N, int C, int P,
                     void* ws, size_t ws_bytes, void* stream) {
    if (!x || !pivot || !sums || N < 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int ns = N > 0 ? pick_split(N, C, P) : 1;
    if (N > 0) {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns), dim3(256), 0, st, x, (float*)ws, N, C, P, ns, pivot);
        JVAE_LAUNCH_CHECK();
    } else {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * 2 * (size_t)C, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, (const float*)ws, sums, C, ns);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// forward apply from all-reduced sums: mean = pivot + S1/n, var = S2/n - (S1/n)^2 with n = N*P*world
int jvae_bn_fwd_sync_f32(const float* x, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked,
                         float* y, float* save_mean, float* save_invstd,
                         int N, int C, int P, float momentum, float eps, int relu,
                         const float* global_sums, const float* pivot, int world,
                         void* ws, size_t ws_bytes, void* stream) {
    if (!global_sums || !pivot || world < 1) return JVAE_EINVAL;
    return bn_fwd_impl(x, gamma, beta, running_mean, running_var, num_batches_tracked, y, save_mean, save_invstd,
                       N, C, P, momentum, eps, 1, relu, global_sums, 1, pivot, world, ws, ws_bytes, stream);
}

// local_sums[c] = (sum g, sum g*xhat) of this rank (g = dy masked by the fused ReLU)
int jvae_bn_bwd_sums_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, float* local_sums,
                         int N, int C, int P, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !local_sums || N < 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int ns = N > 0 ? pick_split(N, C, P) : 1;
    if (N > 0) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, ns), dim3(256), 0, st, dy, x, gamma, beta, save_mean, save_invstd,
                           (float*)ws, N, C, P, ns, jvae_act_kind(relu), 0);
        JVAE_LAUNCH_CHECK();
    } else {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * 2 * (size_t)C, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, (const float*)ws, local_sums, C, ns);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// dx from the GLOBAL means (global_sums / (N*P*world)); dgamma / dbeta (+)= the LOCAL sums (the gradient all-reduce
// of the optimiser averages them afterwards, as DDP + SyncBatchNorm does)
int jvae_bn_bwd_sync_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd,
                         const float* local_sums, const float* global_sums, int world,
                         float* dx, float* dgamma, float* dbeta, int accumulate,
                         int N, int C, int P, int relu, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !dx || !local_sums || !global_sums || world < 1) return JVAE_EINVAL;
    if (N < 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    if (N == 0) return 0;
    const int nc = pick_chunk(N, C, P);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(C, nc), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, beta, save_mean,
                       save_invstd, local_sums, dx, dgamma, dbeta, accumulate, N, C, P, 1, nc, jvae_act_kind(relu), global_sums, world, 0);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
