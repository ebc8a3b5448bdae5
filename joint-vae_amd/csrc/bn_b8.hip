// BatchNorm2d (+ fused ReLU) on bf16 "B8" activations (layout: conv_b8.hip): forward (train / eval), backward.
// Statistics, coefficients, parameter gradients and running statistics are fp32 (partials folded in fp64); only the
// activation tensors are bf16.  HBM-bound: each pass streams the tensor once in 16-byte units (8 channels of a pixel),
// a block owns one channel block (8 channels) x a chunk of images.
// Reference: nn.BatchNorm2d + ReLU after every (de)conv (module/vae_layers/conv.py:214-220).
#include "common.h"
#include "jvae_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int MAX_SPLIT = 512;     // partial sums per channel (few channel blocks -> many image splits to fill the chip)

// unit offset of flat index i (image-major inside this block's image range) without 64-bit division
struct UnitIdx {
    unsigned hw; int sh; long stride, base;     // sh >= 0: HW is a power of two
    __device__ __forceinline__ long operator()(unsigned i) const {
        unsigned n, q;
        if (sh >= 0) { n = i >> sh; q = i & (hw - 1); } else { n = i / hw; q = i - n * hw; }
        return base + (long)n * stride + q;
    }
};
__device__ __forceinline__ UnitIdx unit_idx(int nb, int CB, int cb, long HW) {
    UnitIdx u;
    u.hw = (unsigned)HW;
    u.sh = (HW & (HW - 1)) == 0 ? __ffsll((long long)HW) - 1 : -1;
    u.stride = (long)CB * HW;
    u.base = ((long)nb * CB + cb) * HW;
    return u;
}

__device__ __forceinline__ void bn_coef(float g, float b, float mean, float invstd, float* sc, float* sh) {
    *sc = g * invstd;
    *sh = b - mean * (g * invstd);
}

// reduce 8 per-thread values over the block (256 threads): result[ci] valid in thread 0
__device__ __forceinline__ void block_sum8(float (&v)[8], float (*red)[8]) {
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const float w = wave_sum(v[ci]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][ci] = w;
    }
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) v[ci] = (red[0][ci] + red[1][ci]) + (red[2][ci] + red[3][ci]);
    __syncthreads();
}

// partial[c][s] = (sum(x - p), sum((x - p)^2)) over the images of split s;  p = x[0][c][0]
__global__ __launch_bounds__(256) void bn8_stats_kernel(const bf16x8* __restrict__ x, float* __restrict__ partial,
                                                        int N, int C, int CB, long HW, int nsplit,
                                                        const float* __restrict__ pivot) {
    __shared__ float red[4][8];
    const int cb = blockIdx.x, s = blockIdx.y;
    bf16x8 pv = x[(long)cb * HW];
    float pf[8];            // pivot (C) given: the same on every data-parallel rank (synchronised statistics)
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) pf[ci] = pivot ? (cb * 8 + ci < C ? pivot[cb * 8 + ci] : 0.f) : (float)pv[ci];
    const ImageRange ir = image_range(N, nsplit, s);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    float s1[8], s2[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) s1[ci] = s2[ci] = 0.f;
    const unsigned cnt = (unsigned)((long)(ne - nb) * HW);
    const UnitIdx ui = unit_idx(nb, CB, cb, HW);
#pragma unroll 2
    for (unsigned i = threadIdx.x; i < cnt; i += 256) {
        const bf16x8 v = x[ui(i)];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) { const float d = (float)v[ci] - pf[ci]; s1[ci] += d; s2[ci] += d * d; }
    }
    block_sum8(s1, red);
    block_sum8(s2, red);
    if (threadIdx.x == 0)
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = cb * 8 + ci;
            if (c < C) { partial[((long)c * nsplit + s) * 2 + 0] = s1[ci]; partial[((long)c * nsplit + s) * 2 + 1] = s2[ci]; }
        }
}

// coefficients of one channel per block: folds the partial sums (fp64, fixed order), publishes mean / invstd, updates
// the running statistics and writes coef[c] = scale, coef[C8 + c] = shift (0 for padding channels)
__global__ __launch_bounds__(64) void bn8_finalize_kernel(const bf16x8* __restrict__ x, const float* __restrict__ partial,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, long long* num_batches_tracked,
                                                          float* save_mean, float* save_invstd, float* __restrict__ coef,
                                                          int N, int C, int C8, long HW, int nsplit, float momentum, float eps,
                                                          int training, int ext_pivot, const float* __restrict__ pivot) {
    const int c = blockIdx.x, l = threadIdx.x;
    float sc = 0.f, sh = 0.f;
    if (c < C) {
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        if (training) {
            double s1 = 0., s2 = 0.;
            for (int s = l; s < nsplit; s += 64) {
                s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
                s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            const double n = (double)N * HW;
            const double dm = s1 / n;
            double var = s2 / n - dm * dm;
            if (var < 0.) var = 0.;
            const double pv = ext_pivot ? (pivot ? (double)pivot[c] : 0.) : (double)(float)x[(long)(c >> 3) * HW][c & 7];
            const float mean = (float)(pv + dm);
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
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
            bn_coef(g, b, mean, invstd, &sc, &sh);
        } else {
            bn_coef(g, b, running_mean[c], rsqrtf(running_var[c] + eps), &sc, &sh);
        }
    }
    if (l == 0) { coef[c] = sc; coef[C8 + c] = sh; }
}

// y = [relu](x*scale + shift) in bf16; block (cb, chunk j)
__global__ __launch_bounds__(256) void bn8_apply_kernel(const bf16x8* __restrict__ x, const float* __restrict__ coef,
                                                        bf16x8* __restrict__ y, int N, int CB, long HW, int nchunk, int relu) {
    const int cb = blockIdx.x, j = blockIdx.y;
    const int C8 = CB * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) { sc[ci] = coef[cb * 8 + ci]; sh[ci] = coef[C8 + cb * 8 + ci]; }
    const ImageRange ir = image_range(N, nchunk, j);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    const unsigned cnt = (unsigned)((long)(ne - nb) * HW);
    const UnitIdx ui = unit_idx(nb, CB, cb, HW);
#pragma unroll 2
    for (unsigned i = threadIdx.x; i < cnt; i += 256) {
        const long off = ui(i);
        const bf16x8 v = x[off];
        bf16x8 o;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float t = fmaf((float)v[ci], sc[ci], sh[ci]);          // padding channels: sc = sh = 0 -> stay 0
            o[ci] = (__bf16)(relu ? fmaxf(t, 0.f) : t);
        }
        y[off] = o;
    }
}

// partial[c][s] = (sum g, sum g*xhat), g = dy masked by the recomputed ReLU
__global__ __launch_bounds__(256) void bn8_bwd_reduce_kernel(const bf16x8* __restrict__ dy, const bf16x8* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             float* __restrict__ partial, int N, int C, int CB, long HW,
                                                             int nsplit, int relu) {
    __shared__ float red[4][8];
    const int cb = blockIdx.x, s = blockIdx.y;
    float mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const int c = cb * 8 + ci;
        mu[ci] = c < C ? mean[c] : 0.f;
        is[ci] = c < C ? invstd[c] : 0.f;
        bn_coef(c < C && gamma ? gamma[c] : (c < C ? 1.f : 0.f), c < C && beta ? beta[c] : 0.f, mu[ci], is[ci], &sc[ci], &sh[ci]);
    }
    const ImageRange ir = image_range(N, nsplit, s);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    float s1[8], s2[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) s1[ci] = s2[ci] = 0.f;
    const unsigned cnt = (unsigned)((long)(ne - nb) * HW);
    const UnitIdx ui = unit_idx(nb, CB, cb, HW);
#pragma unroll 2
    for (unsigned i = threadIdx.x; i < cnt; i += 256) {
        const long off = ui(i);
        const bf16x8 xv = x[off], gv = dy[off];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float xf = (float)xv[ci];
            float g = (float)gv[ci];
            if (relu && !(fmaf(xf, sc[ci], sh[ci]) > 0.f)) g = 0.f;
            s1[ci] += g; s2[ci] += g * ((xf - mu[ci]) * is[ci]);
        }
    }
    block_sum8(s1, red);
    block_sum8(s2, red);
    if (threadIdx.x == 0)
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = cb * 8 + ci;
            if (c < C) { partial[((long)c * nsplit + s) * 2 + 0] = s1[ci]; partial[((long)c * nsplit + s) * 2 + 1] = s2[ci]; }
        }
}

// per channel: means of (g, g*xhat) over the batch -> coef[c], coef[C8 + c]; dgamma / dbeta (+)= the sums
__global__ __launch_bounds__(64) void bn8_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ coef,
                                                              float* dgamma, float* dbeta, int accumulate,
                                                              int N, int C, int C8, long HW, int nsplit) {
    const int c = blockIdx.x, l = threadIdx.x;
    double s1 = 0., s2 = 0.;
    if (c < C)
        for (int s = l; s < nsplit; s += 64) {
            s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
            s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (l == 0) {
        const double M = (double)N * HW;
        coef[c] = (float)(s1 / M);
        coef[C8 + c] = (float)(s2 / M);
        if (c < C) {
            if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
            if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
        }
    }
}

// dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) in bf16
__global__ __launch_bounds__(256) void bn8_bwd_apply_kernel(const bf16x8* __restrict__ dy, const bf16x8* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ coef, bf16x8* __restrict__ dx,
                                                            int N, int C, int CB, long HW, int nchunk, int relu) {
    const int cb = blockIdx.x, j = blockIdx.y;
    const int C8 = CB * 8;
    float mu[8], is[8], sc[8], sh[8], k[8], m1[8], m2[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const int c = cb * 8 + ci;
        const float g_ = c < C ? (gamma ? gamma[c] : 1.f) : 0.f;
        mu[ci] = c < C ? mean[c] : 0.f;
        is[ci] = c < C ? invstd[c] : 0.f;
        bn_coef(g_, c < C && beta ? beta[c] : 0.f, mu[ci], is[ci], &sc[ci], &sh[ci]);
        k[ci] = g_ * is[ci];
        m1[ci] = coef[c]; m2[ci] = coef[C8 + c];
    }
    const ImageRange ir = image_range(N, nchunk, j);   // trailing parts may be empty, never negative
    const int nb = ir.nb, ne = ir.ne;
    const unsigned cnt = (unsigned)((long)(ne - nb) * HW);
    const UnitIdx ui = unit_idx(nb, CB, cb, HW);
#pragma unroll 2
    for (unsigned i = threadIdx.x; i < cnt; i += 256) {
        const long off = ui(i);
        const bf16x8 xv = x[off], gv = dy[off];
        bf16x8 o;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float xf = (float)xv[ci];
            float g = (float)gv[ci];
            if (relu && !(fmaf(xf, sc[ci], sh[ci]) > 0.f)) g = 0.f;
            o[ci] = (__bf16)(k[ci] * (g - m1[ci] - ((xf - mu[ci]) * is[ci]) * m2[ci]));   // k = 0 on padding channels
        }
        dx[off] = o;
    }
}

// sums[c] = fold of partial[c][0..nsplit) (fp64, fixed order): the (C,2) block the ranks all-reduce
__global__ __launch_bounds__(64) void bn8_fold_kernel(const float* __restrict__ partial, float* __restrict__ sums, int nsplit) {
    const int c = blockIdx.x, l = threadIdx.x;
    double s1 = 0., s2 = 0.;
    for (int s = l; s < nsplit; s += 64) {
        s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
        s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (l == 0) { sums[2 * c] = (float)s1; sums[2 * c + 1] = (float)s2; }
}

// synchronised backward: means of (g, g*xhat) over ALL ranks from the all-reduced sums; dgamma / dbeta stay the local sums
__global__ void bn8_bwd_sync_coef_kernel(const float* __restrict__ local, const float* __restrict__ global,
                                         float* __restrict__ coef, float* dgamma, float* dbeta, int accumulate,
                                         double count, int C, int C8) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C8) return;
    coef[c] = c < C ? (float)((double)global[2 * c] / count) : 0.f;
    coef[C8 + c] = c < C ? (float)((double)global[2 * c + 1] / count) : 0.f;
    if (c < C) {
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + local[2 * c];
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + local[2 * c + 1];
    }
}

// y = relu(x) / dx = dy * [y > 0] on B8 tensors (layers without BatchNorm)
__global__ __launch_bounds__(256) void relu8_fwd_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 v = x[i];
        bf16x8 o;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) o[ci] = (float)v[ci] > 0.f ? v[ci] : (__bf16)0.f;
        y[i] = o;
    }
}
__global__ __launch_bounds__(256) void relu8_bwd_kernel(const bf16x8* __restrict__ dy, const bf16x8* __restrict__ y,
                                                        bf16x8* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 g = dy[i], v = y[i];
        bf16x8 o;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) o[ci] = (float)v[ci] > 0.f ? g[ci] : (__bf16)0.f;
        dx[i] = o;
    }
}

inline int pick_split(int N, int CB, long HW) {
    long s = (long)N * HW / 2048;
    if (s < 1) s = 1;
    long cap = (2048 + CB - 1) / CB;
    if (s > cap) s = cap;
    if (s > MAX_SPLIT) s = MAX_SPLIT;
    if (s > N) s = N;
    return (int)(s < 1 ? 1 : s);
}
inline int pick_chunk(int N, int CB, long HW) {
    long s = (long)N * HW / 4096;
    if (s < 1) s = 1;
    long cap = (4096 + CB - 1) / CB;
    if (s > cap) s = cap;
    if (s > N) s = N;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" {

// Host-only launch plan of the bf16 (B8 layout) BatchNorm kernels: see jvae_bn_plan.
int jvae_bn_plan_b8(int N, int C, long HW, int* nsplit, int* nchunk) {
    if (N <= 0 || C <= 0 || HW <= 0 || !nsplit || !nchunk) return JVAE_EINVAL;
    const int CB = (C + 7) / 8;
    *nsplit = pick_split(N, CB, HW);
    *nchunk = pick_chunk(N, CB, HW);
    return 0;
}

// partial sums (2 * C8 * MAX_SPLIT floats) followed by 2 * C8 per-channel coefficients
size_t jvae_bn_workspace_bytes_b8(int C) { return sizeof(float) * ((size_t)2 * ((C + 7) / 8 * 8) * (MAX_SPLIT + 1)); }

// x, y: B8 (N, ceil(C/8), HW, 8).  Workspace: jvae_bn_workspace_bytes_b8(C).  ext_stats as in jvae_bn_fwd_ext_f32
// (ext_nsplit == 0: the statistics kernel runs here).
int jvae_bn_fwd_b8(const void* x, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, long long* num_batches_tracked,
                   void* y, float* save_mean, float* save_invstd,
                   int N, int C, long HW, float momentum, float eps, int training, int relu,
                   const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                   void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || N < 0 || C <= 0 || HW <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes_b8(C) || !ws) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int CB = (C + 7) / 8;
    const float* partial = (const float*)ws;
    int ns = 1;
    const bool ext = training && ext_stats && ext_nsplit > 0;
    if (training && (!save_mean || !save_invstd)) return JVAE_EINVAL;
    if (ext) {
        partial = ext_stats;
        ns = ext_nsplit;
    } else if (training) {
        ns = pick_split(N, CB, HW);
        hipLaunchKernelGGL(bn8_stats_kernel, dim3(CB, ns), dim3(256), 0, st, (const bf16x8*)x, (float*)ws, N, C, CB, HW, ns, (const float*)nullptr);
        JVAE_LAUNCH_CHECK();
    } else if (!running_mean || !running_var) {
        return JVAE_EINVAL;
    }
    float* coef = (float*)ws + (size_t)2 * CB * 8 * MAX_SPLIT;
    hipLaunchKernelGGL(bn8_finalize_kernel, dim3(CB * 8), dim3(64), 0, st, (const bf16x8*)x, partial, gamma, beta,
                       running_mean, running_var, num_batches_tracked, save_mean, save_invstd, coef, N, C, CB * 8, HW, ns,
                       momentum, eps, training, ext ? 1 : 0, ext_pivot);
    JVAE_LAUNCH_CHECK();
    const int nc = pick_chunk(N, CB, HW);
    hipLaunchKernelGGL(bn8_apply_kernel, dim3(CB, nc), dim3(256), 0, st, (const bf16x8*)x, (const float*)coef, (bf16x8*)y,
                       N, CB, HW, nc, relu);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// Statistics + coefficients only (the consuming bf16 convolution normalises its input while staging it, see InAff):
// coef = (2, C8) floats, C8 = ceil(C/8)*8: scale then shift, zero for the padding channels.
int jvae_bn_finalize_b8(const void* x, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked,
                        float* save_mean, float* save_invstd, float* coef,
                        int N, int C, long HW, float momentum, float eps, int training,
                        const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                        void* ws, size_t ws_bytes, void* stream) {
    if (!x || !coef || N <= 0 || C <= 0 || HW <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes_b8(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int CB = (C + 7) / 8;
    const float* partial = (const float*)ws;
    int ns = 1;
    const bool ext = training && ext_stats && ext_nsplit > 0;
    if (training && (!save_mean || !save_invstd)) return JVAE_EINVAL;
    if (ext) {
        partial = ext_stats;
        ns = ext_nsplit;
    } else if (training) {
        ns = pick_split(N, CB, HW);
        hipLaunchKernelGGL(bn8_stats_kernel, dim3(CB, ns), dim3(256), 0, st, (const bf16x8*)x, (float*)ws, N, C, CB, HW, ns, (const float*)nullptr);
        JVAE_LAUNCH_CHECK();
    } else if (!running_mean || !running_var) {
        return JVAE_EINVAL;
    }
    hipLaunchKernelGGL(bn8_finalize_kernel, dim3(CB * 8), dim3(64), 0, st, (const bf16x8*)x, partial, gamma, beta,
                       running_mean, running_var, num_batches_tracked, save_mean, save_invstd, coef, N, C, CB * 8, HW, ns,
                       momentum, eps, training, ext ? 1 : 0, ext_pivot);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_bn_bwd_b8(const void* dy, const void* x, const float* gamma, const float* beta,
                   const float* save_mean, const float* save_invstd,
                   void* dx, float* dgamma, float* dbeta, int accumulate,
                   int N, int C, long HW, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !dx || N < 0 || C <= 0 || HW <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes_b8(C) || !ws) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int CB = (C + 7) / 8;
    float* partial = (float*)ws;
    const int ns = pick_split(N, CB, HW);
    hipLaunchKernelGGL(bn8_bwd_reduce_kernel, dim3(CB, ns), dim3(256), 0, st, (const bf16x8*)dy, (const bf16x8*)x, gamma, beta,
                       save_mean, save_invstd, partial, N, C, CB, HW, ns, relu);
    JVAE_LAUNCH_CHECK();
    float* coef = partial + (size_t)2 * CB * 8 * MAX_SPLIT;
    hipLaunchKernelGGL(bn8_bwd_finalize_kernel, dim3(CB * 8), dim3(64), 0, st, (const float*)partial, coef, dgamma, dbeta,
                       accumulate, N, C, CB * 8, HW, ns);
    JVAE_LAUNCH_CHECK();
    const int nc = pick_chunk(N, CB, HW);
    hipLaunchKernelGGL(bn8_bwd_apply_kernel, dim3(CB, nc), dim3(256), 0, st, (const bf16x8*)dy, (const bf16x8*)x, gamma, beta,
                       save_mean, save_invstd, (const float*)coef, (bf16x8*)dx, N, C, CB, HW, nc, relu);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// ReLU on B8 tensors of `units` 16-byte units (backward takes the forward OUTPUT)
// ---- synchronised BatchNorm on B8 tensors (data-parallel ranks share the batch statistics) ----------------
// Same protocol as the fp32 entry points (bn.hip): the host all-reduces the (C,2) sums between the two calls of each
// direction.  Statistics fp32, activations bf16.

// sums[c] = (sum(x - pivot[c]), sum((x - pivot[c])^2)) over this rank's batch; pivot: (C) identical on every rank
int jvae_bn_sums_b8(const void* x, const float* pivot, float* sums, int N, int C, long HW,
                    void* ws, size_t ws_bytes, void* stream) {
    if (!x || !pivot || !sums || N <= 0 || C <= 0 || HW <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes_b8(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int CB = (C + 7) / 8;
    const int ns = pick_split(N, CB, HW);
    hipLaunchKernelGGL(bn8_stats_kernel, dim3(CB, ns), dim3(256), 0, st, (const bf16x8*)x, (float*)ws, N, C, CB, HW, ns, pivot);
    JVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn8_fold_kernel, dim3(C), dim3(64), 0, st, (const float*)ws, sums, ns);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// forward from the all-reduced sums: mean = pivot + S1/n, var = S2/n - (S1/n)^2 with n = N*HW*world; y = [relu](bn(x)) in bf16
int jvae_bn_fwd_sync_b8(const void* x, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked,
                        void* y, float* save_mean, float* save_invstd,
                        int N, int C, long HW, float momentum, float eps, int relu,
                        const float* global_sums, const float* pivot, int world,
                        void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || !global_sums || !pivot || !save_mean || !save_invstd || world < 1 || N <= 0 || C <= 0 || HW <= 0)
        return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes_b8(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int CB = (C + 7) / 8;
    float* coef = (float*)ws + (size_t)2 * CB * 8 * MAX_SPLIT;
    // one "split" holding the global sums, N*world images behind them
    hipLaunchKernelGGL(bn8_finalize_kernel, dim3(CB * 8), dim3(64), 0, st, (const bf16x8*)x, global_sums, gamma, beta,
                       running_mean, running_var, num_batches_tracked, save_mean, save_invstd, coef, N * world, C, CB * 8, HW, 1,
                       momentum, eps, 1, 1, pivot);
    JVAE_LAUNCH_CHECK();
    const int nc = pick_chunk(N, CB, HW);
    hipLaunchKernelGGL(bn8_apply_kernel, dim3(CB, nc), dim3(256), 0, st, (const bf16x8*)x, (const float*)coef, (bf16x8*)y,
                       N, CB, HW, nc, relu);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// local_sums[c] = (sum g, sum g*xhat) of this rank (g = dy masked by the fused ReLU)
int jvae_bn_bwd_sums_b8(const void* dy, const void* x, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, float* local_sums,
                        int N, int C, long HW, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !local_sums || N <= 0 || C <= 0 || HW <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes_b8(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int CB = (C + 7) / 8;
    const int ns = pick_split(N, CB, HW);
    hipLaunchKernelGGL(bn8_bwd_reduce_kernel, dim3(CB, ns), dim3(256), 0, st, (const bf16x8*)dy, (const bf16x8*)x, gamma, beta,
                       save_mean, save_invstd, (float*)ws, N, C, CB, HW, ns, relu);
    JVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn8_fold_kernel, dim3(C), dim3(64), 0, st, (const float*)ws, local_sums, ns);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// dx from the all-reduced sums (means over N*HW*world elements); dgamma / dbeta (+)= the LOCAL sums (the gradient
// all-reduce averages them like every other parameter gradient)
int jvae_bn_bwd_sync_b8(const void* dy, const void* x, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd,
                        const float* local_sums, const float* global_sums, int world,
                        void* dx, float* dgamma, float* dbeta, int accumulate,
                        int N, int C, long HW, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !local_sums || !global_sums || !dx || world < 1 || N <= 0 || C <= 0 || HW <= 0)
        return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes_b8(C) || !ws) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int CB = (C + 7) / 8;
    float* coef = (float*)ws + (size_t)2 * CB * 8 * MAX_SPLIT;
    hipLaunchKernelGGL(bn8_bwd_sync_coef_kernel, dim3(cdiv(CB * 8, 64)), dim3(64), 0, st, local_sums, global_sums, coef, dgamma, dbeta,
                       accumulate, (double)N * (double)HW * world, C, CB * 8);
    JVAE_LAUNCH_CHECK();
    const int nc = pick_chunk(N, CB, HW);
    hipLaunchKernelGGL(bn8_bwd_apply_kernel, dim3(CB, nc), dim3(256), 0, st, (const bf16x8*)dy, (const bf16x8*)x, gamma, beta,
                       save_mean, save_invstd, (const float*)coef, (bf16x8*)dx, N, C, CB, HW, nc, relu);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_relu_fwd_b8(const void* x, void* y, long units, void* stream) {
    if (!x || !y || units < 0) return JVAE_EINVAL;
    if (units == 0) return 0;
    const int blocks = (int)((units + 255) / 256 > 8192 ? 8192 : (units + 255) / 256);
    hipLaunchKernelGGL(relu8_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, (bf16x8*)y, units);
    JVAE_LAUNCH_CHECK();
    return 0;
}
int jvae_relu_bwd_b8(const void* dy, const void* y, void* dx, long units, void* stream) {
    if (!dy || !y || !dx || units < 0) return JVAE_EINVAL;
    if (units == 0) return 0;
    const int blocks = (int)((units + 255) / 256 > 8192 ? 8192 : (units + 255) / 256);
    hipLaunchKernelGGL(relu8_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)dy, (const bf16x8*)y,
                       (bf16x8*)dx, units);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
