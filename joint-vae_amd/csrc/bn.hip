// Training-mode BatchNorm2d (+ fused ReLU) for NCHW fp32: forward, backward, running-stat update.
// HBM-bound: every pass streams the activation once with 16-byte accesses.
//
// Reference: nn.BatchNorm2d(eps 1e-5, momentum 0.1, affine) inserted after every (de)conv by
// build_de_conv_layers (module/vae_layers/conv.py:214-220) followed by the activation (ReLU inplace).
// Forward statistics are the biased batch variance; running_var receives the unbiased one.
//
// Numerics: per-channel sums are taken on data shifted by the channel's first element (robust against
// |mean| >> std), per-block partials are combined in fp64 by a one-block finalize kernel.
#include "common.h"
#include "jvae_internal.h"

namespace {

constexpr int MAX_SPLIT = 64;

// y = fmaf(x, scale, shift): one definition so that backward re-derives the forward's ReLU mask bit-exactly
__device__ __forceinline__ void bn_coef(float g, float b, float mean, float invstd, float* sc, float* sh) {
    *sc = g * invstd;
    *sh = b - mean * (g * invstd);
}

// partial[c][s] = (sum(x-p), sum((x-p)^2)) over the images of split s;  p = x[0][c][0]
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                       int N, int C, int P, int nsplit) {
    __shared__ float red[17];
    const int c = blockIdx.x, s = blockIdx.y;
    const float pivot = x[(long)c * P];
    const int per = (N + nsplit - 1) / nsplit;
    const int nb = s * per, ne = min(N, nb + per);
    float s1 = 0.f, s2 = 0.f;
    if ((P & 3) == 0) {
        const int P4 = P >> 2;
        const long cnt = (long)(ne - nb) * P4;
        for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
            const long n = nb + i / P4, q = i % P4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * C + c) * (long)P) + q * 4);
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

// coef[c] = (scale, shift) with y = x*scale + shift;  updates running stats;  saves mean / invstd.
__global__ void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* num_batches_tracked,
                                   float* save_mean, float* save_invstd, float* coef,
                                   int N, int C, int P, int nsplit, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C) return;
    double s1 = 0., s2 = 0.;
    for (int s = 0; s < nsplit; ++s) {
        s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
        s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
    }
    const double n = (double)N * P;
    const double dm = s1 / n;
    double var = s2 / n - dm * dm;
    if (var < 0.) var = 0.;
    const float mean = (float)((double)x[(long)c * P] + dm);
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    bn_coef(g, b, mean, invstd, &coef[2 * c + 0], &coef[2 * c + 1]);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
        const float unbiased = (float)(n > 1. ? var * n / (n - 1.) : var);
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// eval mode: coefficients from the running statistics
__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    float* coef, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = rsqrtf(running_var[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    bn_coef(g, b, running_mean[c], invstd, &coef[2 * c + 0], &coef[2 * c + 1]);
}

// y = x*scale[c] + shift[c]  (optionally ReLU)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                       float* __restrict__ y, long total, int C, int P, int relu) {
    if ((P & 3) == 0) {
        const long t4 = total >> 2;
        const int P4 = P >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t4; i += (long)gridDim.x * blockDim.x) {
            const int c = (int)((i / P4) % C);
            const float sc = coef[2 * c], sh = coef[2 * c + 1];
            f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) { float t = fmaf(v[j], sc, sh); v[j] = relu ? fmaxf(t, 0.f) : t; }
            reinterpret_cast<f32x4*>(y)[i] = v;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int c = (int)((i / P) % C);
            float t = fmaf(x[i], coef[2 * c], coef[2 * c + 1]);
            y[i] = relu ? fmaxf(t, 0.f) : t;
        }
    }
}

// partial[c][s] = (sum g, sum g*xhat) with g = dy * [y > 0 if relu], xhat = (x - mean)*invstd
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ partial, int N, int C, int P, int nsplit,
                                                            int relu) {
    __shared__ float red[17];
    const int c = blockIdx.x, s = blockIdx.y;
    const float mu = mean[c], is = invstd[c];
    const float g_ = gamma ? gamma[c] : 1.f, b_ = beta ? beta[c] : 0.f;
    float sc, sh;
    bn_coef(g_, b_, mu, is, &sc, &sh);
    const int per = (N + nsplit - 1) / nsplit;
    const int nb = s * per, ne = min(N, nb + per);
    float s1 = 0.f, s2 = 0.f;
    const long cnt = (long)(ne - nb) * P;
    for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
        const long n = nb + i / P, q = i % P;
        const long idx = (n * C + c) * (long)P + q;
        const float xv = x[idx];
        const float xh = (xv - mu) * is;
        float g = dy[idx];
        if (relu && !(fmaf(xv, sc, sh) > 0.f)) g = 0.f;
        s1 += g; s2 += g * xh;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        partial[((long)c * nsplit + s) * 2 + 0] = s1;
        partial[((long)c * nsplit + s) * 2 + 1] = s2;
    }
}

// sums[c] = (sum g / M, sum g*xhat / M); dgamma / dbeta written (or accumulated)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ sums,
                                       float* dgamma, float* dbeta, int accumulate, int N, int C, int P, int nsplit) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0., s2 = 0.;
    for (int s = 0; s < nsplit; ++s) {
        s1 += (double)partial[((long)c * nsplit + s) * 2 + 0];
        s2 += (double)partial[((long)c * nsplit + s) * 2 + 1];
    }
    const double M = (double)N * P;
    sums[2 * c + 0] = (float)(s1 / M);
    sums[2 * c + 1] = (float)(s2 / M);
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
}

// dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ sums, float* __restrict__ dx,
                                                           long total, int C, int P, int relu) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / P) % C);
        const float is = invstd[c];
        const float g_ = gamma ? gamma[c] : 1.f, b_ = beta ? beta[c] : 0.f;
        float sc, sh;
        bn_coef(g_, b_, mean[c], is, &sc, &sh);
        const float xv = x[i];
        const float xh = (xv - mean[c]) * is;
        float g = dy[i];
        if (relu && !(fmaf(xv, sc, sh) > 0.f)) g = 0.f;
        dx[i] = g_ * is * (g - sums[2 * c] - xh * sums[2 * c + 1]);
    }
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

inline int ew_grid(long total4) {
    long b = (total4 + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

// workspace: 2*C*MAX_SPLIT partials + 2*C coefficients
size_t jvae_bn_workspace_bytes(int C) { return sizeof(float) * ((size_t)2 * C * MAX_SPLIT + (size_t)2 * C); }

int jvae_bn_fwd_f32(const float* x, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* y, float* save_mean, float* save_invstd,
                    int N, int C, int P, float momentum, float eps, int training, int relu,
                    void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || N < 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    if (ws_bytes < jvae_bn_workspace_bytes(C) || !ws) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)ws;
    float* coef = partial + (size_t)2 * C * MAX_SPLIT;
    if (training) {
        if (!save_mean || !save_invstd) return JVAE_EINVAL;
        const int ns = pick_split(N, C, P);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns), dim3(256), 0, st, x, partial, N, C, P, ns);
        JVAE_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, x, partial, gamma, beta,
                           running_mean, running_var, num_batches_tracked, save_mean, save_invstd, coef,
                           N, C, P, ns, momentum, eps);
        JVAE_LAUNCH_CHECK();
    } else {
        if (!running_mean || !running_var) return JVAE_EINVAL;
        hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, gamma, beta, running_mean,
                           running_var, coef, C, eps);
        JVAE_LAUNCH_CHECK();
    }
    const long total = (long)N * C * P;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total / 4 + 1)), dim3(256), 0, st, x, coef, y, total, C, P, relu);
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
    float* sums = partial + (size_t)2 * C * MAX_SPLIT;
    const int ns = pick_split(N, C, P);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, ns), dim3(256), 0, st, dy, x, gamma, beta, save_mean, save_invstd,
                       partial, N, C, P, ns, relu);
    JVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, partial, sums, dgamma, dbeta,
                       accumulate, N, C, P, ns);
    JVAE_LAUNCH_CHECK();
    const long total = (long)N * C * P;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, st, dy, x, gamma, beta, save_mean,
                       save_invstd, sums, dx, total, C, P, relu);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
