// Per-sample loss reductions (HBM-bound, wave-shuffle reductions) and small elementwise kernels.
//
//   recon  : wmse[l][n] = mean_D((x_reco[l+1][n] - x[n])^2) / sigma^2       module/losses.py:8-27, cvae.py:649-652
//   xent   : ce[r] = -log softmax(logits[r])[y[r % N]]                      module/losses.py:52-86 (F.cross_entropy)
//   act    : ReLU / sigmoid forward and backward                             module/vae_layers/misc.py:25-28
#include "common.h"
#include "jvae_internal.h"

namespace {

// one block per (l, n); x_reco row l+1 is compared with x[n]
__global__ __launch_bounds__(256) void recon_fwd_kernel(const float* __restrict__ xr, const float* __restrict__ x,
                                                        const float* __restrict__ sigma, int sigma_is_log,
                                                        float* __restrict__ wmse, int L, int N, int D) {
    __shared__ float red[17];
    const int n = blockIdx.x, l = blockIdx.y;
    const float* a = xr + ((long)(l + 1) * N + n) * D;
    const float* b = x + (long)n * D;
    float s = 0.f;
    if ((D & 3) == 0) {
        for (int i = threadIdx.x; i < (D >> 2); i += blockDim.x) {
            const f32x4 u = reinterpret_cast<const f32x4*>(a)[i], v = reinterpret_cast<const f32x4*>(b)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = u[j] - v[j]; s += d * d; }
        }
    } else {
        for (int i = threadIdx.x; i < D; i += blockDim.x) { const float d = a[i] - b[i]; s += d * d; }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float sg = sigma[0];
        const float inv2 = sigma_is_log ? __expf(-2.f * sg) : 1.f / (sg * sg);
        wmse[(long)l * N + n] = s / D * inv2;
    }
}

// gxr[l+1][n][:] = g[l][n] * 2 (xr - x) / (sigma^2 D);   gxr[0] = 0;   gsigma_partial[l*N+n] = g * dwmse/dsigma
__global__ __launch_bounds__(256) void recon_bwd_kernel(const float* __restrict__ xr, const float* __restrict__ x,
                                                        const float* __restrict__ sigma, int sigma_is_log,
                                                        const float* __restrict__ g, const float* __restrict__ wmse,
                                                        float* __restrict__ gxr, float* __restrict__ gsig_part,
                                                        int L, int N, int D) {
    const int n = blockIdx.x, l = blockIdx.y;       // l in [0, L]: row of x_reco
    float* o = gxr + ((long)l * N + n) * D;
    if (l == 0) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) o[i] = 0.f;
        return;
    }
    const float* a = xr + ((long)l * N + n) * D;
    const float* b = x + (long)n * D;
    const float sg = sigma[0];
    const float inv2 = sigma_is_log ? __expf(-2.f * sg) : 1.f / (sg * sg);
    const float gv = g[(long)(l - 1) * N + n];
    const float c = gv * 2.f * inv2 / D;
    if ((D & 3) == 0) {
        for (int i = threadIdx.x; i < (D >> 2); i += blockDim.x) {
            const f32x4 u = reinterpret_cast<const f32x4*>(a)[i], v = reinterpret_cast<const f32x4*>(b)[i];
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = c * (u[j] - v[j]);
            reinterpret_cast<f32x4*>(o)[i] = r;
        }
    } else {
        for (int i = threadIdx.x; i < D; i += blockDim.x) o[i] = c * (a[i] - b[i]);
    }
    if (threadIdx.x == 0 && gsig_part) {
        const float wv = wmse[(long)(l - 1) * N + n];
        gsig_part[(long)(l - 1) * N + n] = sigma_is_log ? gv * (-2.f * wv) : gv * (-2.f * wv / sg);
    }
}

// out[0] (+)= sum(v[0..n))
__global__ __launch_bounds__(256) void vec_sum_kernel(const float* __restrict__ v, float* out, int n, int accumulate) {
    __shared__ float red[17];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

// one wave per row r of logits (R, C); target y[r % N]
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ y,
                                                       float* __restrict__ ce, int R, int N, int C) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = logits + (long)r * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(row[c] - mx);
    s = wave_sum(s);
    if (lane == 0) ce[r] = __logf(s) + mx - row[y[r % N]];
}

__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ y,
                                                       const float* __restrict__ g, float* __restrict__ glogits,
                                                       int R, int N, int C) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = logits + (long)r * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(row[c] - mx);
    s = wave_sum(s);
    const float gv = g[r], inv = 1.f / s;
    const int t = (int)y[r % N];
    for (int c = lane; c < C; c += 64) glogits[(long)r * C + c] = gv * (__expf(row[c] - mx) * inv - (c == t ? 1.f : 0.f));
}

// kind: 0 identity, 1 relu, 2 sigmoid
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = kind == 1 ? fmaxf(v, 0.f) : (kind == 2 ? 1.f / (1.f + __expf(-v)) : v);
    }
}

// uses the OUTPUT y: relu -> [y > 0], sigmoid -> y (1 - y)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long n, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float g = dy[i], v = y[i];
        dx[i] = kind == 1 ? (v > 0.f ? g : 0.f) : (kind == 2 ? g * v * (1.f - v) : g);
    }
}

inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int jvae_recon_fwd_f32(const float* x_reco, const float* x, const float* sigma, int sigma_is_log,
                       float* wmse, int L, int N, int D, void* stream) {
    if (!x_reco || !x || !sigma || !wmse || L < 0 || N < 0 || D <= 0) return JVAE_EINVAL;
    if (N == 0 || L == 0) return 0;
    hipLaunchKernelGGL(recon_fwd_kernel, dim3(N, L), dim3(256), 0, (hipStream_t)stream, x_reco, x, sigma, sigma_is_log,
                       wmse, L, N, D);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// gsigma (1 float, may be null) receives sum over (l,n) of g * dwmse/dsigma; ws: L*N floats when gsigma != null
int jvae_recon_bwd_f32(const float* x_reco, const float* x, const float* sigma, int sigma_is_log,
                       const float* g_wmse, const float* wmse, float* g_x_reco, float* gsigma, int accumulate_sigma,
                       int L, int N, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!x_reco || !x || !sigma || !g_wmse || !wmse || !g_x_reco || L < 0 || N < 0 || D <= 0) return JVAE_EINVAL;
    if (gsigma && (!ws || ws_bytes < sizeof(float) * (size_t)L * N)) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(recon_bwd_kernel, dim3(N, L + 1), dim3(256), 0, st, x_reco, x, sigma, sigma_is_log, g_wmse, wmse,
                       g_x_reco, gsigma ? (float*)ws : nullptr, L, N, D);
    JVAE_LAUNCH_CHECK();
    if (gsigma) {
        hipLaunchKernelGGL(vec_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, gsigma, L * N, accumulate_sigma);
        JVAE_LAUNCH_CHECK();
    }
    return 0;
}

int jvae_xent_fwd_f32(const float* logits, const long long* y, float* ce, int R, int N, int C, void* stream) {
    if (!logits || !y || !ce || R < 0 || N <= 0 || C <= 0) return JVAE_EINVAL;
    if (R == 0) return 0;
    hipLaunchKernelGGL(xent_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, logits, y, ce, R, N, C);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_xent_bwd_f32(const float* logits, const long long* y, const float* g_ce, float* g_logits, int R, int N, int C,
                      void* stream) {
    if (!logits || !y || !g_ce || !g_logits || R < 0 || N <= 0 || C <= 0) return JVAE_EINVAL;
    if (R == 0) return 0;
    hipLaunchKernelGGL(xent_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, logits, y, g_ce, g_logits, R, N, C);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_act_fwd_f32(const float* x, float* y, long n, int kind, void* stream) {
    if (!x || !y || n < 0 || kind < 0 || kind > 2) return JVAE_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, kind);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_act_bwd_f32(const float* dy, const float* y, float* dx, long n, int kind, void* stream) {
    if (!dy || !y || !dx || n < 0 || kind < 0 || kind > 2) return JVAE_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, kind);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
