// Global-norm gradient clipping + Adam over flat fp32 buffers (HBM-bound: 4 streams in, 3 out).
//
// Reference: Optimizer.clip -> nn.utils.clip_grad_norm_(params, max_norm) and Optimizer.step ->
// torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay = L2 added to the gradient)
// (module/optimizers.py:39-47,79-81,120-121, called at cvae.py:2460-2461), plus the per-parameter
// NaN/Inf scan of cvae.py:2454-2457 folded into the update as one flag word.
#include "common.h"
#include "jvae_internal.h"

namespace {

// partial[block] = sum of squares of this block's grid-stride share; sqnorm_fold_kernel adds the partials in block order
// (no float atomics: the clip coefficient, hence every parameter update, is reproducible run to run)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[17];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sqnorm_fold_kernel(const float* __restrict__ partial, int nblocks, float* acc, int reset) {
    __shared__ float red[17];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *acc = (reset ? 0.f : *acc) + s;
}

struct AdamP {
    float* p; const float* g; float* m; float* v; long n;
    float lr_over_bc1, b1, b2, eps, wd, inv_sqrt_bc2, max_norm;
    const float* sqnorm; int* flag;
    const float* hyper;  // optional device block (adam_hyper_kernel): [lr, beta1, beta2, step, lr/bc1, 1/sqrt(bc2)] - the step
                         // count and learning rate then live on the device, so a captured HIP graph of the training step
                         // replays with the right bias correction
};

// hyper[3] += 1 (the step count), then the two bias-correction factors of that step
__global__ void adam_hyper_kernel(float* hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double step = (double)hyper[3] + 1.0;
    hyper[3] = (float)step;
    const double bc1 = 1.0 - pow((double)hyper[1], step), bc2 = 1.0 - pow((double)hyper[2], step);
    hyper[4] = (float)((double)hyper[0] / bc1);
    hyper[5] = (float)(1.0 / sqrt(bc2));
}

__device__ __forceinline__ float adam1(float p, float g, float& m, float& v, const AdamP& a, float coef) {
    g = fmaf(g, coef, a.wd * p);
    m = a.b1 * m + (1.f - a.b1) * g;
    v = a.b2 * v + (1.f - a.b2) * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    return p - a.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamP a) {
    if (a.hyper) { a.lr_over_bc1 = a.hyper[4]; a.inv_sqrt_bc2 = a.hyper[5]; a.b1 = a.hyper[1]; a.b2 = a.hyper[2]; }
    float coef = 1.f;
    if (a.max_norm > 0.f && a.sqnorm) coef = fminf(1.f, a.max_norm / (sqrtf(a.sqnorm[0]) + 1e-6f));
    bool bad = false;
    const long n4 = a.n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
        const f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i];
        f32x4 m = reinterpret_cast<f32x4*>(a.m)[i], v = reinterpret_cast<f32x4*>(a.v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mj = m[j], vj = v[j];
            p[j] = adam1(p[j], g[j], mj, vj, a, coef);
            m[j] = mj; v[j] = vj;
            bad |= !isfinite(p[j]);
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < a.n; i += blockDim.x) {
            float mj = a.m[i], vj = a.v[i];
            const float pn = adam1(a.p[i], a.g[i], mj, vj, a, coef);
            a.p[i] = pn; a.m[i] = mj; a.v[i] = vj;
            bad |= !isfinite(pn);
        }
    if (a.flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(a.flag, 1);
}

// torch.optim.SGD (module/optimizers.py:39-40): g' = coef*g + wd*p; buf = first ? g' : mu*buf + g' (dampening 0);
// d = nesterov ? g' + mu*buf : buf (d = g' when mu == 0); p -= lr*d
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long n, float lr, float mu, int nesterov, float wd, int first,
                                                  float max_norm, const float* __restrict__ sqnorm, int* __restrict__ flag) {
    float coef = 1.f;
    if (max_norm > 0.f && sqnorm) coef = fminf(1.f, max_norm / (sqrtf(sqnorm[0]) + 1e-6f));
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        float d = fmaf(wd, pi, coef * g[i]);
        if (mu != 0.f) {
            const float b = first ? d : fmaf(mu, buf[i], d);
            buf[i] = b;
            d = nesterov ? fmaf(mu, b, d) : b;
        }
        const float pn = pi - lr * d;
        p[i] = pn;
        bad |= !isfinite(pn);
    }
    if (flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// g *= min(1, max_norm / (sqrt(sqnorm) + 1e-6))   (what clip_grad_norm_ leaves in .grad)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* g, long n, const float* sqnorm, float max_norm) {
    const float coef = fminf(1.f, max_norm / (sqrtf(sqnorm[0]) + 1e-6f));
    if (coef >= 1.f) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] *= coef;
}

inline int grid_for(long n4) {
    long b = (n4 + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

// *acc (device float) (+)= sum(g^2) (reset != 0: = ); deterministic.  ws: jvae_sqnorm_workspace_bytes() bytes.
size_t jvae_sqnorm_workspace_bytes(void) { return sizeof(float) * 8192; }

int jvae_sqnorm_accum_f32(const float* g, long n, float* acc, int reset, void* ws, size_t ws_bytes, void* stream) {
    if (!acc || n < 0 || (n > 0 && !g)) return JVAE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (reset) {
            hipError_t e = hipMemsetAsync(acc, 0, sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    }
    if (!al16(g)) return JVAE_EINVAL;
    if (!ws || ws_bytes < jvae_sqnorm_workspace_bytes()) return JVAE_EWORKSPACE;
    const int blocks = grid_for(n / 4);                      // <= 8192
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, st, g, n, (float*)ws);
    JVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(sqnorm_fold_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, blocks, acc, reset);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_clip_scale_f32(float* g, long n, const float* sqnorm, float max_norm, void* stream) {
    if (!g || !sqnorm || n < 0 || !(max_norm > 0.f)) return JVAE_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, sqnorm, max_norm);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// step >= 1.  sqnorm / flag may be null (no clipping / no non-finite report).
int jvae_adam_step_f32(float* p, const float* g, float* m, float* v, long n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                       float max_norm, const float* sqnorm, int* nonfinite_flag, void* stream) {
    if (n < 0 || step < 1) return JVAE_EINVAL;
    if (n == 0) return 0;
    if (!p || !g || !m || !v) return JVAE_EINVAL;
    if (!al16(p) || !al16(g) || !al16(m) || !al16(v)) return JVAE_EINVAL;
    AdamP a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.lr_over_bc1 = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.max_norm = max_norm;
    a.sqnorm = sqnorm; a.flag = nonfinite_flag; a.hyper = nullptr;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// Same update with the step count, learning rate and betas in DEVICE memory: hyper = 6 floats [lr, beta1, beta2, step
// (completed steps), scratch, scratch].  advance != 0 first increments hyper[3] and refreshes the bias corrections (once per
// optimiser step; further parameter groups of the same step pass advance = 0 with their own block, or share one).  Nothing
// here depends on host state, so the call can be captured into a HIP graph and replayed.
int jvae_adam_step_dev_f32(float* p, const float* g, float* m, float* v, long n, float* hyper, int advance,
                           float eps, float weight_decay, float max_norm, const float* sqnorm, int* nonfinite_flag,
                           void* stream) {
    if (n < 0 || !hyper) return JVAE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (advance) {
        hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, st, hyper);
        JVAE_LAUNCH_CHECK();
    }
    if (n == 0) return 0;
    if (!p || !g || !m || !v) return JVAE_EINVAL;
    if (!al16(p) || !al16(g) || !al16(m) || !al16(v)) return JVAE_EINVAL;
    AdamP a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
    a.lr_over_bc1 = 0.f; a.inv_sqrt_bc2 = 0.f; a.b1 = 0.f; a.b2 = 0.f;
    a.eps = eps; a.wd = weight_decay; a.max_norm = max_norm;
    a.sqnorm = sqnorm; a.flag = nonfinite_flag; a.hyper = hyper;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, a);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_sgd_step_f32(float* p, const float* g, float* buf, long n, float lr, float momentum, int nesterov,
                      float weight_decay, int first_step, float max_norm, const float* sqnorm, int* nonfinite_flag,
                      void* stream) {
    if (n < 0 || momentum < 0.f || (nesterov && momentum <= 0.f)) return JVAE_EINVAL;
    if (n == 0) return 0;
    if (!p || !g || (momentum != 0.f && !buf)) return JVAE_EINVAL;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum, nesterov,
                       weight_decay, first_step, max_norm, sqnorm, nonfinite_flag);
    JVAE_LAUNCH_CHECK();
    return 0;
}

const char* jvae_version(void) { return "jvae_hip 0.1 (gfx950)"; }

}  // extern "C"
