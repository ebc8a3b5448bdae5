// Fused latent kernel: clip(log sigma^2) -> reparameterised samples z -> per-sample KL terms to the
// class-conditional prior, forward and backward.  One wavefront per sample; lanes stride the latent
// dimension K; per-sample reductions are wave shuffles (no LDS, no atomics in forward).
//
// Reference path replaced:
//   Encoder.forward clip            module/vae_layers/layers.py:388-394
//   Sampling.forward                module/vae_layers/layers.py:230-244   (eps is an INPUT here: the caller draws it)
//   GaussianPrior.kl / mahala / whiten / trace_prod_by_var / log_det_per_class   module/priors.py:173-326
//   TiltedGaussianPrior.kl          module/priors.py:389-408
//   UniformWithGaussianTailPrior.kl module/priors.py:429-476
//   dzdist                          cvae.py:747-753
//
// Naming trap kept from the reference: the prior's `_var_parameter` T is the WHITENING factor
// (whiten = T.(mu - m), precision diag = T^2, log|Sigma| = -2 sum log|T|), not a variance.
#include "common.h"
#include "jvae_internal.h"

namespace {

enum { PRIOR_GAUSS = 0, PRIOR_TILTED = 1, PRIOR_UNIFORM = 2 };
enum { VAR_SCALAR = 0, VAR_DIAG = 1, VAR_FULL = 2 };

struct LatentP {
    const float* mu;        // (N,K)
    const float* lv_raw;    // (N,K) before the +-20 clip
    const float* eps;       // (L+1,N,K), row 0 is zero
    const long long* y;     // (N,) class of each sample
    const float* means;     // (C,K)
    const float* T;         // (C,) | (C,K) | (C,K,K) whitening factor (full: lower triangle is used)
    const float* dict;      // [K] mean of the dictionary rows, then [1] dict_norm_var (may be null)
    int N, K, L, C;
    int prior, var_dim;
    float tau, alpha;       // tilted: tau; uniform: tau and alpha = log(2 tau) - log(2 Phi(tau) - 1)
    float w;                // warm-up weight on var_kl
    float sampled;          // 0/1: is_sampled
    float forced_lv;        // used when has_forced
    int has_forced;
};

__device__ __forceinline__ float hardtanh1(float v) { return fminf(1.f, fmaxf(-1.f, v)); }

__global__ __launch_bounds__(256) void latent_fwd_kernel(LatentP p, float* __restrict__ lv_out, float* __restrict__ z,
                                                         float* __restrict__ kl, float* __restrict__ zdist,
                                                         float* __restrict__ var_kl, float* __restrict__ dzdist) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= p.N) return;
    const int K = p.K;
    const long long cls = p.y[n];
    const float* m = p.means + (long)cls * K;
    const long row = (long)n * K;
    float dist = 0.f, trace = 0.f, logdet = 0.f, logdet_p = 0.f, dz = 0.f;
    float u_elogq = 0.f, u_nel = 0.f;    // uniform prior partial sums
    for (int k = lane; k < K; k += 64) {
        const float mu = p.mu[row + k];
        float lv = p.has_forced ? p.forced_lv : fminf(20.f, fmaxf(-20.f, p.lv_raw[row + k]));
        lv_out[row + k] = lv;
        const float sd = __expf(0.5f * lv);
        for (int l = 0; l <= p.L; ++l) {
            const long o = ((long)l * p.N + n) * K + k;
            z[o] = mu + sd * p.eps[o] * p.sampled;
        }
        const float d = mu - m[k];
        if (p.dict) { const float t = mu - p.dict[k]; dz += t * t; }
        if (p.prior == PRIOR_UNIFORM) {
            const float c = 1.8378770664093453f;            // log(2 pi)
            const float span = 3.4641016151377544f * sd;    // 2 sqrt(3) sd
            const float a_ = p.tau * hardtanh1((d - 0.5f * span) / p.tau);
            const float b_ = p.tau * hardtanh1((d + 0.5f * span) / p.tau);
            const float elogq = -0.5f * lv - 1.2424533248940002f;   // -0.5 log 12
            float nel = (c + d * d + span * span / 12.f) * 0.5f;
            nel += (p.alpha - 0.5f * c) * (b_ - a_) / span;
            nel -= (b_ * b_ * b_ - a_ * a_ * a_) / span / 6.f;
            dist += d * d;
            u_elogq += elogq;
            u_nel += nel;
            continue;
        }
        float wd, pd;
        if (p.var_dim == VAR_SCALAR)      { const float t = p.T[cls]; wd = d * t; pd = t * t; }
        else if (p.var_dim == VAR_DIAG)   { const float t = p.T[(long)cls * K + k]; wd = d * t; pd = t * t;
                                            logdet_p += -2.f * __logf(fabsf(t)); }
        else {
            const float* Tc = p.T + (long)cls * K * K;
            wd = 0.f;                                        // row k of tril(T) . d
            for (int j = 0; j <= k; ++j) wd += Tc[(long)k * K + j] * (p.mu[row + j] - m[j]);
            pd = 0.f;                                        // column k of tril(T)^2 summed over rows
            for (int i = k; i < K; ++i) { const float t = Tc[(long)i * K + k]; pd += t * t; }
            logdet_p += -2.f * __logf(fabsf(Tc[(long)k * K + k]));
        }
        dist += wd * wd;
        trace += __expf(lv) * pd;
        logdet += lv;
    }
    dist = wave_sum(dist);
    if (p.dict) dz = wave_sum(dz);
    float kl_v, vkl_v;
    if (p.prior == PRIOR_UNIFORM) {
        u_elogq = wave_sum(u_elogq);
        u_nel = wave_sum(u_nel);
        const float vk = u_elogq + K * p.alpha;
        kl_v = fmaxf(u_elogq + u_nel, vk);
        if (p.w != 1.f) kl_v += (p.w - 1.f) * vk;
        vkl_v = 2.f * vk;
    } else if (p.prior == PRIOR_TILTED) {
        const float r = sqrtf(dist) - p.tau;
        kl_v = 0.5f * r * r;
        vkl_v = 0.f;
    } else {
        trace = wave_sum(trace);
        logdet = wave_sum(logdet);
        if (p.var_dim == VAR_SCALAR) logdet_p = -2.f * K * __logf(p.T[cls]);
        else logdet_p = wave_sum(logdet_p);
        vkl_v = trace - logdet + logdet_p - (float)K;
        kl_v = 0.5f * (dist + p.w * vkl_v);
    }
    if (lane == 0) {
        kl[n] = kl_v;
        zdist[n] = dist;
        var_kl[n] = vkl_v;
        if (dzdist) dzdist[n] = p.dict ? dz + p.dict[K] : 0.f;
    }
}

// Backward.  Upstream: gz (L+1,N,K) [may be null], g_kl, g_zdist, g_vkl (N,) [each may be null],
// gmu_direct / glv_direct (N,K) [may be null: gradients reaching mu / clipped log_var from other consumers].
// Outputs: gmu, glv_raw (N,K); gd_scratch (N,K) = per-sample contribution to the gradient of its class mean (reduced in a
// fixed order by means_grad_kernel: deterministic).  The gradient of the whitening factor T (diag / full variance) is
// deterministic as well - no float atomics: diag writes the per-sample contribution to gt_scratch (N,K), folded per class
// in sample order by means_grad_kernel; full writes wd = tril(T).d to gt_scratch and (g_dist, g_var) to gpair (N,2), from
// which gT_full_kernel rebuilds the rank-one terms per class in sample order.
// Full variance: d and wd of the wave's sample are staged in LDS (2 K floats per wave) so that the per-sample cost is
// O(K^2) (it was O(K^3): every lane recomputed every wd_i).
__global__ __launch_bounds__(256) void latent_bwd_kernel(LatentP p, const float* __restrict__ lv,   // clipped
                                                         const float* __restrict__ gz, const float* __restrict__ g_kl,
                                                         const float* __restrict__ g_zdist, const float* __restrict__ g_vkl,
                                                         const float* __restrict__ gmu_direct,
                                                         const float* __restrict__ glv_direct,
                                                         const float* __restrict__ kl_fwd_terms,   // uniform: (N,2) = (elogq+nel, vk)
                                                         float* __restrict__ gmu, float* __restrict__ glv_raw,
                                                         float* gd_scratch, float* gt_scratch, float* gpair) {
    extern __shared__ float lds_dw[];          // full variance only: per wave d[K] then wd[K]
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int n_raw = blockIdx.x * (blockDim.x >> 6) + wv;
    const bool valid = n_raw < p.N;
    const int n = valid ? n_raw : p.N - 1;     // out-of-range waves shadow the last sample and store nothing
    const int K = p.K;
    const long long cls = p.y[n];
    const float* m = p.means + (long)cls * K;
    const long row = (long)n * K;
    const float gk = g_kl ? g_kl[n] : 0.f;
    const float gd_in = g_zdist ? g_zdist[n] : 0.f;
    const float gv_in = g_vkl ? g_vkl[n] : 0.f;

    // gradient reaching `distance` and `var_kl` (gaussian) ...
    float g_dist = gd_in, g_var = gv_in;
    bool uni_first = true;          // uniform: which branch of the max() was taken
    if (p.prior == PRIOR_GAUSS) { g_dist += 0.5f * gk; g_var += 0.5f * p.w * gk; }
    else if (p.prior == PRIOR_TILTED) {
        float dist = 0.f;
        for (int k = lane; k < K; k += 64) { const float d = (p.mu[row + k] - m[k]) * p.T[cls]; dist += d * d; }
        dist = wave_sum(dist);
        const float r = sqrtf(dist);
        g_dist += gk * 0.5f * (1.f - p.tau / r);     // d/dD 0.5 (sqrt D - tau)^2
        g_var = 0.f;
    } else {
        uni_first = kl_fwd_terms[2 * n] >= kl_fwd_terms[2 * n + 1];
    }

    const bool full = p.prior != PRIOR_UNIFORM && p.var_dim == VAR_FULL;
    float* d_s = lds_dw + (size_t)wv * 2 * K;
    float* wd_s = d_s + K;
    if (full) {                                // uniform over the block: every wave takes the barriers
        const float* Tc = p.T + (long)cls * K * K;
        for (int k = lane; k < K; k += 64) d_s[k] = p.mu[row + k] - m[k];
        __syncthreads();
        for (int k = lane; k < K; k += 64) {   // wd_k = sum_{j<=k} T_kj d_j  (the forward's summation order)
            float wd = 0.f;
            for (int j = 0; j <= k; ++j) wd += Tc[(long)k * K + j] * d_s[j];
            wd_s[k] = wd;
            if (valid && gt_scratch) gt_scratch[row + k] = wd;
        }
        __syncthreads();
        if (valid && gpair && lane == 0) { gpair[2 * n] = g_dist; gpair[2 * n + 1] = g_var; }
    }
    if (!valid) return;

    for (int k = lane; k < K; k += 64) {
        const float mu = p.mu[row + k];
        const float lvk = lv[row + k];
        const float sd = __expf(0.5f * lvk);
        float g_mu = gmu_direct ? gmu_direct[row + k] : 0.f;
        float g_lv = glv_direct ? glv_direct[row + k] : 0.f;
        if (gz) {
            for (int l = 0; l <= p.L; ++l) {
                const long o = ((long)l * p.N + n) * K + k;
                const float g = gz[o];
                g_mu += g;
                g_lv += g * 0.5f * sd * p.eps[o] * p.sampled;
            }
        }
        const float d = mu - m[k];
        float g_d = 0.f;          // gradient wrt d = mu - m_y (flows to mu and, negated, to the mean)
        if (p.prior == PRIOR_UNIFORM) {
            // kl = max(sum(elogq) + sum(nel), vk) + (w-1) vk ; var_kl = 2 vk ; vk = sum(elogq) + K alpha
            const float c = 1.8378770664093453f;
            const float span = 3.4641016151377544f * sd;
            const float lo = (d - 0.5f * span) / p.tau, hi = (d + 0.5f * span) / p.tau;
            const float a_ = p.tau * hardtanh1(lo), b_ = p.tau * hardtanh1(hi);
            const float da = (lo > -1.f && lo < 1.f) ? 1.f : 0.f;   // d a_/d (d - span/2)
            const float db = (hi > -1.f && hi < 1.f) ? 1.f : 0.f;
            const float coef = p.alpha - 0.5f * c;
            const float g_sum_elogq = (uni_first ? gk : 0.f) + (uni_first ? 0.f : gk) + (p.w - 1.f) * gk + 2.f * gv_in;
            const float g_nel = uni_first ? gk : 0.f;
            // nel = (c + d^2 + span^2/12)/2 + coef (b_-a_)/span - (b_^3 - a_^3)/(6 span)
            const float dnel_da = -coef / span + a_ * a_ / (2.f * span);
            const float dnel_db = coef / span - b_ * b_ / (2.f * span);
            const float dnel_dspan_direct = span / 12.f - coef * (b_ - a_) / (span * span)
                                            + (b_ * b_ * b_ - a_ * a_ * a_) / (6.f * span * span);
            const float dnel_dd = d + dnel_da * da + dnel_db * db;
            const float dnel_dspan = dnel_dspan_direct + dnel_da * da * (-0.5f) + dnel_db * db * 0.5f;
            g_d += g_nel * dnel_dd + gd_in * 2.f * d;
            g_lv += g_nel * dnel_dspan * 0.5f * span + g_sum_elogq * (-0.5f);
        } else {
            float Td;         // (T^T T d)_k : gradient of distance/2 wrt d_k
            if (p.var_dim == VAR_SCALAR) { const float t = p.T[cls]; Td = t * t * d; if (p.prior == PRIOR_GAUSS) g_lv += g_var * (__expf(lvk) * t * t - 1.f); }
            else if (p.var_dim == VAR_DIAG) {
                const float t = p.T[(long)cls * K + k];
                Td = t * t * d;
                g_lv += g_var * (__expf(lvk) * t * t - 1.f);
                if (gt_scratch) gt_scratch[row + k] = g_dist * 2.f * t * d * d + g_var * (2.f * __expf(lvk) * t - 2.f / t);
            } else {
                const float* Tc = p.T + (long)cls * K * K;
                // (T^T wd)_k = sum_{i>=k} T_ik wd_i ;  pd_k = sum_{i>=k} T_ik^2
                Td = 0.f;
                float pd = 0.f;
                for (int i = k; i < K; ++i) {
                    const float t = Tc[(long)i * K + k];
                    Td += t * wd_s[i];
                    pd += t * t;
                }
                g_lv += g_var * (__expf(lvk) * pd - 1.f);
            }
            g_d += g_dist * 2.f * Td;
        }
        g_mu += g_d;
        if (gd_scratch) gd_scratch[row + k] = -g_d;          // summed per class in sample order by means_grad_kernel
        gmu[row + k] = g_mu;
        const float raw = p.lv_raw ? p.lv_raw[row + k] : 0.f;
        const bool pass = p.has_forced ? false : (raw >= -20.f && raw <= 20.f);
        glv_raw[row + k] = pass ? g_lv : 0.f;
    }
}

// Full variance: gT[c][i][k] += sum over the samples n of class c, IN SAMPLE ORDER, of
//   2 g_dist_n wd_ni d_nk + 2 g_var_n exp(lv_nk) T_cik - [i == k] 2 g_var_n / T_ckk          (i >= k: lower triangle)
// One block = 256 (i, k) cells of one class; every thread walks the N samples (the class test is block-uniform).
__global__ __launch_bounds__(256) void gT_full_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                      const long long* __restrict__ y, const float* __restrict__ means,
                                                      const float* __restrict__ T, const float* __restrict__ wd,
                                                      const float* __restrict__ gpair, float* __restrict__ gT,
                                                      int N, int K) {
    const int cls = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= K * K) return;
    const int i = cell / K, k = cell % K;
    if (i < k) return;
    const float t = T[((long)cls * K + i) * K + k];
    const float mk = means[(long)cls * K + k];
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        if ((int)y[n] != cls) continue;
        const float g_dist = gpair[2 * n], g_var = gpair[2 * n + 1];
        const float d = mu[(long)n * K + k] - mk;
        float g = g_dist * 2.f * wd[(long)n * K + i] * d + g_var * 2.f * __expf(lv[(long)n * K + k]) * t;
        if (i == k) g += g_var * (-2.f / t);
        acc += g;
    }
    gT[((long)cls * K + i) * K + k] += acc;
}

// gmeans[cls][k] += sum over the samples n of class cls of gd[n][k]: 16 contiguous sample chunks per output are summed
// in sample order by 16 threads, the 16 partial sums are folded in chunk order (deterministic, no atomics; a single
// thread per output walking all N samples took 58 us at N = 512 on the critical path of backward)
__global__ __launch_bounds__(1024) void means_grad_kernel(const float* __restrict__ gd, const long long* __restrict__ y,
                                                          float* __restrict__ gmeans, int N, int C, int K) {
    __shared__ float part[16][64];
    const int ix = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + ix;
    const int per = (N + 15) / 16, n0 = chunk * per, n1 = min(N, n0 + per);
    float s = 0.f;
    if (i < C * K) {
        const int cls = i / K, k = i % K;
        for (int n = n0; n < n1; ++n)
            if ((int)y[n] == cls) s += gd[(long)n * K + k];
    }
    part[chunk][ix] = s;
    __syncthreads();
    if (chunk == 0 && i < C * K) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) t += part[c][ix];
        gmeans[i] += t;
    }
}

// dict[0..K) = mean over classes of the dictionary rows; dict[K] = mean_c |m_c|^2 - |mean|^2   (cvae.py:747-752)
// One block; the C rows are split over blockDim/K (at least 1) thread groups per latent dimension.
__global__ __launch_bounds__(1024) void dict_stats_kernel(const float* __restrict__ means, float* __restrict__ dict, int C, int K) {
    __shared__ float red[17];
    __shared__ float part[1024];
    const int tid = threadIdx.x;
    const int groups = max(1, (int)blockDim.x / K);          // row groups
    float sq = 0.f, msq = 0.f;
    for (int k0 = 0; k0 < K; k0 += blockDim.x / groups) {
        const int k = k0 + tid % (blockDim.x / groups), grp = tid / (blockDim.x / groups);
        float s = 0.f;
        if (k < K && grp < groups)
            for (int c = grp; c < C; c += groups) { const float v = means[(long)c * K + k]; s += v; sq += v * v; }
        part[tid] = s;
        __syncthreads();
        if (grp == 0 && k < K) {
            float t = 0.f;
            for (int g2 = 0; g2 < groups; ++g2) t += part[g2 * (blockDim.x / groups) + tid % (blockDim.x / groups)];
            t /= C;
            dict[k] = t;
            msq += t * t;
        }
        __syncthreads();
    }
    sq = block_sum(sq, red);
    msq = block_sum(msq, red);
    if (tid == 0) dict[K] = sq / C - msq;
}

// forward-only helper for the uniform prior's backward: terms[n] = (sum elogq + sum nel, vk)
__global__ __launch_bounds__(256) void uniform_terms_kernel(LatentP p, const float* __restrict__ lv, float* __restrict__ terms) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= p.N) return;
    const int K = p.K;
    const float* m = p.means + (long)p.y[n] * K;
    float se = 0.f, sn = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float c = 1.8378770664093453f;
        const float d = p.mu[(long)n * K + k] - m[k];
        const float lvk = lv[(long)n * K + k];
        const float span = 3.4641016151377544f * __expf(0.5f * lvk);
        const float a_ = p.tau * hardtanh1((d - 0.5f * span) / p.tau);
        const float b_ = p.tau * hardtanh1((d + 0.5f * span) / p.tau);
        se += -0.5f * lvk - 1.2424533248940002f;
        float nel = (c + d * d + span * span / 12.f) * 0.5f;
        nel += (p.alpha - 0.5f * c) * (b_ - a_) / span;
        nel -= (b_ * b_ * b_ - a_ * a_ * a_) / span / 6.f;
        sn += nel;
    }
    se = wave_sum(se);
    sn = wave_sum(sn);
    if (lane == 0) { terms[2 * n] = se + sn; terms[2 * n + 1] = se + K * p.alpha; }
}

bool fill(LatentP* p, const float* mu, const float* lv_raw, const float* eps, const long long* y, const float* means,
          const float* T, const float* dict, int N, int K, int L, int C, int prior, int var_dim, float tau, float alpha,
          float w, int sampled, int has_forced, float forced_lv) {
    if (!mu || !y || !means || N < 0 || K <= 0 || L < 0 || C <= 0) return false;
    if (prior < 0 || prior > 2 || var_dim < 0 || var_dim > 2) return false;
    if (prior != PRIOR_UNIFORM && !T) return false;
    if (prior != PRIOR_GAUSS && var_dim != VAR_SCALAR) return false;
    if (!has_forced && !lv_raw) return false;
    p->mu = mu; p->lv_raw = lv_raw; p->eps = eps; p->y = y; p->means = means; p->T = T; p->dict = dict;
    p->N = N; p->K = K; p->L = L; p->C = C; p->prior = prior; p->var_dim = var_dim; p->tau = tau; p->alpha = alpha;
    p->w = w; p->sampled = sampled ? 1.f : 0.f; p->has_forced = has_forced; p->forced_lv = forced_lv;
    return true;
}

}  // namespace

extern "C" {

int jvae_dict_stats_f32(const float* means, float* dict, int C, int K, void* stream) {
    if (!means || !dict || C <= 0 || K <= 0) return JVAE_EINVAL;
    hipLaunchKernelGGL(dict_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, means, dict, C, K);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_latent_fwd_f32(const float* mu, const float* lv_raw, const float* eps, const long long* y,
                        const float* means, const float* T, const float* dict,
                        float* lv, float* z, float* kl, float* zdist, float* var_kl, float* dzdist,
                        int N, int K, int L, int C, int prior, int var_dim, float tau, float alpha, float w,
                        int sampled, int has_forced, float forced_lv, void* stream) {
    LatentP p;
    if (!fill(&p, mu, lv_raw, eps, y, means, T, dict, N, K, L, C, prior, var_dim, tau, alpha, w, sampled, has_forced, forced_lv))
        return JVAE_EINVAL;
    if (!eps || !lv || !z || !kl || !zdist || !var_kl) return JVAE_EINVAL;
    if (N == 0) return 0;
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, p, lv, z, kl, zdist, var_kl, dzdist);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// ws: 4*N + 2*N*K floats cover every mode (uniform terms, mean contributions, T contributions)
int jvae_latent_bwd_f32(const float* mu, const float* lv_raw, const float* lv, const float* eps, const long long* y,
                        const float* means, const float* T,
                        const float* gz, const float* g_kl, const float* g_zdist, const float* g_vkl,
                        const float* gmu_direct, const float* glv_direct,
                        float* gmu, float* glv_raw, float* gmeans, float* gT,
                        int N, int K, int L, int C, int prior, int var_dim, float tau, float alpha, float w,
                        int sampled, int has_forced, void* ws, size_t ws_bytes, void* stream) {
    LatentP p;
    if (!fill(&p, mu, lv_raw, eps, y, means, T, nullptr, N, K, L, C, prior, var_dim, tau, alpha, w, sampled, has_forced, 0.f))
        return JVAE_EINVAL;
    if (!lv || !gmu || !glv_raw || (gz && !eps)) return JVAE_EINVAL;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float* terms = nullptr;
    if (prior == PRIOR_UNIFORM) {
        if (!ws || ws_bytes < sizeof(float) * 2 * (size_t)N) return JVAE_EWORKSPACE;
        terms = (float*)ws;
        hipLaunchKernelGGL(uniform_terms_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, p, lv, terms);
        JVAE_LAUNCH_CHECK();
    }
    // workspace layout (floats): [2N uniform terms][N*K mean contributions][N*K T contributions / wd][2N (g_dist, g_var)]
    const size_t NK = (size_t)N * K;
    float* gd = nullptr;
    if (gmeans) {
        if (!ws || ws_bytes < sizeof(float) * ((size_t)2 * N + NK)) return JVAE_EWORKSPACE;
        gd = (float*)ws + (size_t)2 * N;
    }
    float *gts = nullptr, *gpair = nullptr;
    const bool needT = gT && prior == PRIOR_GAUSS && var_dim != VAR_SCALAR;
    if (needT) {
        if (!ws || ws_bytes < sizeof(float) * ((size_t)4 * N + 2 * NK)) return JVAE_EWORKSPACE;
        gts = (float*)ws + (size_t)2 * N + NK;
        gpair = gts + NK;
    }
    const size_t lds = (prior != PRIOR_UNIFORM && var_dim == VAR_FULL) ? sizeof(float) * 4 * 2 * (size_t)K : 0;
    if (lds > 64 * 1024) return JVAE_ENOTSUP;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(cdiv(N, 4)), dim3(256), lds, st, p, lv, gz, g_kl, g_zdist, g_vkl,
                       gmu_direct, glv_direct, terms, gmu, glv_raw, gd, gts, gpair);
    JVAE_LAUNCH_CHECK();
    if (gmeans) {
        hipLaunchKernelGGL(means_grad_kernel, dim3(cdiv((long)C * K, 64)), dim3(1024), 0, st, (const float*)gd, y, gmeans, N, C, K);
        JVAE_LAUNCH_CHECK();
    }
    if (needT && var_dim == VAR_DIAG) {
        hipLaunchKernelGGL(means_grad_kernel, dim3(cdiv((long)C * K, 64)), dim3(1024), 0, st, (const float*)gts, y, gT, N, C, K);
        JVAE_LAUNCH_CHECK();
    } else if (needT) {
        hipLaunchKernelGGL(gT_full_kernel, dim3(cdiv((long)K * K, 256), C), dim3(256), 0, st, mu, lv, y, means, T,
                           (const float*)gts, (const float*)gpair, gT, N, K);
        JVAE_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
