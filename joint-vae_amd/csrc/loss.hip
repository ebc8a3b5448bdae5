// Per-sample loss reductions (HBM-bound, wave-shuffle reductions) and small elementwise kernels.
//
//   recon  : wmse[l][n] = mean_D((x_reco[l+1][n] - x[n])^2) / sigma^2       module/losses.py:8-27, cvae.py:649-652
//   xent   : ce[r] = -log softmax(logits[r])[y[r % N]]                      module/losses.py:52-86 (F.cross_entropy)
//   act    : ReLU / sigmoid forward and backward                             module/vae_layers/misc.py:25-28
#include "common.h"
#include "jvae_internal.h"

namespace {

// sigma_mode (cvae.py:626-670): 0 = one value, 1 = one log value (learned), 2 = log sigma per sample (coded by the
// encoder: sigma[n]), 3 = sigma follows each sample's own rmse (the division happens in the ELBO kernels: 1 here)
enum { SIG_VALUE = 0, SIG_LOG = 1, SIG_CODED = 2, SIG_RMSE = 3 };
__device__ __forceinline__ float sigma_inv2(const float* sigma, int mode, int n) {
    if (mode == SIG_RMSE || !sigma) return 1.f;      // sigma NULL: the plain mean square (jvae_mse_rows_*)
    const float sg = sigma[mode == SIG_CODED ? n : 0];
    return mode == SIG_VALUE ? 1.f / (sg * sg) : __expf(-2.f * sg);
}

// one block per (l, n); x_reco row l+1 is compared with x[n]
// row0 = 1: xr is the (L+1, N, D) reconstruction (row 0 = the mean path, not part of the loss); row0 = 0: xr holds the L rows only
__global__ __launch_bounds__(256) void recon_fwd_kernel(const float* __restrict__ xr, const float* __restrict__ x,
                                                        const float* __restrict__ sigma, int sigma_mode,
                                                        float* __restrict__ wmse, int L, int N, int D, int row0) {
    __shared__ float red[17];
    const int n = blockIdx.x, l = blockIdx.y;
    const float* a = xr + ((long)(l + row0) * N + n) * D;
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
    if (threadIdx.x == 0) wmse[(long)l * N + n] = s / D * sigma_inv2(sigma, sigma_mode, n);
}

// gxr[l+1][n][:] = g[l][n] * 2 (xr - x) / (sigma^2 D);   gxr[0] = 0;   gsigma_partial[l*N+n] = g * dwmse/dsigma
// sigma_fwd (mode 0 only, may be null): the value sigma had in the forward pass.  The reference's decay rule changes
// `sigma.data` IN PLACE between forward and backward (layers.py:168, called at cvae.py:769) and autograd's saved divisor is
// that very tensor, so its backward is 2 (x_reco - x) / (D sigma_fwd sigma_now): reproduced here (golden c2_n8_decay).
__global__ __launch_bounds__(256) void recon_bwd_kernel(const float* __restrict__ xr, const float* __restrict__ x,
                                                        const float* __restrict__ sigma, int sigma_mode,
                                                        const float* __restrict__ sigma_fwd,
                                                        const float* __restrict__ g, const float* __restrict__ wmse,
                                                        float* __restrict__ gxr, float* __restrict__ gsig_part,
                                                        int L, int N, int D, int rows_only) {
    // l in [0, L]: row of x_reco (row 0 receives zeros); rows_only: xr / gxr hold the L sample rows only (grid.y = L)
    const int n = blockIdx.x, l = blockIdx.y + rows_only;
    float* o = gxr + ((long)(l - rows_only) * N + n) * D;
    if (l == 0) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) o[i] = 0.f;
        return;
    }
    const float* a = xr + ((long)(l - rows_only) * N + n) * D;
    const float* b = x + (long)n * D;
    float inv2 = sigma_inv2(sigma, sigma_mode, n);
    if (sigma_mode == SIG_VALUE && sigma_fwd) inv2 = 1.f / (sigma_fwd[0] * sigma[0]);
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
        gsig_part[(long)(l - 1) * N + n] = sigma_mode == SIG_VALUE ? gv * (-2.f * wv / sigma[0]) : gv * (-2.f * wv);
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

// out[n] (+)= sum_l part[l][n]   (coded sigma: per-sample gradient of log sigma_n)
__global__ __launch_bounds__(256) void rows_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int L, int N,
                                                        int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = accumulate ? out[n] : 0.f;
    for (int l = 0; l < L; ++l) s += part[(long)l * N + n];
    out[n] = s;
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

// kind: 0 identity, 1 relu, 2 sigmoid, 3 leaky relu (negative slope 0.01: nn.LeakyReLU(), module/vae_layers/misc.py:24-27)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = kind == 1 ? fmaxf(v, 0.f) : (kind == 2 ? 1.f / (1.f + __expf(-v)) : (kind == 3 ? fmaxf(v, JVAE_LEAKY_SLOPE * v) : v));
    }
}

// uses the OUTPUT y: relu -> [y > 0], sigmoid -> y (1 - y), leaky relu -> [y > 0] + 0.01 [y <= 0] (the sign of y is that of x)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long n, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float g = dy[i], v = y[i];
        dx[i] = kind == 1 ? (v > 0.f ? g : 0.f) : (kind == 2 ? g * v * (1.f - v) : (kind == 3 ? (v > 0.f ? g : JVAE_LEAKY_SLOPE * g) : g));
    }
}

// ---- ELBO assembly (cvae.py:773-791,887-902): per-sample, N threads ------------------------------------------
//   wmse = mean_l wmse_s ; cross_x = D/2 (2 log sigma + wmse + log 2pi) ; total = cross_x + cw*ce + beta*kl
__global__ __launch_bounds__(256) void elbo_fwd_kernel(const float* __restrict__ wmse_s, const float* __restrict__ kl,
                                                       const float* __restrict__ ce, const float* __restrict__ sigma,
                                                       int sigma_mode, float* __restrict__ wmse, float* __restrict__ cross_x,
                                                       float* __restrict__ total, float* __restrict__ mse,
                                                       int L, int N, float D, float beta, float cw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += wmse_s[(long)l * N + n];
    s /= L;
    float ls, sig2;
    if (sigma_mode == SIG_RMSE) {              // sigma_n^2 = the sample's mse over its L draws (cvae.py:662-666)
        const float m = s;
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += wmse_s[(long)l * N + n] / m;
        s = t / L;                             // = 1 up to rounding, as the reference computes it
        ls = 0.5f * __logf(m);
        sig2 = m;
    } else {
        const float sg = sigma[sigma_mode == SIG_CODED ? n : 0];
        ls = sigma_mode == SIG_VALUE ? __logf(sg) : sg;
        sig2 = sigma_mode == SIG_VALUE ? sg * sg : __expf(2.f * sg);
    }
    const float cx = 0.5f * D * (2.f * ls + s + 1.8378770664093453f);
    wmse[n] = s;
    cross_x[n] = cx;
    total[n] = cx + (ce ? cw * ce[n] : 0.f) + beta * kl[n];
    if (mse) mse[n] = s * sig2;
}

// upstream g_wmse / g_cx / g_tot (N,) (any may be null) -> g_wmse_s (L,N), g_kl, g_ce (N,), gsig_part (N,)
// sigma_mode 3: `sigma` points to the forward's wmse_s (L,N) (the per-sample mse the normalisation needs)
__global__ __launch_bounds__(256) void elbo_bwd_kernel(const float* __restrict__ g_wmse, const float* __restrict__ g_cx,
                                                       const float* __restrict__ g_tot, const float* __restrict__ sigma,
                                                       int sigma_mode, float* __restrict__ g_wmse_s, float* __restrict__ g_kl,
                                                       float* __restrict__ g_ce, float* __restrict__ gsig_part,
                                                       int L, int N, float D, float beta, float cw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float gt = g_tot ? g_tot[n] : 0.f;
    const float gc = (g_cx ? g_cx[n] : 0.f) + gt;
    float gw;
    if (sigma_mode == SIG_RMSE) {
        // wmse = mean_l(w_l / m) with m = mean_l w_l is constant (= 1): only log sigma_n = log(m) / 2 carries gradient
        float m = 0.f;
        for (int l = 0; l < L; ++l) m += sigma[(long)l * N + n];
        m /= L;
        gw = gc * 0.5f * D / (m * L);
    } else {
        gw = ((g_wmse ? g_wmse[n] : 0.f) + gc * 0.5f * D) / L;
    }
    for (int l = 0; l < L; ++l) g_wmse_s[(long)l * N + n] = gw;
    if (g_kl) g_kl[n] = beta * gt;
    if (g_ce) g_ce[n] = cw * gt;
    if (gsig_part) gsig_part[n] = sigma_mode == SIG_VALUE ? gc * D / sigma[0] : gc * D;
}

// ---- packed measures (cvae.py:619-624,689-724,747-762 + layers.py:323-348): ONE device buffer, ONE read-back -----
// out[0] sigma rms, [1] mean x^2 (from sumsq), [2] mean mse, [3] rmse = sqrt([2]), [4] mean zdist, [5] mean var_kl,
// [6] ld-norm = mean(m^2), [7] imut-zy (capacity bound), [8] d-mind, [9] non-finite flag of the optimiser
__global__ __launch_bounds__(256) void measures_kernel(const float* __restrict__ sumsq_x, float nx,
                                                       const float* __restrict__ wmse, const float* __restrict__ zdist,
                                                       const float* __restrict__ var_kl, int N, int Nz,
                                                       const float* __restrict__ sigma, int sigma_is_log,
                                                       const float* __restrict__ means, int C, int K,
                                                       const int* __restrict__ flag, const float* __restrict__ prev,
                                                       int batch, float* __restrict__ out) {
    __shared__ float red[17];
    const int tid = threadIdx.x;
    // sigma_mode 0 / 1: `wmse` holds wmse, sigma the (log) value; >= 2 (coded / rmse sigma): `wmse` holds the per-sample
    // MSE already and sigma[0] the rms value to report
    const float sg = sigma_is_log == 1 ? __expf(sigma[0]) : sigma[0];
    const float mse_scale = sigma_is_log >= 2 ? 1.f : sg * sg;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = tid; i < N; i += blockDim.x) a += wmse[i];
    for (int i = tid; i < Nz; i += blockDim.x) { b += zdist[i]; c += var_kl[i]; }
    a = block_sum(a, red); b = block_sum(b, red); c = block_sum(c, red);
    float msq = 0.f, mx = 0.f;
    if (means) {
        for (int i = tid; i < C * K; i += blockDim.x) msq += means[i] * means[i];
        msq = block_sum(msq, red);
    }
    float cap = 0.f, dmin = INFINITY;
    if (means) {
        // all C*C pairs spread over the block; per-row sums of exp(-d^2/4) accumulate in LDS (C <= 1024 rows).
        // The dictionary itself is staged in LDS when it fits (C*K <= 12288 floats), with an odd row pitch.
        __shared__ float rowsum[1024];
        __shared__ float dict_lds[12288 + 1024];
        const int KP = K | 1;
        const bool in_lds = (long)C * KP <= 12288 + 1024;
        for (int r = tid; r < C; r += blockDim.x) rowsum[r] = 0.f;
        if (in_lds)
            for (int i = tid; i < C * K; i += blockDim.x) dict_lds[(i / K) * KP + (i % K)] = means[i];
        __syncthreads();
        const float* mm = in_lds ? dict_lds : means;
        const int pitch = in_lds ? KP : K;
        for (int pr = tid; pr < C * C; pr += blockDim.x) {
            const int r = pr / C, q = pr % C;
            float d2 = 0.f;
            for (int k = 0; k < K; ++k) { const float t = mm[r * pitch + k] - mm[q * pitch + k]; d2 += t * t; }
            atomicAdd(&rowsum[r], __expf(-d2 / 4.f));
            if (q != r) dmin = fminf(dmin, sqrtf(d2));
            if (q == r) {                                    // row norm once per row
                float nr = 0.f;
                for (int k = 0; k < K; ++k) nr += mm[r * pitch + k] * mm[r * pitch + k];
                mx = fmaxf(mx, sqrtf(nr));
            }
        }
        __syncthreads();
        for (int r = tid; r < C; r += blockDim.x) cap += __logf(rowsum[r]);
        cap = block_sum(cap, red);
        float vmax = wave_max(mx), vmin = -wave_max(-dmin);
        __shared__ float wred[16];
        __syncthreads();
        if ((tid & 63) == 0) { wred[tid >> 6] = vmax; wred[8 + (tid >> 6)] = vmin; }
        __syncthreads();
        if (tid == 0) {
            const int nw = (blockDim.x + 63) >> 6;
            for (int w = 0; w < nw; ++w) { mx = fmaxf(mx, wred[w]); dmin = fminf(dmin, wred[8 + w]); }
        }
    }
    if (tid == 0) {
        const float mse = a / N * mse_scale;
        out[0] = sg;
        out[1] = sumsq_x[0] / nx;
        out[2] = mse;
        out[3] = sqrtf(mse);
        out[4] = b / Nz;
        out[5] = c / Nz;
        out[6] = means ? msq / (C * K) : 0.f;
        out[7] = means ? __logf((float)C) - cap / C : 0.f;
        out[8] = means ? fminf(dmin, 2.f * mx) : 0.f;          // (cdist + 2 max|m| I).min(), layers.py:338-348
        out[9] = flag ? (float)flag[0] : 0.f;
        // running means over the batches seen so far (cvae.py:689-699,720-724): out[10..15] =
        // xpow, mse, rmse, dB, zdist, var_kl;  prev = the previous call's `out` (or NULL at batch 0)
        const float nb = (float)batch, inv = 1.f / (nb + 1.f);
        const float pxp = (prev && batch > 0) ? prev[10] : 0.f, pms = (prev && batch > 0) ? prev[11] : 0.f;
        const float pzd = (prev && batch > 0) ? prev[14] : 0.f, pvk = (prev && batch > 0) ? prev[15] : 0.f;
        const float rxp = (pxp * nb + out[1]) * inv, rms = (pms * nb + mse) * inv;
        out[10] = rxp;
        out[11] = rms;
        out[12] = sqrtf(rms);
        out[13] = 10.f * log10f(rxp / rms);
        out[14] = (pzd * nb + out[4]) * inv;
        out[15] = (pvk * nb + out[5]) * inv;
    }
}

inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ---- importance-weighted bound of the evaluation path (cvae.py:672-676,793-873) -------------------------------------
// rows[l][n] = log p(x | z_l) - log q(z_l | x) = -D/2 (wmse_s + 2 log sigma + log 2pi) + (|eps_l|^2 + sum_k log_var)/2
//              + K/2 log 2pi : one wave per (l, n), lanes over the latent dimension
__global__ __launch_bounds__(256) void iws_rows_kernel(const float* __restrict__ wmse_s, const float* __restrict__ eps,
                                                       const float* __restrict__ log_var, const float* __restrict__ sigma,
                                                       int sigma_is_log, int L, int N, int K, float D, float* __restrict__ rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= L * N) return;
    const int n = row % N;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float e = eps[(long)row * K + k];
        s += fmaf(e, e, log_var[(long)n * K + k]);
    }
    s = wave_sum(s);
    if (lane == 0) {
        float w = wmse_s[row], log_sigma;
        if (sigma_is_log == SIG_RMSE) {        // sigma_n^2 = mean over the draws of the sample's mse
            float m = 0.f;
            for (int l = 0; l < L; ++l) m += wmse_s[(long)l * N + n];
            m /= L;
            w /= m;
            log_sigma = 0.5f * logf(m);
        } else {
            const float sg = sigma[sigma_is_log == SIG_CODED ? n : 0];
            log_sigma = sigma_is_log == SIG_VALUE ? logf(sg) : sg;
        }
        const float LOG2PI = 1.8378770664093453f;
        rows[row] = -0.5f * D * (w + 2.f * log_sigma + LOG2PI) + 0.5f * s + 0.5f * (float)K * LOG2PI;
    }
}

// iws[c][n] = mean_l exp(li - m) + m,  li = rows[l][n] + log_pz[l][c][n],  m = max_l li   (sic: cvae.py:868 adds m to the
// mean, not its logarithm); thread per (c, n), consecutive threads along n
__global__ __launch_bounds__(256) void iws_fold_kernel(const float* __restrict__ rows, const float* __restrict__ log_pz,
                                                       int L, int C, int N, float* __restrict__ iws) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * N) return;
    const int n = (int)(i % N);
    float m = -INFINITY;
    for (int l = 0; l < L; ++l) m = fmaxf(m, rows[(long)l * N + n] + log_pz[(long)l * C * N + i]);
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += expf(rows[(long)l * N + n] + log_pz[(long)l * C * N + i] - m);
    iws[i] = s / (float)L + m;
}

// ---- nn.Dropout of the dense trunks (layers.py:287-288, cvae.py:297-298) -----------------------------------------------
// keep(i) = hash(seed, i) >= p * 2^32 (counter-based: forward and backward regenerate the same mask, nothing is stored);
// y = keep ? x / (1 - p) : 0.  The stream of random bits is this kernel's own (no parity with torch's Philox draws:
// the reference draws them from the global generator, so only the distribution is comparable).
__device__ __forceinline__ unsigned dropout_hash(unsigned long long seed, unsigned long long i) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);           // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32);
}

// seed_dev (may be null): the seed lives in device memory (advanced by the caller on the stream: graph-capturable, no host
// read-back); the mask seed is seed + *seed_dev
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                                                      unsigned long long seed, const long long* __restrict__ seed_dev) {
    if (seed_dev) seed += (unsigned long long)seed_dev[0];
    const unsigned thr = (unsigned)fminf(p * 4294967296.f, 4294967295.f);
    const float scale = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = dropout_hash(seed, (unsigned long long)i) >= thr ? x[i] * scale : 0.f;
}

}  // namespace

extern "C" {

int jvae_recon_fwd_f32(const float* x_reco, const float* x, const float* sigma, int sigma_is_log,
                       float* wmse, int L, int N, int D, void* stream) {
    if (!x_reco || !x || !sigma || !wmse || L < 0 || N < 0 || D <= 0 || sigma_is_log < 0 || sigma_is_log > 3) return JVAE_EINVAL;
    if (N == 0 || L == 0) return 0;
    hipLaunchKernelGGL(recon_fwd_kernel, dim3(N, L), dim3(256), 0, (hipStream_t)stream, x_reco, x, sigma, sigma_is_log,
                       wmse, L, N, D, 1);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// mse_loss(x_output, x_target, batch_mean=False) of module/losses.py:8-27 on EVERY row: rows (L, N, D), x (N, D) ->
// wmse (L, N) = mean_D((rows[l][n] - x[n])^2).  The same kernels as jvae_recon_*, told that there is no mean-path row in front
// (the Python side used to pass `rows - N*D` to the (L+1)-row entry point: an address in front of the allocation).
int jvae_mse_rows_fwd_f32(const float* rows, const float* x, float* wmse, int L, int N, int D, void* stream) {
    if (!rows || !x || !wmse || L < 0 || N < 0 || D <= 0) return JVAE_EINVAL;
    if ((D & 3) == 0 && (((uintptr_t)rows | (uintptr_t)x) & 15)) return JVAE_EINVAL;      // 16-byte loads
    if (N == 0 || L == 0) return 0;
    hipLaunchKernelGGL(recon_fwd_kernel, dim3(N, L), dim3(256), 0, (hipStream_t)stream, rows, x, (const float*)nullptr, SIG_VALUE,
                       wmse, L, N, D, 0);                                                   // sigma NULL: 1
    JVAE_LAUNCH_CHECK();
    return 0;
}

// g_rows (L, N, D) = g_wmse[l][n] * 2 (rows - x) / D
int jvae_mse_rows_bwd_f32(const float* rows, const float* x, const float* g_wmse, float* g_rows, int L, int N, int D, void* stream) {
    if (!rows || !x || !g_wmse || !g_rows || L < 0 || N < 0 || D <= 0) return JVAE_EINVAL;
    if ((D & 3) == 0 && (((uintptr_t)rows | (uintptr_t)x | (uintptr_t)g_rows) & 15)) return JVAE_EINVAL;
    if (N == 0 || L == 0) return 0;
    hipLaunchKernelGGL(recon_bwd_kernel, dim3(N, L), dim3(256), 0, (hipStream_t)stream, rows, x, (const float*)nullptr, SIG_VALUE,
                       (const float*)nullptr, g_wmse, (const float*)nullptr, g_rows, (float*)nullptr, L, N, D, 1);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// gsigma (may be null): modes 0 / 1: 1 float = sum over (l,n) of g * dwmse/dsigma; mode 2: N floats (per-sample log sigma);
// mode 3: must be null.  ws: L*N floats when gsigma != null.  sigma_fwd: see recon_bwd_kernel (mode 0, may be null).
int jvae_recon_bwd_f32(const float* x_reco, const float* x, const float* sigma, int sigma_is_log, const float* sigma_fwd,
                       const float* g_wmse, const float* wmse, float* g_x_reco, float* gsigma, int accumulate_sigma,
                       int L, int N, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!x_reco || !x || !sigma || !g_wmse || !wmse || !g_x_reco || L < 0 || N < 0 || D <= 0) return JVAE_EINVAL;
    if (sigma_is_log < 0 || sigma_is_log > 3 || (sigma_is_log == SIG_RMSE && gsigma)) return JVAE_EINVAL;
    if (gsigma && (!ws || ws_bytes < sizeof(float) * (size_t)L * N)) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(recon_bwd_kernel, dim3(N, L + 1), dim3(256), 0, st, x_reco, x, sigma, sigma_is_log, sigma_fwd, g_wmse,
                       wmse, g_x_reco, gsigma ? (float*)ws : nullptr, L, N, D, 0);
    JVAE_LAUNCH_CHECK();
    if (gsigma && sigma_is_log == SIG_CODED) {
        hipLaunchKernelGGL(rows_fold_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, (const float*)ws, gsigma, L, N, accumulate_sigma);
        JVAE_LAUNCH_CHECK();
    } else if (gsigma) {
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
    if (!x || !y || n < 0 || kind < 0 || kind > 3) return JVAE_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, kind);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_act_bwd_f32(const float* dy, const float* y, float* dx, long n, int kind, void* stream) {
    if (!dy || !y || !dx || n < 0 || kind < 0 || kind > 3) return JVAE_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, kind);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_elbo_fwd_f32(const float* wmse_s, const float* kl, const float* ce, const float* sigma, int sigma_is_log,
                      float* wmse, float* cross_x, float* total, float* mse, int L, int N, int D, float beta, float cw,
                      void* stream) {
    if (!wmse_s || !kl || !sigma || !wmse || !cross_x || !total || L < 1 || N < 0 || D <= 0) return JVAE_EINVAL;
    if (sigma_is_log < 0 || sigma_is_log > 3) return JVAE_EINVAL;
    if (N == 0) return 0;
    hipLaunchKernelGGL(elbo_fwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, wmse_s, kl, ce, sigma,
                       sigma_is_log, wmse, cross_x, total, mse, L, N, (float)D, beta, cw);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// gsigma (may be null): modes 0 / 1: 1 float = sum_n (g_cx + g_tot) * D [/ sigma]; mode 2: N floats, ADDED to when
// accumulate_sigma; mode 3: must be null and `sigma` points to the forward's wmse_s (L,N).  ws: N floats when gsigma != null
int jvae_elbo_bwd_f32(const float* g_wmse, const float* g_cx, const float* g_tot, const float* sigma, int sigma_is_log,
                      float* g_wmse_s, float* g_kl, float* g_ce, float* gsigma, int accumulate_sigma,
                      int L, int N, int D, float beta, float cw, void* ws, size_t ws_bytes, void* stream) {
    if (!sigma || !g_wmse_s || L < 1 || N < 0 || D <= 0) return JVAE_EINVAL;
    if (sigma_is_log < 0 || sigma_is_log > 3 || (sigma_is_log == SIG_RMSE && gsigma)) return JVAE_EINVAL;
    if (gsigma && (!ws || ws_bytes < sizeof(float) * (size_t)N)) return JVAE_EWORKSPACE;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(elbo_bwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, g_wmse, g_cx, g_tot, sigma, sigma_is_log,
                       g_wmse_s, g_kl, g_ce, gsigma ? (float*)ws : nullptr, L, N, (float)D, beta, cw);
    JVAE_LAUNCH_CHECK();
    if (gsigma && sigma_is_log == SIG_CODED) {
        hipLaunchKernelGGL(rows_fold_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, (const float*)ws, gsigma, 1, N, accumulate_sigma);
        JVAE_LAUNCH_CHECK();
    } else if (gsigma) {
        hipLaunchKernelGGL(vec_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, gsigma, N, accumulate_sigma);
        JVAE_LAUNCH_CHECK();
    }
    return 0;
}

// forward (x -> y) and backward (dy -> dx) are the same masked scaling for the same seed
int jvae_dropout_f32(const float* x, float* y, long n, float p, long seed, void* stream) {
    if (n < 0 || !(p >= 0.f && p < 1.f) || (n > 0 && (!x || !y))) return JVAE_EINVAL;
    if (n == 0) return 0;
    long b = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream, x, y, n, p,
                       (unsigned long long)seed, (const long long*)nullptr);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_dropout_dev_f32(const float* x, float* y, long n, float p, const long long* seed_dev, long salt, void* stream) {
    if (n < 0 || !(p >= 0.f && p < 1.f) || !seed_dev || (n > 0 && (!x || !y))) return JVAE_EINVAL;
    if (n == 0) return 0;
    long b = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream, x, y, n, p,
                       (unsigned long long)salt, seed_dev);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_iws_f32(const float* wmse_s, const float* eps, const float* log_var, const float* log_pz, const float* sigma,
                 int sigma_is_log, int L, int N, int K, int C, int D, float* rows, float* iws, void* stream) {
    if (!wmse_s || !eps || !log_var || !log_pz || !sigma || !rows || !iws || L < 1 || N < 0 || K < 1 || C < 1 || D <= 0)
        return JVAE_EINVAL;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(iws_rows_kernel, dim3(cdiv((long)L * N, 4)), dim3(256), 0, st, wmse_s, eps, log_var, sigma, sigma_is_log,
                       L, N, K, (float)D, rows);
    JVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(iws_fold_kernel, dim3(cdiv((long)C * N, 256)), dim3(256), 0, st, (const float*)rows, log_pz, L, C, N, iws);
    JVAE_LAUNCH_CHECK();
    return 0;
}

// out: 16 floats on the device (layout: see measures_kernel); prev: the previous batch's `out` or NULL.  sumsq_x: device scalar = sum(x^2) (jvae_sqnorm_accum_f32).
int jvae_measures_f32(const float* sumsq_x, long nx, const float* wmse, const float* zdist, const float* var_kl, int N, int Nz,
                      const float* sigma, int sigma_is_log, const float* means, int C, int K, const int* flag,
                      const float* prev, int batch, float* out, void* stream) {
    if (!sumsq_x || !wmse || !zdist || !var_kl || !sigma || !out || N <= 0 || Nz <= 0 || nx <= 0) return JVAE_EINVAL;
    if (means && (C <= 0 || K <= 0 || C > 1024)) return JVAE_EINVAL;
    hipLaunchKernelGGL(measures_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sumsq_x, (float)nx, wmse, zdist, var_kl,
                       N, Nz, sigma, sigma_is_log, means, C, K, flag, prev, batch, out);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
