// MaxPool2d / AvgPool2d / UpsamplingNearest2d: the `M`, `A` and `U` tokens of the layer DSL (reference:
// module/vae_layers/conv.py:201-212, used by the vgg* / ivgg* entries of conv-models.ini:13-18,28-30).
//
// HBM-bound element-wise gathers on (planes = N*C, H, W) fp32 NCHW tensors: one thread per output element, consecutive
// threads along W (coalesced), no atomics - the backward kernels gather over the windows that cover an input pixel in
// the order PyTorch's scatter visits them (output rows, then columns), so the sums are reproducible.
//   max : y = max over the KxK window (padding cells = -inf), idx = h*W + w of the FIRST maximum in row-major window
//         order (`val > max || isnan(val)`, as at::native max_pool2d); backward routes dy to idx.
//   avg : count_include_pad = True (nn.AvgPool2d default): divisor K*K.
//   nearest up-sampling by an integer factor: y[oy][ox] = x[oy / s][ox / s]; backward = s x s block sums.
#include <math.h>
#include "common.h"
#include "jvae_hip.h"

namespace {

__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int* __restrict__ idx, long planes, int H, int W, int OH, int OW,
                                                       int K, int S, int P, int mode) {
    const long total = planes * OH * OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const long pl = i / ((long)OW * OH);
        const float* xp = x + pl * H * W;
        const int h0 = oy * S - P, w0 = ox * S - P;
        const int hs = max(h0, 0), ws = max(w0, 0), he = min(h0 + K, H), we = min(w0 + K, W);
        if (mode == 0) {
            float best = -INFINITY;
            int bi = hs * W + ws;
            for (int h = hs; h < he; ++h)
                for (int w = ws; w < we; ++w) {
                    const float v = xp[h * W + w];
                    if (v > best || isnan(v)) { best = v; bi = h * W + w; }
                }
            y[i] = best;
            idx[i] = bi;
        } else {
            float s = 0.f;
            for (int h = hs; h < he; ++h)
                for (int w = ws; w < we; ++w) s += xp[h * W + w];
            y[i] = s / (float)(K * K);
        }
    }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                       float* __restrict__ dx, long planes, int H, int W, int OH, int OW,
                                                       int K, int S, int P, int mode) {
    const long total = planes * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        const long pl = i / ((long)W * H);
        // outputs whose window [o*S - P, o*S - P + K) contains the pixel
        const int oy0 = max(0, (h + P - K + S) / S), oy1 = min(OH - 1, (h + P) / S);
        const int ox0 = max(0, (w + P - K + S) / S), ox1 = min(OW - 1, (w + P) / S);
        const float* gp = dy + pl * OH * OW;
        float s = 0.f;
        if (mode == 0) {
            const int* ip = idx + pl * OH * OW;
            const int me = h * W + w;
            for (int oy = oy0; oy <= oy1; ++oy)
                for (int ox = ox0; ox <= ox1; ++ox)
                    if (ip[oy * OW + ox] == me) s += gp[oy * OW + ox];
        } else {
            const float inv = (float)(K * K);
            for (int oy = oy0; oy <= oy1; ++oy)
                for (int ox = ox0; ox <= ox1; ++ox) s += gp[oy * OW + ox] / inv;
        }
        dx[i] = s;
    }
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long planes, int H, int W, int sc) {
    const int OH = H * sc, OW = W * sc;
    const long total = planes * OH * OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const long pl = i / ((long)OW * OH);
        y[i] = x[(pl * H + oy / sc) * W + ox / sc];
    }
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           long planes, int H, int W, int sc) {
    const int OW = W * sc;
    const long total = planes * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        const long pl = i / ((long)W * H);
        const float* gp = dy + (pl * H * sc + (long)h * sc) * OW + (long)w * sc;
        float s = 0.f;
        for (int a = 0; a < sc; ++a)
            for (int b = 0; b < sc; ++b) s += gp[(long)a * OW + b];
        dx[i] = s;
    }
}

inline int blocks_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

inline bool pool_args_ok(long planes, int H, int W, int K, int S, int P) {
    return planes >= 0 && H >= 1 && W >= 1 && K >= 1 && S >= 1 && P >= 0 && 2 * P <= K && H + 2 * P >= K && W + 2 * P >= K;
}

}  // namespace

extern "C" {

int jvae_pool2d_out_shape(int H, int W, int K, int S, int P, int* OH, int* OW) {
    if (!pool_args_ok(1, H, W, K, S, P)) return JVAE_EINVAL;
    *OH = (H + 2 * P - K) / S + 1;
    *OW = (W + 2 * P - K) / S + 1;
    return 0;
}

int jvae_pool2d_fwd_f32(const float* x, float* y, int* idx, long planes, int H, int W, int K, int S, int P, int mode,
                        void* stream) {
    if (!pool_args_ok(planes, H, W, K, S, P) || (mode != 0 && mode != 1) || !x || !y || (mode == 0 && !idx)) return JVAE_EINVAL;
    const int OH = (H + 2 * P - K) / S + 1, OW = (W + 2 * P - K) / S + 1;
    const long total = planes * OH * OW;
    if (total == 0) return 0;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, idx, planes, H, W,
                       OH, OW, K, S, P, mode);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_pool2d_bwd_f32(const float* dy, const int* idx, float* dx, long planes, int H, int W, int K, int S, int P,
                        int mode, void* stream) {
    if (!pool_args_ok(planes, H, W, K, S, P) || (mode != 0 && mode != 1) || !dy || !dx || (mode == 0 && !idx)) return JVAE_EINVAL;
    const int OH = (H + 2 * P - K) / S + 1, OW = (W + 2 * P - K) / S + 1;
    const long total = planes * H * W;
    if (total == 0) return 0;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, planes, H, W,
                       OH, OW, K, S, P, mode);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_upsample_nearest_fwd_f32(const float* x, float* y, long planes, int H, int W, int scale, void* stream) {
    if (planes < 0 || H < 1 || W < 1 || scale < 1 || !x || !y) return JVAE_EINVAL;
    const long total = planes * H * W * scale * scale;
    if (total == 0) return 0;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, scale);
    JVAE_LAUNCH_CHECK();
    return 0;
}

int jvae_upsample_nearest_bwd_f32(const float* dy, float* dx, long planes, int H, int W, int scale, void* stream) {
    if (planes < 0 || H < 1 || W < 1 || scale < 1 || !dy || !dx) return JVAE_EINVAL;
    const long total = planes * H * W;
    if (total == 0) return 0;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, planes, H, W, scale);
    JVAE_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
