// Input pipeline step in front of the training path: uint8 image batch -> random horizontal flip ->
// edge-padded random crop -> float32 in [0,1], NCHW, in ONE pass on the device.
//
// Replaces the per-sample host-side torchvision chain of the reference (utils/torch_load.py:405-426: for
// data_augmentation = ['flip', 'crop']: transforms.RandomHorizontalFlip(), transforms.RandomCrop(size, padding =
// size // 8, padding_mode='edge'), then transforms.ToTensor() = uint8 HWC -> float CHW / 255) that runs with
// DataLoader(num_workers=0) (cvae.py:2245-2249) and would cap the step at a few thousand images/s.
// The random decisions (flip flag, crop offsets in [0, 2*pad]) are inputs, so the result is a pure, bit-exact function.
#include "common.h"
#include "jvae_internal.h"

namespace {

// out[n][c][y][x] = in[n][ys][xs][c] / 255 with ys = clamp(y + dy[n] - pad), xs = clamp(x + dx[n] - pad), then mirrored
// when flip[n] (flip is applied BEFORE pad + crop, as in the reference's transform order).
__global__ __launch_bounds__(256) void augment_kernel(const unsigned char* __restrict__ in, const unsigned char* __restrict__ flip,
                                                      const int* __restrict__ dy, const int* __restrict__ dx,
                                                      float* __restrict__ out, int N, int C, int H, int W, int pad, int nhwc) {
    const long total = (long)N * C * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        int ys = y + (dy ? dy[n] : pad) - pad;
        int xs = x + (dx ? dx[n] : pad) - pad;
        ys = ys < 0 ? 0 : (ys >= H ? H - 1 : ys);
        xs = xs < 0 ? 0 : (xs >= W ? W - 1 : xs);
        if (flip && flip[n]) xs = W - 1 - xs;
        const long src = nhwc ? (((long)n * H + ys) * W + xs) * C + c : (((long)n * C + c) * H + ys) * W + xs;
        out[i] = (float)in[src] / 255.0f;
    }
}

}  // namespace

extern "C" int jvae_augment_u8_f32(const unsigned char* in, const unsigned char* flip, const int* dy, const int* dx,
                                   float* out, int N, int C, int H, int W, int pad, int nhwc, void* stream) {
    if (!in || !out || N < 0 || C <= 0 || H <= 0 || W <= 0 || pad < 0) return JVAE_EINVAL;
    if (N == 0) return 0;
    const long total = (long)N * C * H * W;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, flip, dy, dx, out,
                       N, C, H, W, pad, nhwc);
    JVAE_LAUNCH_CHECK();
    return 0;
}
