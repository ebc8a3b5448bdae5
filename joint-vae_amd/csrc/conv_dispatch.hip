// Direction -> primitive mapping (see conv_generic.hip header) and fast-path selection.
#include "common.h"
#include "conv_dispatch.h"

size_t jvae_conv_ws(const ConvGeom& g, int transposed) {
    (void)transposed;
    return jvae_conv_generic_ws(g);
}

int jvae_conv_fwd(const ConvGeom& g, int transposed, const float* x, const float* w, const float* bias, float* y,
                  float* ws, size_t ws_bytes, hipStream_t st) {
    if (!transposed) return jvae_fold_fwd(g, x, w, bias, y, ws, ws_bytes, st);
    return jvae_fold_bwd(g, x, w, bias, y, ws, ws_bytes, st);
}

int jvae_conv_dgrad(const ConvGeom& g, int transposed, const float* dy, const float* w, float* dx,
                    float* ws, size_t ws_bytes, hipStream_t st) {
    if (!transposed) return jvae_fold_bwd(g, dy, w, nullptr, dx, ws, ws_bytes, st);
    return jvae_fold_fwd(g, dy, w, nullptr, dx, ws, ws_bytes, st);
}

int jvae_conv_wgrad(const ConvGeom& g, int transposed, const float* x, const float* dy, float* dw,
                    float* ws, size_t ws_bytes, hipStream_t st) {
    if (!transposed) return jvae_fold_wgrad(g, x, dy, dw, ws, ws_bytes, st);
    return jvae_fold_wgrad(g, dy, x, dw, ws, ws_bytes, st);
}
