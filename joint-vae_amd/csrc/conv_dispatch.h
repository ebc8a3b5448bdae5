// Kernel selection for one (transposed) convolution: fast implicit kernels where the geometry matches,
// the generic unfold+GEMM path otherwise.  x = layer input, y = layer output, w = the layer's weight in
// its PyTorch layout ([Cout][Cin][KH][KW] for Conv2d, [Cin][Cout][KH][KW] for ConvTranspose2d).
#pragma once
#include "jvae_internal.h"

// conv_generic.hip
size_t jvae_conv_generic_ws(const ConvGeom& g);
int jvae_fold_fwd(const ConvGeom& g, const float* xb, const float* w, const float* bias, float* ys,
                  float* ws, size_t ws_bytes, hipStream_t st);
int jvae_fold_bwd(const ConvGeom& g, const float* ys, const float* w, const float* bias, float* xb,
                  float* ws, size_t ws_bytes, hipStream_t st);
int jvae_fold_wgrad(const ConvGeom& g, const float* xb, const float* ys, float* dw,
                    float* ws, size_t ws_bytes, hipStream_t st);
int jvae_channel_sum(const float* t, float* out, int N, int C, int P, int accumulate, hipStream_t st);

// conv_dispatch.hip
size_t jvae_conv_ws(const ConvGeom& g, int transposed);
int jvae_conv_fwd(const ConvGeom& g, int transposed, const float* x, const float* w, const float* bias, float* y,
                  float* ws, size_t ws_bytes, hipStream_t st);
int jvae_conv_dgrad(const ConvGeom& g, int transposed, const float* dy, const float* w, float* dx,
                    float* ws, size_t ws_bytes, hipStream_t st);
int jvae_conv_wgrad(const ConvGeom& g, int transposed, const float* x, const float* dy, float* dw,
                    float* ws, size_t ws_bytes, hipStream_t st);
