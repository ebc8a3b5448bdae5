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
size_t jvae_channel_sum_ws_bytes(int C);
int jvae_channel_sum(const float* t, float* out, int N, int C, int P, int accumulate, float* ws, size_t ws_bytes,
                     hipStream_t st);

// conv_mfma.hip: implicit-GEMM 5x5 kernels (forward-type)
bool jvae_conv5_fwd_ok(int Cin, int H, int W, int Cout, int OH, int OW, int S, int P);
size_t jvae_conv5_pack_floats(int Cin, int Cout);
int jvae_conv5_fwd(const float* in, const float* w, int swap, int flip, const float* bias, float* out,
                   int N, int Cin, int H, int W, int Cout, int OW, int S, int P, float* ws, hipStream_t st,
                   float* stats = nullptr, int* nsplit = nullptr, const InAff* aff = nullptr);
int jvae_conv5_fwd_max_splits(int N, int OW);

int jvae_conv5_pack(const float* w, float* wp, int C, int O, int swap, int flip, hipStream_t st);

// conv_x3.hip: the same operator on the bf16 matrix cores, every fp32 operand split exactly into three bf16 terms
bool jvae_conv5_x3_ok(int Cin, int H, int W, int Cout, int OH, int OW, int S, int P);
size_t jvae_conv5_x3_pack_bytes(int Cin, int Cout);
int jvae_conv5_x3_set(int mode);
int jvae_conv5_x3_set_shape16(int on);
bool jvae_conv5_x3_enabled();
int jvae_conv5_x3_wpack(const float* w, float* ws, int C, int O, int swap, int flip, hipStream_t st);
// conv_t2_x3.hip: the 4-phase stride-2 transposed convolution in the same arithmetic (w = raw weight, [c][o][tap])
bool jvae_convt2_x3_ok(int N, int C, int WS, int O);
int jvae_convt2_x3(const float* in, const float* w, const float* bias, float* out, int N, int C, int WS, int O, float* ws,
                   hipStream_t st, float* stats = nullptr, int* nsplit = nullptr, const InAff* aff = nullptr);
int jvae_conv5_x3_fwd(const float* in, const float* w, int swap, int flip, const float* bias, float* out,
                      int N, int Cin, int H, int W, int Cout, int OW, int S, int P, float* ws, hipStream_t st,
                      float* stats = nullptr, int* nsplit = nullptr, const InAff* aff = nullptr);

// conv_t2_mfma.hip: stride-2 transposed 5x5 (4-phase): small (C,HS,WS) -> big (O,2HS,2WS), wpacked = (C,25,O)
bool jvae_convt2_ok(int C, int HS, int WS, int O, int HB, int WB, int KH, int KW, int S, int P);
int jvae_convt2(const float* in, const float* wpacked, const float* bias, float* out, int N, int C, int WS, int O,
                hipStream_t st, float* stats = nullptr, int* nsplit = nullptr, const InAff* aff = nullptr);

// conv_smallco.hip: 5x5 stride-1 'same' convolution with <= 4 output channels (vector ALUs)
bool jvae_conv5_smallco_ok(int Cin, int H, int W, int Cout, int KH, int KW, int S, int P);
int jvae_conv5_smallco(const float* in, const float* w, const float* bias, float* out, int N, int Cin, int W, int Cout,
                       hipStream_t st, const InAff* aff = nullptr);

// ... and with <= 4 INPUT channels (forward-type operator, any weight role: the first layer's forward, the head's dgrad)
bool jvae_conv5_smallci_ok(int Cin, int H, int W, int Cout, int OW, int S, int P, bool dgrad_role);
int jvae_conv5_smallci(const float* in, const float* w, int swap, int flip, const float* bias, float* out,
                       int N, int Cin, int W, int Cout, float* ws, hipStream_t st, float* stats = nullptr, int* nsplit = nullptr);

// conv_wgrad_mfma.hip: dW[a][b][tap] = sum Ps[n][a][u][v] Q[n][b][u*S+kh-P][v*S+kw-P]
bool jvae_conv5_wgrad_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P);
size_t jvae_conv5_wgrad_ws_floats(int N, int Ca, int Cb, int S, int WS);
int jvae_conv5_wgrad(const float* ps, const float* q, float* dw, int accumulate, int swapflip,
                     int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                     const InAff* aff_p = nullptr, const InAff* aff_q = nullptr);

// conv_wgrad_x3.hip: the same operator on the bf16 matrix cores (3-way exact operand split), WS in {8, 16, 32}
bool jvae_conv5_wgrad_x3_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P);
size_t jvae_conv5_wgrad_x3_ws_floats(int N, int Ca, int Cb, int S);
int jvae_conv5_wgrad_x3(const float* ps, const float* q, float* dw, int accumulate, int swapflip,
                        int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                        const InAff* aff_p = nullptr, const InAff* aff_q = nullptr);
int jvae_wgrad_slab_reduce(const float* slab, float* dw, int G, int Ca, int Cb, int accumulate, int swapflip, hipStream_t st,
                           int tapmajor = 0);

// conv_dispatch.hip
size_t jvae_conv_ws(const ConvGeom& g, int transposed);
// aff (input transform, see InAff): only the implicit kernels apply it; JVAE_ENOTSUP otherwise (jvae_conv_affine_ok)
int jvae_conv_fwd(const ConvGeom& g, int transposed, const float* x, const float* w, const float* bias, float* y,
                  float* ws, size_t ws_bytes, hipStream_t st, float* stats = nullptr, int* nsplit = nullptr,
                  const InAff* aff = nullptr);
bool jvae_conv_affine_ok(const ConvGeom& g, int transposed);
int jvae_conv_stats_splits(const ConvGeom& g, int transposed);
int jvae_conv_dgrad(const ConvGeom& g, int transposed, const float* dy, const float* w, float* dx,
                    float* ws, size_t ws_bytes, hipStream_t st);
int jvae_conv_wgrad(const ConvGeom& g, int transposed, const float* x, const float* dy, float* dw,
                    float* ws, size_t ws_bytes, hipStream_t st, const InAff* aff = nullptr);
