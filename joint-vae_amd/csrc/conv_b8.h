// bf16 "B8" activation path: internal declarations (see conv_b8.hip for the layout).
#pragma once
#include "jvae_internal.h"

// Deferred BatchNorm(+ReLU) on a B8 operand (see InAff in jvae_internal.h): sc / sh hold ceil(C/8)*8 floats each (zero for
// padding channels, so they stay exact zeros).
// conv_b8.hip
int jvae_b8_pack(const float* x, void* y, int N, int C, long HW, hipStream_t st);
int jvae_b8_unpack(const void* y, float* x, int N, int C, long HW, int accumulate, hipStream_t st);
int jvae_b8_channel_sum(const void* t, float* out, int N, int C, long HW, int accumulate, float* ws, hipStream_t st);
bool jvae_conv5_b8_fwd_ok(int Cin, int H, int W, int Cout, int OH, int OW, int S, int P);
size_t jvae_conv5_b8_pack_bytes(int Cin, int Cout);
int jvae_conv5_b8_max_splits(int N, int OW);
int jvae_conv5_b8_wpack(const float* w, void* wp, int C, int O, int swap, int flip, hipStream_t st);
int jvae_conv5_b8_fwd(const void* in, const float* w, int swap, int flip, const float* bias, void* out, int out_f32,
                      int N, int Cin, int H, int W, int Cout, int OW, int S, int P, void* ws, hipStream_t st,
                      float* stats = nullptr, int* nsplit = nullptr, const InAff* aff = nullptr);

// conv_t2_b8.hip: stride-2 transposed 5x5 (4-phase), small (C,WS,WS) -> big (O,2WS,2WS)
bool jvae_convt2_b8_ok(int C, int HS, int WS, int O, int HB, int WB, int KH, int KW, int S, int P);
int jvae_convt2_b8(const void* in, const float* w, const float* bias, void* out, int N, int C, int WS, int O,
                   void* ws, hipStream_t st, float* stats = nullptr, int* nsplit = nullptr, const InAff* aff = nullptr);

// conv_wgrad_mfma.hip
int jvae_wgrad_slab_reduce(const float* slab, float* dw, int G, int Ca, int Cb, int accumulate, int swapflip, hipStream_t st,
                           int tapmajor = 0);

// conv_wgrad_b8.hip: dW[a][b][tap] = sum Ps[n][a][u][v] Q[n][b][u*S+kh-P][v*S+kw-P] from B8 tensors
bool jvae_conv5_wgrad_b8_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P);
size_t jvae_conv5_wgrad_b8_ws_floats(int N, int Ca, int Cb);
int jvae_conv5_wgrad_b8(const void* ps, const void* q, float* dw, int accumulate, int swapflip,
                        int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                        const InAff* aff_p = nullptr, const InAff* aff_q = nullptr);

// conv_wgrad_x3.hip: the same operator on the LDS image / pipeline of the split-bf16 weight-gradient kernel, one plane
bool jvae_conv5_wgrad_b8x_ok(int Ca, int HS, int WS, int Cb, int HB, int WB, int S, int P);
size_t jvae_conv5_wgrad_b8x_ws_floats(int N, int Ca, int Cb, int S);
int jvae_conv5_wgrad_b8x(const void* ps, const void* q, float* dw, int accumulate, int swapflip,
                         int N, int Ca, int WS, int Cb, int S, int P, float* ws, hipStream_t st,
                         const InAff* aff_p = nullptr, const InAff* aff_q = nullptr);
