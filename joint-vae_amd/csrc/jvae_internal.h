// Internal (non-ABI) declarations shared between the translation units of libjvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>

// C[b](m,n) (+)= sum_k A[b](m,k) B[b](k,n); element strides in floats.  flags: 1 accumulate, 2 ReLU.
// splitk > 1 adds partial products with float atomics: C must then hold the value to accumulate onto
// (zeros for a plain product).  bias_mode: 0 none, 1 per-n, 2 per-m.
int jvae_gemm_launch(int M, int N, int K, int batch,
                     const float* A, long sAm, long sAk, long sAb,
                     const float* B, long sBk, long sBn, long sBb,
                     float* C, long sCm, long sCn, long sCb,
                     const float* bias, int bias_mode, int flags, int splitk, hipStream_t st);

int jvae_gemm_launch_ex(int M, int N, int K, int batch,
                        const float* A, long sAm, long sAk, long sAb,
                        const float* B, long sBk, long sBn, long sBb,
                        float* C, long sCm, long sCn, long sCb,
                        const float* bias, int bias_mode, int bias_div, int flags, int splitk, hipStream_t st);

// y[i] = [relu]([y[i] +] bias[i % N] + sum_s part[s][i]): fixed-order fold of S partial products (gemm.hip)
// bias index = (i / bias_div) % N
int jvae_splitk_fold_qn(const float* part, const float* bias, float* y, int S, int N, int Cs, int Ps, hipStream_t st);
int jvae_splitk_fold(const float* part, const float* bias, float* y, int S, long MN, int N, int relu, int accumulate,
                     hipStream_t st, int bias_div = 1);

// deterministic split-K: partial products stored side by side (gemm.hip); *splits receives the number of pieces
int jvae_gemm_launch_part(int M, int N, int K, int batch,
                          const float* A, long sAm, long sAk, long sAb,
                          const float* B, long sBk, long sBn, long sBb,
                          float* part, long sCm, long sCn, long sCb, long sCsplit, int want_splits, int* splits,
                          hipStream_t st);

// Optional per-input-channel transform applied while a convolution stages its INPUT: a = [relu](x*sc[c] + sh[c]).
// This is the BatchNorm(+ReLU) that produced the layer input, deferred into the consumer so that the normalised
// activation is never written to / re-read from HBM (sc == nullptr: identity).
struct InAff { const float* sc; const float* sh; int relu; };

// Geometry of one (transposed) convolution.  "big" side = the tensor that is unfolded (conv input /
// transposed-conv output), "small" side = the tensor on the folded grid (conv output / transposed-conv input).
struct ConvGeom {
    int N;            // images
    int Cb, Hb, Wb;   // big side:   channels, height, width
    int Cs, Hs, Ws;   // small side: channels, height, width
    int KH, KW, S, P; // kernel, stride, padding (as in nn.Conv2d / nn.ConvTranspose2d)
};
