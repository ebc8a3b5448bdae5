/* libjvae_hip.so — C ABI of the MI355X (gfx950) kernels behind the joint-CVAE training step.
 *
 * Drop-in boundary: the reference (moxime/joint-vae) has no FFI of its own — the path sits
 * behind a Python import surface (cvae.py:8-15) and bottoms out in PyTorch ops.  Each entry point below
 * names the PyTorch op / reference lines it replaces; the Python host in joint-vae_amd/ binds them with
 * ctypes (joint-vae_amd/jvae_hip/lib.py) — INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers to DEVICE memory (fp32 unless stated; class labels int64), sizes as int/long,
 *     `stream` is a hipStream_t passed as void*; everything is stream-ordered, nothing synchronises,
 *     nothing allocates: scratch comes from the caller (`ws`, size from the matching *_workspace_bytes).
 *   - tensors are dense, NCHW for images.
 *   - return 0 = ok, <0 = invalid argument (-1), unsupported (-2), workspace too small (-3),
 *     >0 = hipError_t of the failing launch.
 *   - thread-compatible: no global mutable state.
 */
#ifndef JVAE_HIP_H
#define JVAE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* jvae_version(void);

/* ---- dense products -------------------------------------------------------------------------------
 * C[b](m,n) (+)= sum_k A[b](m,k) B[b](k,n) [+ bias] [ReLU]; strides in elements; batch over b.
 * bias_mode 0 none / 1 per-n / 2 per-m.  flags: 1 accumulate into C, 2 ReLU, 4 add with float atomics
 * (C must hold the base value; implied by splitk > 1).
 * Replaces nn.Linear forward/backward (module/vae_layers/layers.py:283-296,380-394; cvae.py:291-301,319-326;
 * Classifier layers.py:456-483) and the GEMM inside every convolution. */
int jvae_gemm_f32(int M, int N, int K, int batch,
                  const float* A, long sAm, long sAk, long sAb,
                  const float* B, long sBk, long sBn, long sBb,
                  float* C, long sCm, long sCn, long sCb,
                  const float* bias, int bias_mode, int flags, int splitk, void* stream);

/* y[i] = [relu]([y[i] +] bias[i % N] + sum_{s<S} part[s][i]), i < MN (bias may be NULL): deterministic fold of a split-K product whose K slices were
 * computed by one batched jvae_gemm_f32 launch (used by the dense heads when M*N gives too few tiles, e.g. 256 x 200
 * with K = 7200 in the 64x64 model). */
int jvae_splitk_fold_f32(const float* part, const float* bias, float* y, int S, long MN, int N, int relu, int accumulate,
                         void* stream);

/* ---- (transposed) convolution ---------------------------------------------------------------------
 * x: (N,Cin,H,W) layer input; y: (N,Cout,OH,OW) layer output; w in the PyTorch layout of the layer kind
 * ((Cout,Cin,KH,KW) for Conv2d, (Cin,Cout,KH,KW) for ConvTranspose2d); S stride, P padding,
 * OP output_padding (transposed only).  Replaces nn.Conv2d / nn.ConvTranspose2d built by
 * build_de_conv_layers (module/vae_layers/conv.py:186-196) and their autograd backward.
 * wgrad: accumulate != 0 adds into dw/dbias (autograd .grad accumulation), else overwrites. */
size_t jvae_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP,
                                   int transposed);
/* Arithmetic unit of the stride-1 5x5 layers with >= 16 input channels (forward, ConvTranspose2d forward, dgrad):
 * mode 1 (default, JVAE_X3=0 in the environment sets 0 at start-up) = bf16 matrix cores with every fp32 operand split
 * EXACTLY into three bf16 terms and six bf16 MFMA products per fp32 product (csrc/conv_x3.hip: fp32-accurate, error
 * below the fp32 FMA chain's own rounding); mode 0 = v_mfma_f32_32x32x2_f32 (csrc/conv_mfma.hip).  Same tensors, same
 * results to fp32 rounding; returns the previous mode.  Same reference ops as jvae_conv2d_fwd_f32. */
int jvae_conv2d_set_split_bf16(int mode);
/* MFMA shape of that split-bf16 kernel on maps up to 32 wide: on = 1 (default) =
 * v_mfma_f32_16x16x32_bf16 with K = 2 taps x 16 channels, on = 0 = v_mfma_f32_32x32x16_bf16 with K = 16 channels of one tap.
 * Same six products per fp32 product, same fp32 accumulation order per output up to the pairing of taps; returns the
 * previous setting (A/B switch for tests and benches). */
int jvae_conv2d_set_split_shape16(int on);
int jvae_conv2d_out_shape(int H, int W, int KH, int KW, int S, int P, int OP, int transposed, int* OH, int* OW);
int jvae_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                        int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                        void* ws, size_t ws_bytes, void* stream);
/* Forward that also emits BatchNorm partial statistics of (y - bias) from the kernel's epilogue when the selected
 * kernel supports it.  stats: (Cout, cap, 2) floats with cap >= jvae_conv2d_stats_splits(...) (0 = this layer never
 * produces them); *nsplit (HOST int) = partials per channel actually written, laid out (Cout, *nsplit, 2). */
int jvae_conv2d_stats_splits(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed);
int jvae_conv2d_fwd_stats_f32(const float* x, const float* w, const float* bias, float* y, float* stats, int* nsplit,
                              int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                              void* ws, size_t ws_bytes, void* stream);
int jvae_conv2d_dgrad_f32(const float* dy, const float* w, float* dx,
                          int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                          void* ws, size_t ws_bytes, void* stream);
int jvae_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                          int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- per-step cache of the re-packed convolution weights (csrc/pack_cache.hip; no reference counterpart: PyTorch re-lays
 * weights out inside its conv library).  The 5x5 kernels read their weights in a packed operand layout that depends on the
 * weights only; the host brackets the span in which the weights are constant (evaluate() ... end of backward of
 * cvae.py:2429-2461): begin() re-packs every registered (weight, layout) pair of `owner` in ONE launch and arms the lookups,
 * the convolution entry points then skip their own pack launch, end() disarms (backward is over / the optimiser is about to
 * change the weights).  Outside a bracket every convolution packs for itself, as without a cache.
 * configure(): caller-owned persistent device buffer (256-byte aligned; NULL = off), shared out in equal regions to at most
 * four owners.  owner: a value that changes whenever the set of weight ADDRESSES in use changes; weights[0..n): those addresses
 * - entries are only ever created for them (any other tensor, e.g. a temporary whose address is re-used later, packs per call).
 * pin(): the armed owner's region is never recycled for another owner (a HIP graph captured in the bracket has its slot
 * addresses baked in).  stats(): host-side counters for tests. */
int jvae_pack_cache_configure(void* buf, size_t bytes);
int jvae_pack_cache_begin(void* stream, long long owner, const void* const* weights, int n);
int jvae_pack_cache_end(void);
int jvae_pack_cache_pin(void);
int jvae_pack_cache_reset(void);
int jvae_pack_cache_stats(int* entries, long long* hits, long long* misses, long long* refreshes);

/* out[c] (+)= sum_{n,q} t[n][c][q]: bias gradient of a conv (P = OH*OW) or linear (P = 1) layer. */
size_t jvae_channel_sum_workspace_bytes(int C);
int jvae_channel_sum_f32(const float* t, float* out, int N, int C, int P, int accumulate, void* ws, size_t ws_bytes,
                         void* stream);

/* ---- BatchNorm2d (+ fused ReLU) -------------------------------------------------------------------
 * x,y,dx,dy: (N,C,P) with P = H*W.  training != 0: batch statistics (biased variance in the forward,
 * unbiased into running_var, momentum update, num_batches_tracked += 1), else running statistics.
 * `relu` (here and as `in_relu` of the convolutions below) names the activation that follows the BatchNorm in the reference's stacks
 * (conv.py:214-220, misc.py:24-27), fused into these kernels: 0 none, 1 nn.ReLU, 2 nn.LeakyReLU() - negative slope 0.01, the
 * reference's activation='leaky' (round 5); any other non-zero value means ReLU.  Backward recomputes the mask from x.  The bf16
 * ("b8") entry points know 0 and 1 only. */
size_t jvae_bn_workspace_bytes(int C);
/* Host-only (no GPU work): launch plan of the BatchNorm kernels - nsplit image parts for the reduction kernels, nchunk for the
 * apply kernels - and the image range [nb, ne) of part j out of `parts` (trailing parts may be EMPTY, nb == ne, never
 * negative: the reference keeps the ragged last batch, cvae.py:2245-2249, so every N must partition safely). */
int jvae_bn_plan(int N, int C, int P, int* nsplit, int* nchunk);
int jvae_image_range(int N, int parts, int j, int* nb, int* ne);
int jvae_bn_fwd_f32(const float* x, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* y, float* save_mean, float* save_invstd,
                    int N, int C, int P, float momentum, float eps, int training, int relu,
                    void* ws, size_t ws_bytes, void* stream);
/* Same as jvae_bn_fwd_f32 with the batch statistics supplied by the producing convolution: ext_stats (C, ext_nsplit, 2)
 * = per-workgroup (sum, sum of squares) of (x - ext_pivot[c]), ext_pivot = the conv bias (NULL = 0). */
int jvae_bn_fwd_ext_f32(const float* x, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked,
                        float* y, float* save_mean, float* save_invstd,
                        int N, int C, int P, float momentum, float eps, int training, int relu,
                        const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                        void* ws, size_t ws_bytes, void* stream);
int jvae_bn_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd,
                    float* dx, float* dgamma, float* dbeta, int accumulate,
                    int N, int C, int P, int relu, void* ws, size_t ws_bytes, void* stream);

/* Deferred BatchNorm: statistics + per-channel coefficients only (scale, shift: C floats each, y = fmaf(x, scale, shift));
 * the normalisation (+ReLU) itself is applied by the CONSUMING convolution while it stages its input
 * (jvae_conv2d_fwd_aff_f32 / jvae_conv2d_wgrad_aff_f32), so the normalised activation is never written to HBM.  Backward
 * is the ordinary jvae_bn_bwd_f32.  Statistics arguments as jvae_bn_fwd_ext_f32 (ext_nsplit == 0: own statistics pass). */
int jvae_bn_finalize_f32(const float* x, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked,
                         float* save_mean, float* save_invstd, float* scale, float* shift,
                         int N, int C, int P, float momentum, float eps, int training,
                         const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                         void* ws, size_t ws_bytes, void* stream);
/* 1 when both the forward and the weight gradient of this geometry can apply in_scale / in_shift / in_relu to the layer
 * input (the implicit 5x5 kernels); the *_aff entry points return -2 otherwise. */
int jvae_conv2d_affine_ok(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed);
int jvae_conv2d_fwd_aff_f32(const float* x, const float* w, const float* bias, float* y, float* stats, int* nsplit,
                            const float* in_scale, const float* in_shift, int in_relu,
                            int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                            void* ws, size_t ws_bytes, void* stream);
int jvae_conv2d_wgrad_aff_f32(const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                              const float* in_scale, const float* in_shift, int in_relu,
                              int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                              void* ws, size_t ws_bytes, void* stream);

/* Synchronised BatchNorm for data-parallel ranks (no counterpart in the single-process reference, it
 * reproduces what the reference's BatchNorm2d computes on the WHOLE global batch).  Forward: jvae_bn_sums_f32 ->
 * all-reduce(SUM) of the (C,2) sums by the host -> jvae_bn_fwd_sync_f32.  Backward: jvae_bn_bwd_sums_f32 ->
 * all-reduce(SUM) -> jvae_bn_bwd_sync_f32 (dx from the global means, dgamma/dbeta from the local sums).
 * pivot: (C) floats identical on all ranks (the running mean before the update). */
int jvae_bn_sums_f32(const float* x, const float* pivot, float* sums, int N, int C, int P,
                     void* ws, size_t ws_bytes, void* stream);
int jvae_bn_fwd_sync_f32(const float* x, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked,
                         float* y, float* save_mean, float* save_invstd,
                         int N, int C, int P, float momentum, float eps, int relu,
                         const float* global_sums, const float* pivot, int world,
                         void* ws, size_t ws_bytes, void* stream);
int jvae_bn_bwd_sums_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, float* local_sums,
                         int N, int C, int P, int relu, void* ws, size_t ws_bytes, void* stream);
int jvae_bn_bwd_sync_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd,
                         const float* local_sums, const float* global_sums, int world,
                         float* dx, float* dgamma, float* dbeta, int accumulate,
                         int N, int C, int P, int relu, void* stream);

/* ---- bf16 activation path ("B8" layout) --------------------------------------------------------------
 * BASELINE.json configs[4] ("bf16 ... 64x64 deeper conv CVAE"): no counterpart in the fp32 reference.  An activation
 * tensor (N, C, H, W) is stored as (N, ceil(C/8), H, W, 8) bf16: the 8 channels of a pixel are one 16-byte unit, which is
 * one lane's operand of v_mfma_f32_32x32x16_bf16; padding channels are 0.  Master weights (PyTorch layouts, as in the
 * fp32 entry points), biases, BatchNorm statistics / parameters and every gradient of a parameter stay fp32; products
 * are bf16 x bf16 accumulated in fp32.  Geometry arguments as in the *_f32 convolution entry points.
 * jvae_conv2d_native_b8 -> bit mask of the directions that have a bf16 kernel (1 forward, 2 dgrad, 4 wgrad: the 5x5
 * stride-1/2 "same"/"half"/"double" layers of conv32(+) / deconv32(+)); the other entry points return -2 (not
 * supported) for the rest and the host runs those layers on the fp32 kernels between two conversions. */
int jvae_b8_pack_f32(const float* x, void* y, int N, int C, long HW, void* stream);            /* fp32 NCHW -> B8 */
int jvae_b8_unpack_f32(const void* y, float* x, int N, int C, long HW, int accumulate, void* stream);
int jvae_conv2d_native_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed);
size_t jvae_conv2d_workspace_bytes_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP,
                                      int transposed);
int jvae_conv2d_stats_splits_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed);
/* y: B8, or fp32 NCHW when y_f32 (last layer of a stack).  stats / nsplit (both may be NULL): BatchNorm partial sums
 * of (y - bias) taken from the fp32 accumulators, laid out (Cout, *nsplit, 2) as in jvae_conv2d_fwd_stats_f32. */
int jvae_conv2d_fwd_b8(const void* x, const float* w, const float* bias, void* y, int y_f32, float* stats, int* nsplit,
                       int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                       void* ws, size_t ws_bytes, void* stream);
int jvae_conv2d_dgrad_b8(const void* dy, const float* w, void* dx,
                         int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                         void* ws, size_t ws_bytes, void* stream);
/* dw / dbias fp32 (dbias may be NULL); accumulate: add onto their current contents. */
int jvae_conv2d_wgrad_b8(const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                         int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                         void* ws, size_t ws_bytes, void* stream);
/* BatchNorm2d (+ReLU) on B8 tensors; arguments as jvae_bn_fwd_ext_f32 / jvae_bn_bwd_f32 with HW = pixels per image. */
size_t jvae_bn_workspace_bytes_b8(int C);
int jvae_bn_plan_b8(int N, int C, long HW, int* nsplit, int* nchunk);      /* host-only, see jvae_bn_plan */
int jvae_bn_fwd_b8(const void* x, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, long long* num_batches_tracked,
                   void* y, float* save_mean, float* save_invstd,
                   int N, int C, long HW, float momentum, float eps, int training, int relu,
                   const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                   void* ws, size_t ws_bytes, void* stream);
int jvae_bn_bwd_b8(const void* dy, const void* x, const float* gamma, const float* beta,
                   const float* save_mean, const float* save_invstd,
                   void* dx, float* dgamma, float* dbeta, int accumulate,
                   int N, int C, long HW, int relu, void* ws, size_t ws_bytes, void* stream);
/* Deferred form (as jvae_bn_finalize_f32 / *_aff_f32 for fp32): statistics + coef = (2, ceil(C/8)*8) floats (scale, shift;
 * zero on the padding channels); the consuming bf16 convolution applies them (+ReLU) to its input while staging it. */
int jvae_bn_finalize_b8(const void* x, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked,
                        float* save_mean, float* save_invstd, float* coef,
                        int N, int C, long HW, float momentum, float eps, int training,
                        const float* ext_stats, int ext_nsplit, const float* ext_pivot,
                        void* ws, size_t ws_bytes, void* stream);
/* Synchronised BatchNorm on B8 tensors (data-parallel ranks share the batch statistics): the bf16 counterpart of
 * jvae_bn_sums_f32 / jvae_bn_fwd_sync_f32 / jvae_bn_bwd_sums_f32 / jvae_bn_bwd_sync_f32 (same protocol: the host all-reduces
 * the (C,2) sums between the two calls of each direction; reference semantics: nn.BatchNorm2d of the single-process
 * reference on the GLOBAL batch, module/vae_layers/conv.py:214-220).  Workspace: jvae_bn_workspace_bytes_b8(C). */
int jvae_bn_sums_b8(const void* x, const float* pivot, float* sums, int N, int C, long HW,
                    void* ws, size_t ws_bytes, void* stream);
int jvae_bn_fwd_sync_b8(const void* x, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked,
                        void* y, float* save_mean, float* save_invstd,
                        int N, int C, long HW, float momentum, float eps, int relu,
                        const float* global_sums, const float* pivot, int world,
                        void* ws, size_t ws_bytes, void* stream);
int jvae_bn_bwd_sums_b8(const void* dy, const void* x, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, float* local_sums,
                        int N, int C, long HW, int relu, void* ws, size_t ws_bytes, void* stream);
int jvae_bn_bwd_sync_b8(const void* dy, const void* x, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd,
                        const float* local_sums, const float* global_sums, int world,
                        void* dx, float* dgamma, float* dbeta, int accumulate,
                        int N, int C, long HW, int relu, void* ws, size_t ws_bytes, void* stream);
int jvae_conv2d_affine_ok_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed);
int jvae_conv2d_fwd_aff_b8(const void* x, const float* w, const float* bias, void* y, int y_f32, float* stats, int* nsplit,
                           const float* in_scale, const float* in_shift, int in_relu,
                           int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                           void* ws, size_t ws_bytes, void* stream);
int jvae_conv2d_wgrad_aff_b8(const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                             const float* in_scale, const float* in_shift, int in_relu,
                             int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                             void* ws, size_t ws_bytes, void* stream);
int jvae_relu_fwd_b8(const void* x, void* y, long units, void* stream);                /* units of 8 bf16 */
int jvae_relu_bwd_b8(const void* dy, const void* y, void* dx, long units, void* stream);

/* ---- activations (kind 0 identity, 1 ReLU, 2 sigmoid, 3 leaky ReLU with slope 0.01); backward takes the forward OUTPUT ---------- */
int jvae_act_fwd_f32(const float* x, float* y, long n, int kind, void* stream);
int jvae_act_bwd_f32(const float* dy, const float* y, float* dx, long n, int kind, void* stream);

/* ---- latent: clip + reparameterise + KL to the class-conditional prior ----------------------------
 * Replaces Encoder.forward's clip (layers.py:388-394), Sampling.forward (layers.py:230-244; eps is an
 * input: eps (L+1,N,K) with eps[0] = 0), GaussianPrior.kl & helpers (priors.py:173-326),
 * TiltedGaussianPrior.kl (priors.py:389-408), UniformWithGaussianTailPrior.kl (priors.py:429-476) and
 * dzdist (cvae.py:747-753).
 * prior: 0 gaussian, 1 tilted, 2 uniform.  var_dim: 0 scalar T (C,), 1 diag (C,K), 2 full (C,K,K).
 * dict (K+1 floats from jvae_dict_stats_f32, or NULL): dictionary mean and its norm variance.
 * Outputs: lv (N,K) clipped log-variance, z (L+1,N,K), kl / zdist / var_kl / dzdist (N,). */
int jvae_dict_stats_f32(const float* means, float* dict, int C, int K, void* stream);
int jvae_latent_fwd_f32(const float* mu, const float* lv_raw, const float* eps, const long long* y,
                        const float* means, const float* T, const float* dict,
                        float* lv, float* z, float* kl, float* zdist, float* var_kl, float* dzdist,
                        int N, int K, int L, int C, int prior, int var_dim, float tau, float alpha, float w,
                        int sampled, int has_forced, float forced_lv, void* stream);
/* Upstream gradients (any may be NULL): gz (L+1,N,K), g_kl / g_zdist / g_vkl (N,), gmu_direct /
 * glv_direct (N,K).  gmeans (C,K) and gT (shape of T; diag/full only) are ADDED to: per-sample contributions go to
 * the workspace and are folded per class in sample order (deterministic, no float atomics).
 * ws: (4*N + 2*N*K) floats cover every mode. */
int jvae_latent_bwd_f32(const float* mu, const float* lv_raw, const float* lv, const float* eps, const long long* y,
                        const float* means, const float* T,
                        const float* gz, const float* g_kl, const float* g_zdist, const float* g_vkl,
                        const float* gmu_direct, const float* glv_direct,
                        float* gmu, float* glv_raw, float* gmeans, float* gT,
                        int N, int K, int L, int C, int prior, int var_dim, float tau, float alpha, float w,
                        int sampled, int has_forced, void* ws, size_t ws_bytes, void* stream);

/* ---- reconstruction term ---------------------------------------------------------------------------
 * wmse[l][n] = mean_D((x_reco[l+1][n] - x[n])^2) / sigma^2   (mse_loss, module/losses.py:8-27, called at
 * cvae.py:649-652 on x_reco[1:]/sigma, x/sigma).  `sigma_is_log` selects the kind of sigma (cvae.py:626-670):
 *   0  sigma = 1 float on the device (fixed / decayed value)      1  the float is log(sigma) (learned, layers.py:84-89)
 *   2  sigma = N floats, log sigma_n coded by the encoder         3  sigma follows each sample's own rmse: no division here
 *      (cvae.py:631-634)                                             (the ELBO kernels normalise, cvae.py:662-666)
 * Backward fills ALL L+1 rows of g_x_reco (row 0 with zeros); gsigma (may be NULL): 1 float (modes 0/1) or N floats (2).
 * sigma_fwd (mode 0, may be NULL): the value sigma had in the forward pass - the reference's decay rule changes the
 * parameter in place between forward and backward and its backward divides by sigma_fwd * sigma_now (layers.py:168). */
int jvae_recon_fwd_f32(const float* x_reco, const float* x, const float* sigma, int sigma_is_log,
                       float* wmse, int L, int N, int D, void* stream);
int jvae_recon_bwd_f32(const float* x_reco, const float* x, const float* sigma, int sigma_is_log, const float* sigma_fwd,
                       const float* g_wmse, const float* wmse, float* g_x_reco, float* gsigma, int accumulate_sigma,
                       int L, int N, int D, void* ws, size_t ws_bytes, void* stream);

/* mse_loss(x_output, x_target, batch_mean=False) itself (module/losses.py:8-27) for EVERY row of x_output: rows (L,N,D),
 * x (N,D) -> wmse (L,N) = mean_D((rows[l][n] - x[n])^2); backward: g_rows = g_wmse * 2 (rows - x) / D.  (The jvae_recon_*
 * pair above skips row 0 of an (L+1)-row reconstruction; these take the rows as they are.)  16-byte aligned when D % 4 == 0. */
int jvae_mse_rows_fwd_f32(const float* rows, const float* x, float* wmse, int L, int N, int D, void* stream);
int jvae_mse_rows_bwd_f32(const float* rows, const float* x, const float* g_wmse, float* g_rows, int L, int N, int D, void* stream);

/* ---- ELBO assembly (cvae.py:773-791,887-902): wmse = mean_l wmse_s; cross_x = D/2 (2 log sigma + wmse + log 2pi);
 * total = cross_x + cw * ce + beta * kl   (ce may be NULL).  Backward takes the upstream gradients of the three
 * outputs (any may be NULL) and returns g_wmse_s (L,N), g_kl, g_ce (N,) and d/d sigma of the 2 log sigma term.
 * sigma_is_log: the kinds of jvae_recon_fwd_f32; kind 3 divides wmse_s by the sample's mean over l (sigma_n^2) and uses
 * log sigma_n = log(mean)/2; its backward takes the forward's wmse_s in the `sigma` argument.  mse (N, may be NULL)
 * receives wmse * sigma^2 per sample (the `mse` measure). */
int jvae_elbo_fwd_f32(const float* wmse_s, const float* kl, const float* ce, const float* sigma, int sigma_is_log,
                      float* wmse, float* cross_x, float* total, float* mse, int L, int N, int D, float beta, float cw,
                      void* stream);
int jvae_elbo_bwd_f32(const float* g_wmse, const float* g_cx, const float* g_tot, const float* sigma, int sigma_is_log,
                      float* g_wmse_s, float* g_kl, float* g_ce, float* gsigma, int accumulate_sigma,
                      int L, int N, int D, float beta, float cw, void* ws, size_t ws_bytes, void* stream);

/* ---- nn.Dropout(p) of the dense trunks (module/vae_layers/layers.py:287-288, cvae.py:297-298), train mode:
 * y[i] = keep(seed, i) ? x[i] / (1 - p) : 0 with a counter-based mask (the backward pass calls it on dy with the same
 * seed).  The random stream is the kernel's own; the reference draws from torch's global generator. */
int jvae_dropout_f32(const float* x, float* y, long n, float p, long seed, void* stream);
/* the same with the seed in DEVICE memory (mask seed = *seed_dev + salt): nothing about the call depends on host state that
 * changes from step to step, so a captured HIP graph draws a fresh mask at every replay once the caller advances *seed_dev
 * on the stream. */
int jvae_dropout_dev_f32(const float* x, float* y, long n, float p, const long long* seed_dev, long salt, void* stream);

/* ---- importance-weighted bound of the evaluation path (cvae.py:672-676,793-873): li[l][c][n] = log p(x|z_l) +
 * log p(z_l|c) - log q(z_l|x) with log p(x|z_l) = -D/2 (wmse_s + 2 log sigma + log 2pi), -log q = (|eps_l|^2 + sum_k
 * log_var)/2 + K/2 log 2pi; iws[c][n] = mean_l exp(li - max_l li) + max_l li (as the reference writes it).
 * wmse_s (L,N); eps (L,N,K): rows 1..L of the sampling noise; log_var (N,K); log_pz (L,C,N) from the prior's
 * log-density (C = 1 for a non-conditional prior); rows: scratch of L*N floats. */
int jvae_iws_f32(const float* wmse_s, const float* eps, const float* log_var, const float* log_pz, const float* sigma,
                 int sigma_is_log, int L, int N, int K, int C, int D, float* rows, float* iws, void* stream);

/* ---- running measures of evaluate() in ONE device buffer (cvae.py:619-624,689-724,747-762; Encoder.capacity /
 * dict_min_distance layers.py:323-348): out[16]: [0..9] = sigma, mean x^2, mean mse, rmse, mean zdist, mean var_kl, ld-norm,
 * imut-zy, d-mind, optimiser non-finite flag; [10..15] = running means over `batch`+1 calls of xpow, mse, rmse, dB,
 * zdist, var_kl (prev = the previous call's out, NULL at batch 0); wmse has N entries, zdist / var_kl Nz (= N, or C*N
 * for the all-class evaluation).  sumsq_x = sum(x^2) from jvae_sqnorm_accum_f32; means may be NULL.  sigma_is_log >= 2
 * (coded / rmse sigma): `wmse` holds the per-sample MSE itself and sigma[0] the rms value to report. */
int jvae_measures_f32(const float* sumsq_x, long nx, const float* wmse, const float* zdist, const float* var_kl, int N, int Nz,
                      const float* sigma, int sigma_is_log, const float* means, int C, int K, const int* flag,
                      const float* prev, int batch, float* out, void* stream);

/* ---- classification term: per-row cross entropy, target y[r % N] (x_loss, module/losses.py:52-86) -- */
int jvae_xent_fwd_f32(const float* logits, const long long* y, float* ce, int R, int N, int C, void* stream);
int jvae_xent_bwd_f32(const float* logits, const long long* y, const float* g_ce, float* g_logits, int R, int N, int C,
                      void* stream);

/* ---- optimiser: clip_grad_norm_ + Adam (module/optimizers.py:39-47,79-81,120-121; cvae.py:2454-2461)
 * jvae_sqnorm_accum_f32: *acc += sum g^2 (reset != 0 zeroes acc first).
 * jvae_adam_step_f32: g' = g*min(1, max_norm/(sqrt(*sqnorm)+1e-6)) + weight_decay*p, Adam with bias
 * correction for `step` (>= 1); *nonfinite_flag |= 1 when an updated parameter is NaN/Inf. */
size_t jvae_sqnorm_workspace_bytes(void);
int jvae_sqnorm_accum_f32(const float* g, long n, float* acc, int reset, void* ws, size_t ws_bytes, void* stream);
int jvae_clip_scale_f32(float* g, long n, const float* sqnorm, float max_norm, void* stream);
int jvae_adam_step_f32(float* p, const float* g, float* m, float* v, long n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                       float max_norm, const float* sqnorm, int* nonfinite_flag, void* stream);
/* The same update with step count / learning rate / betas in device memory (hyper: 6 floats [lr, beta1, beta2, completed
 * steps, 2 scratch]); advance != 0 increments the step and refreshes the bias corrections first.  No host state: the
 * whole training step can be captured into a HIP graph and replayed (ClassificationVariationalNetwork.graph_train_step). */
int jvae_adam_step_dev_f32(float* p, const float* g, float* m, float* v, long n, float* hyper, int advance,
                           float eps, float weight_decay, float max_norm, const float* sqnorm, int* nonfinite_flag,
                           void* stream);

/* torch.optim.SGD (module/optimizers.py:39-40; momentum / nesterov / weight_decay, dampening 0) with the same fused
 * clip coefficient and NaN/Inf flag as the Adam entry points; buf = momentum buffer (may be NULL when momentum == 0),
 * first_step != 0 initialises it with the gradient as torch does. */
int jvae_sgd_step_f32(float* p, const float* g, float* buf, long n, float lr, float momentum, int nesterov,
                      float weight_decay, int first_step, float max_norm, const float* sqnorm, int* nonfinite_flag,
                      void* stream);

/* ---- MaxPool2d / AvgPool2d / UpsamplingNearest2d: tokens M, A, U of the layer DSL (module/vae_layers/conv.py:201-212;
 * conv-models.ini:13-18,28-30).  Tensors are (planes = N*C, H, W) fp32.  mode 0 = max (idx: int32 (planes, OH, OW), flat
 * h*W + w of the first maximum, as nn.MaxPool2d's return_indices), 1 = average with count_include_pad (divisor K*K).
 * OH = (H + 2P - K)/S + 1.  Up-sampling by an integer factor: y (planes, H*scale, W*scale). */
int jvae_pool2d_out_shape(int H, int W, int K, int S, int P, int* OH, int* OW);
int jvae_pool2d_fwd_f32(const float* x, float* y, int* idx, long planes, int H, int W, int K, int S, int P, int mode,
                        void* stream);
int jvae_pool2d_bwd_f32(const float* dy, const int* idx, float* dx, long planes, int H, int W, int K, int S, int P,
                        int mode, void* stream);
int jvae_upsample_nearest_fwd_f32(const float* x, float* y, long planes, int H, int W, int scale, void* stream);
int jvae_upsample_nearest_bwd_f32(const float* dy, float* dx, long planes, int H, int W, int scale, void* stream);

/* ---- input pipeline in front of the path: uint8 batch (NHWC if nhwc else NCHW) -> horizontal flip
 * where flip[n] != 0 -> edge padding by `pad` + crop at offsets (dy[n], dx[n]) in [0, 2*pad] -> float32 NCHW / 255.
 * Replaces RandomHorizontalFlip + RandomCrop(padding_mode='edge') + ToTensor of utils/torch_load.py:405-426.
 * flip / dy / dx may be NULL (no flip / centred crop).  dy, dx: int32 on the device. */
int jvae_augment_u8_f32(const unsigned char* in, const unsigned char* flip, const int* dy, const int* dx,
                        float* out, int N, int C, int H, int W, int pad, int nhwc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JVAE_HIP_H */
