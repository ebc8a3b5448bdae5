// C-ABI entry points for (transposed) convolution: argument checking, geometry, kernel selection.
// See include/jvae_hip.h for the contract of each function.
#include "common.h"
#include "jvae_internal.h"
#include "conv_dispatch.h"

namespace {

// (x: the layer's input, y: the layer's output) -> big/small-side geometry.
bool make_geom(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
               ConvGeom* g, int* OH, int* OW) {
    if (N < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || S <= 0 || P < 0 || OP < 0)
        return false;
    g->N = N; g->KH = KH; g->KW = KW; g->S = S; g->P = P;
    if (!transposed) {
        if (OP != 0) return false;
        *OH = (H + 2 * P - KH) / S + 1;
        *OW = (W + 2 * P - KW) / S + 1;
        g->Cb = Cin; g->Hb = H; g->Wb = W;
        g->Cs = Cout; g->Hs = *OH; g->Ws = *OW;
    } else {
        if (OP >= S && OP != 0) return false;
        *OH = (H - 1) * S - 2 * P + KH + OP;
        *OW = (W - 1) * S - 2 * P + KW + OP;
        g->Cs = Cin; g->Hs = H; g->Ws = W;
        g->Cb = Cout; g->Hb = *OH; g->Wb = *OW;
    }
    return *OH > 0 && *OW > 0;
}

}  // namespace

extern "C" {

int jvae_conv2d_set_split_bf16(int mode) { return jvae_conv5_x3_set(mode); }
int jvae_conv2d_set_split_shape16(int on) { return jvae_conv5_x3_set_shape16(on); }

size_t jvae_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP,
                                   int transposed) {
    ConvGeom g; int oh, ow;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return 0;
    return jvae_conv_ws(g, transposed);
}

int jvae_conv2d_out_shape(int H, int W, int KH, int KW, int S, int P, int OP, int transposed, int* OH, int* OW) {
    ConvGeom g;
    if (!OH || !OW) return JVAE_EINVAL;
    return make_geom(1, 1, H, W, 1, KH, KW, S, P, OP, transposed, &g, OH, OW) ? 0 : JVAE_EINVAL;
}

int jvae_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                        int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                        void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!x || !w || !y) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (N == 0) return 0;
    return jvae_conv_fwd(g, transposed, x, w, bias, y, (float*)ws, ws_bytes, (hipStream_t)stream);
}

// Forward that also emits BatchNorm partial statistics of (y - bias) when the selected kernel can produce them.
// stats: (Cout, stats_cap, 2) floats with stats_cap >= jvae_conv2d_stats_splits(...); *nsplit (HOST int) receives the
// number of partials per channel actually written, laid out as (Cout, *nsplit, 2); 0 = not produced (run the BN
// statistics kernel instead).
int jvae_conv2d_stats_splits(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed) {
    ConvGeom g; int oh, ow;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return 0;
    return jvae_conv_stats_splits(g, transposed);
}

int jvae_conv2d_fwd_stats_f32(const float* x, const float* w, const float* bias, float* y, float* stats, int* nsplit,
                              int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                              void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!x || !w || !y || !nsplit) return JVAE_EINVAL;
    *nsplit = 0;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (N == 0) return 0;
    return jvae_conv_fwd(g, transposed, x, w, bias, y, (float*)ws, ws_bytes, (hipStream_t)stream, stats, nsplit);
}

// ---- deferred BatchNorm on the layer input (DESIGN.md "Streams, fusion"): a = [relu](x*in_scale[c] + in_shift[c]) is
// applied while the kernel stages x, for the layers whose forward AND weight gradient run on the implicit kernels.
int jvae_conv2d_affine_ok(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed) {
    ConvGeom g; int oh, ow;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return 0;
    return jvae_conv_affine_ok(g, transposed) ? 1 : 0;
}

int jvae_conv2d_fwd_aff_f32(const float* x, const float* w, const float* bias, float* y, float* stats, int* nsplit,
                            const float* in_scale, const float* in_shift, int in_relu,
                            int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                            void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!x || !w || !y || !in_scale || !in_shift) return JVAE_EINVAL;
    if (stats && !nsplit) return JVAE_EINVAL;
    if (nsplit) *nsplit = 0;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (!jvae_conv_affine_ok(g, transposed)) return JVAE_ENOTSUP;
    if (N == 0) return 0;
    const InAff aff{in_scale, in_shift, in_relu};
    return jvae_conv_fwd(g, transposed, x, w, bias, y, (float*)ws, ws_bytes, (hipStream_t)stream, stats, nsplit, &aff);
}

int jvae_conv2d_wgrad_aff_f32(const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                              const float* in_scale, const float* in_shift, int in_relu,
                              int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                              void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!x || !dy || !dw || !in_scale || !in_shift) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (!jvae_conv_affine_ok(g, transposed)) return JVAE_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cin * Cout * KH * KW, st);
        if (e != hipSuccess) return (int)e;
    }
    if (N == 0) {
        if (dbias && !accumulate) {
            hipError_t e = hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, st);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    }
    const InAff aff{in_scale, in_shift, in_relu};
    int rc = jvae_conv_wgrad(g, transposed, x, dy, dw, (float*)ws, ws_bytes, st, &aff);
    if (rc) return rc;
    if (dbias) rc = jvae_channel_sum(dy, dbias, N, Cout, oh * ow, accumulate, (float*)ws, ws_bytes, st);   // ws is free again
    return rc;
}

int jvae_conv2d_dgrad_f32(const float* dy, const float* w, float* dx,
                          int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                          void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!dy || !w || !dx) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (N == 0) return 0;
    return jvae_conv_dgrad(g, transposed, dy, w, dx, (float*)ws, ws_bytes, (hipStream_t)stream);
}

int jvae_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                          int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                          void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!x || !dy || !dw) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cin * Cout * KH * KW, st);
        if (e != hipSuccess) return (int)e;
    }
    if (N == 0) {
        if (dbias && !accumulate) {
            hipError_t e = hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, st);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    }
    int rc = jvae_conv_wgrad(g, transposed, x, dy, dw, (float*)ws, ws_bytes, st);
    if (rc) return rc;
    if (dbias) rc = jvae_channel_sum(dy, dbias, N, Cout, oh * ow, accumulate, (float*)ws, ws_bytes, st);   // ws is free again
    return rc;
}

// out[c] (+)= sum over n, q of t[n][c][q]   (bias gradients of conv / linear layers)
size_t jvae_channel_sum_workspace_bytes(int C) { return jvae_channel_sum_ws_bytes(C); }

int jvae_channel_sum_f32(const float* t, float* out, int N, int C, int P, int accumulate, void* ws, size_t ws_bytes,
                         void* stream) {
    if (!t || !out || N < 0 || C <= 0 || P <= 0) return JVAE_EINVAL;
    return jvae_channel_sum(t, out, N, C, P, accumulate, (float*)ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
