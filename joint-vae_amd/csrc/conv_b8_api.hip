// C-ABI entry points of the bf16 "B8" convolution path: geometry, kernel selection.  A direction without a native
// bf16 kernel returns JVAE_ENOTSUP (jvae_conv2d_native_b8 tells in advance); the host then runs that layer through the
// fp32 kernels between two layout conversions.
#include "common.h"
#include "jvae_internal.h"
#include "conv_b8.h"

namespace {

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

inline bool is5(const ConvGeom& g) { return g.KH == 5 && g.KW == 5; }
// big --conv S,P--> small
inline bool fold_fwd_fast(const ConvGeom& g) {
    return is5(g) && jvae_conv5_b8_fwd_ok(g.Cb, g.Hb, g.Wb, g.Cs, g.Hs, g.Ws, g.S, g.P);
}
// small --conv 1, 4-P, flipped--> big
inline bool fold_bwd_fast_s1(const ConvGeom& g) {
    return is5(g) && g.S == 1 && g.P <= 4 && jvae_conv5_b8_fwd_ok(g.Cs, g.Hs, g.Ws, g.Cb, g.Hb, g.Wb, 1, 4 - g.P);
}

// small -> big by the 4-phase kernel (stride-2 transposed forward / stride-2 conv dgrad)
inline bool fold_bwd_fast_s2(const ConvGeom& g) {
    return jvae_convt2_b8_ok(g.Cs, g.Hs, g.Ws, g.Cb, g.Hb, g.Wb, g.KH, g.KW, g.S, g.P);
}

// wgrad: role swap when the folded side has <= 8 channels (Conv 32 -> 3): the 8-channel side becomes `b` (MODE 1)
inline bool wgrad_swap(const ConvGeom& g) { return g.S == 1 && g.Cs <= 8 && g.Cb > 8 && g.Hs == g.Hb; }
inline bool wgrad_fast(const ConvGeom& g) {
    if (!is5(g)) return false;
    if (wgrad_swap(g)) return jvae_conv5_wgrad_b8_ok(g.Cb, g.Hb, g.Wb, g.Cs, g.Hs, g.Ws, 1, 4 - g.P);
    return jvae_conv5_wgrad_b8_ok(g.Cs, g.Hs, g.Ws, g.Cb, g.Hb, g.Wb, g.S, g.P);
}
inline size_t wgrad_ws_bytes(const ConvGeom& g) {
    size_t slab = 4 * (wgrad_swap(g) ? jvae_conv5_wgrad_b8_ws_floats(g.N, g.Cb, g.Cs) : jvae_conv5_wgrad_b8_ws_floats(g.N, g.Cs, g.Cb));
    return slab;
}
inline size_t chsum_ws_bytes(int C) { return (size_t)((C + 7) / 8) * 8 * 64 * 4; }

enum { DIR_FWD = 1, DIR_DGRAD = 2, DIR_WGRAD = 4 };

int native_mask(const ConvGeom& g, int transposed) {
    int m = 0;
    if (!transposed) {
        if (fold_fwd_fast(g)) m |= DIR_FWD;
        if (fold_bwd_fast_s1(g) || fold_bwd_fast_s2(g)) m |= DIR_DGRAD;
    } else {
        if (fold_bwd_fast_s1(g) || fold_bwd_fast_s2(g)) m |= DIR_FWD;
        if (fold_fwd_fast(g)) m |= DIR_DGRAD;
    }
    if (wgrad_fast(g)) m |= DIR_WGRAD;
    return m;
}

}  // namespace

extern "C" {

int jvae_b8_pack_f32(const float* x, void* y, int N, int C, long HW, void* stream) {
    if (!x || !y || N < 0 || C <= 0 || HW <= 0) return JVAE_EINVAL;
    return jvae_b8_pack(x, y, N, C, HW, (hipStream_t)stream);
}

int jvae_b8_unpack_f32(const void* y, float* x, int N, int C, long HW, int accumulate, void* stream) {
    if (!x || !y || N < 0 || C <= 0 || HW <= 0) return JVAE_EINVAL;
    return jvae_b8_unpack(y, x, N, C, HW, accumulate, (hipStream_t)stream);
}

int jvae_conv2d_native_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed) {
    ConvGeom g; int oh, ow;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return 0;
    return native_mask(g, transposed);
}

size_t jvae_conv2d_workspace_bytes_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP,
                                      int transposed) {
    ConvGeom g; int oh, ow;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return 0;
    if (!is5(g)) return 0;
    size_t a = jvae_conv5_b8_pack_bytes(g.Cb, g.Cs), b = jvae_conv5_b8_pack_bytes(g.Cs, g.Cb);
    if (b > a) a = b;
    if (wgrad_fast(g)) {
        b = wgrad_ws_bytes(g) + chsum_ws_bytes(Cout);
        if (b > a) a = b;
    }
    return a;
}

int jvae_conv2d_stats_splits_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed) {
    ConvGeom g; int oh, ow;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return 0;
    if (!(native_mask(g, transposed) & DIR_FWD)) return 0;
    if (transposed && fold_bwd_fast_s2(g)) return jvae_conv5_b8_max_splits(N, g.Ws);
    return jvae_conv5_b8_max_splits(N, ow);
}

// x: B8; y: B8, or fp32 NCHW when y_f32.  stats (Cout, cap, 2) / nsplit as jvae_conv2d_fwd_stats_f32 (both may be NULL).
int jvae_conv2d_fwd_b8(const void* x, const float* w, const float* bias, void* y, int y_f32, float* stats, int* nsplit,
                       int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                       void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (nsplit) *nsplit = 0;
    if (!x || !w || !y) return JVAE_EINVAL;
    if (stats && !nsplit) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!transposed) {
        if (!fold_fwd_fast(g)) return JVAE_ENOTSUP;
        if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cb, g.Cs) || !ws) return JVAE_EWORKSPACE;
        return jvae_conv5_b8_fwd(x, w, 0, 0, bias, y, y_f32, g.N, g.Cb, g.Hb, g.Wb, g.Cs, g.Ws, g.S, g.P, ws, st, stats, nsplit);
    }
    if (fold_bwd_fast_s1(g)) {
        if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cs, g.Cb) || !ws) return JVAE_EWORKSPACE;
        return jvae_conv5_b8_fwd(x, w, 1, 1, bias, y, y_f32, g.N, g.Cs, g.Hs, g.Ws, g.Cb, g.Wb, 1, 4 - g.P, ws, st, stats, nsplit);
    }
    if (fold_bwd_fast_s2(g) && !y_f32) {
        if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cs, g.Cb) || !ws) return JVAE_EWORKSPACE;
        return jvae_convt2_b8(x, w, bias, y, g.N, g.Cs, g.Ws, g.Cb, ws, st, stats, nsplit);
    }
    return JVAE_ENOTSUP;
}

// ---- deferred BatchNorm(+ReLU) on the B8 layer input: in_scale / in_shift hold ceil(Cin/8)*8 floats (jvae_bn_finalize_b8)
int jvae_conv2d_affine_ok_b8(int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed) {
    ConvGeom g; int oh, ow;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return 0;
    if ((Cin + 7) / 8 * 8 > 256) return 0;
    const int m = native_mask(g, transposed);
    return (m & DIR_FWD) && (m & DIR_WGRAD) ? 1 : 0;
}

int jvae_conv2d_fwd_aff_b8(const void* x, const float* w, const float* bias, void* y, int y_f32, float* stats, int* nsplit,
                           const float* in_scale, const float* in_shift, int in_relu,
                           int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                           void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (nsplit) *nsplit = 0;
    if (!x || !w || !y || !in_scale || !in_shift) return JVAE_EINVAL;
    if (stats && !nsplit) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const InAff aff{in_scale, in_shift, in_relu};
    if (!transposed) {
        if (!fold_fwd_fast(g)) return JVAE_ENOTSUP;
        if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cb, g.Cs) || !ws) return JVAE_EWORKSPACE;
        return jvae_conv5_b8_fwd(x, w, 0, 0, bias, y, y_f32, g.N, g.Cb, g.Hb, g.Wb, g.Cs, g.Ws, g.S, g.P, ws, st, stats, nsplit, &aff);
    }
    if (fold_bwd_fast_s1(g)) {
        if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cs, g.Cb) || !ws) return JVAE_EWORKSPACE;
        return jvae_conv5_b8_fwd(x, w, 1, 1, bias, y, y_f32, g.N, g.Cs, g.Hs, g.Ws, g.Cb, g.Wb, 1, 4 - g.P, ws, st, stats, nsplit, &aff);
    }
    if (fold_bwd_fast_s2(g) && !y_f32) {
        if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cs, g.Cb) || !ws) return JVAE_EWORKSPACE;
        return jvae_convt2_b8(x, w, bias, y, g.N, g.Cs, g.Ws, g.Cb, ws, st, stats, nsplit, &aff);
    }
    return JVAE_ENOTSUP;
}

// dy: B8 -> dx: B8
int jvae_conv2d_dgrad_b8(const void* dy, const float* w, void* dx,
                         int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                         void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!dy || !w || !dx) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!transposed) {
        if (fold_bwd_fast_s2(g)) {
            if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cs, g.Cb) || !ws) return JVAE_EWORKSPACE;
            return jvae_convt2_b8(dy, w, nullptr, dx, g.N, g.Cs, g.Ws, g.Cb, ws, st);
        }
        if (!fold_bwd_fast_s1(g)) return JVAE_ENOTSUP;
        if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cs, g.Cb) || !ws) return JVAE_EWORKSPACE;
        return jvae_conv5_b8_fwd(dy, w, 1, 1, nullptr, dx, 0, g.N, g.Cs, g.Hs, g.Ws, g.Cb, g.Wb, 1, 4 - g.P, ws, st);
    }
    if (!fold_fwd_fast(g)) return JVAE_ENOTSUP;
    if (ws_bytes < jvae_conv5_b8_pack_bytes(g.Cb, g.Cs) || !ws) return JVAE_EWORKSPACE;
    return jvae_conv5_b8_fwd(dy, w, 0, 0, nullptr, dx, 0, g.N, g.Cb, g.Hb, g.Wb, g.Cs, g.Ws, g.S, g.P, ws, st);
}

// x, dy: B8 (layer input / gradient of the layer output); dw fp32 in the layer's own layout; dbias may be NULL
static int wgrad_b8_impl(const void* x, const void* dy, float* dw, float* dbias, int accumulate, const InAff* aff,
                         int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                         void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g; int oh, ow;
    if (!x || !dy || !dw) return JVAE_EINVAL;
    if (!make_geom(N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, &g, &oh, &ow)) return JVAE_EINVAL;
    if (!wgrad_fast(g)) return JVAE_ENOTSUP;
    if (!ws || ws_bytes < wgrad_ws_bytes(g) + (dbias ? chsum_ws_bytes(Cout) : 0)) return JVAE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cin * Cout * 25, st);
        if (e != hipSuccess) return (int)e;
        if (dbias && (e = hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, st)) != hipSuccess) return (int)e;
    }
    if (N == 0) return 0;
    const void* big = transposed ? dy : x;
    const void* small = transposed ? x : dy;
    const InAff* aff_big = transposed ? nullptr : aff;       // the deferred BatchNorm belongs to the layer input x
    const InAff* aff_small = transposed ? aff : nullptr;
    int rc;
    if (wgrad_swap(g))
        rc = jvae_conv5_wgrad_b8(big, small, dw, 1, 1, g.N, g.Cb, g.Wb, g.Cs, 1, 4 - g.P, (float*)ws, st, aff_big, aff_small);
    else
        rc = jvae_conv5_wgrad_b8(small, big, dw, 1, 0, g.N, g.Cs, g.Ws, g.Cb, g.S, g.P, (float*)ws, st, aff_small, aff_big);
    if (rc) return rc;
    if (dbias) rc = jvae_b8_channel_sum(dy, dbias, N, Cout, (long)oh * ow, 1, (float*)((char*)ws + wgrad_ws_bytes(g)), st);
    return rc;
}

int jvae_conv2d_wgrad_b8(const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                         int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                         void* ws, size_t ws_bytes, void* stream) {
    return wgrad_b8_impl(x, dy, dw, dbias, accumulate, nullptr, N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, ws, ws_bytes,
                         stream);
}

int jvae_conv2d_wgrad_aff_b8(const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                             const float* in_scale, const float* in_shift, int in_relu,
                             int N, int Cin, int H, int W, int Cout, int KH, int KW, int S, int P, int OP, int transposed,
                             void* ws, size_t ws_bytes, void* stream) {
    if (!in_scale || !in_shift) return JVAE_EINVAL;
    const InAff aff{in_scale, in_shift, in_relu};
    return wgrad_b8_impl(x, dy, dw, dbias, accumulate, &aff, N, Cin, H, W, Cout, KH, KW, S, P, OP, transposed, ws, ws_bytes,
                         stream);
}

}  // extern "C"
