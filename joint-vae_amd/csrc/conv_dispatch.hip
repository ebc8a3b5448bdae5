// Direction -> primitive mapping (see conv_generic.hip header) and fast-path selection.
//
// Fast paths (conv_mfma.hip) exist for the 5x5 "same-size / half-size" layers that carry >95 % of the FLOPs of
// conv32 / deconv32; everything else (7x7, 8x8, 3x3, 4x4 heads, odd sizes) takes the unfold + GEMM path.
#include "common.h"
#include "conv_dispatch.h"
#include "pack_elems.h"

namespace {

inline bool is5(const ConvGeom& g) { return g.KH == 5 && g.KW == 5; }

// conv forward / transposed dgrad:  big (Cb,Hb,Wb) --conv S,P--> small (Cs,Hs,Ws)
inline bool fold_fwd_fast(const ConvGeom& g) {
    return is5(g) && jvae_conv5_fwd_ok(g.Cb, g.Hb, g.Wb, g.Cs, g.Hs, g.Ws, g.S, g.P);
}
// stride-1 conv dgrad / stride-1 transposed forward: small --conv 1, 4-P, flipped--> big
inline bool fold_bwd_fast_s1(const ConvGeom& g) {
    return is5(g) && g.S == 1 && g.P <= 4 && jvae_conv5_fwd_ok(g.Cs, g.Hs, g.Ws, g.Cb, g.Hb, g.Wb, 1, 4 - g.P);
}

// stride-2 transposed forward / stride-2 conv dgrad: small -> big by the 4-phase kernel
inline bool fold_bwd_fast_s2(const ConvGeom& g) {
    return jvae_convt2_ok(g.Cs, g.Hs, g.Ws, g.Cb, g.Hb, g.Wb, g.KH, g.KW, g.S, g.P);
}
// fp32 operand of the 4-phase kernel: the step's cache slot (pack_cache.hip) or the call's workspace; nullptr: launch error
inline const float* packed_f32(const float* w, float* ws, int C, int O, int swap, int flip, hipStream_t st) {
    bool fresh = true;
    float* slot = (float*)jvae_pack_cache_get(JVAE_PACK_F32, w, C, O, swap, flip, &fresh);
    if (slot && fresh) return slot;
    float* dst = slot ? slot : ws;
    return jvae_conv5_pack(w, dst, C, O, swap, flip, st) == 0 ? dst : nullptr;
}
inline int run_t2(const ConvGeom& g, const float* small, const float* w, const float* bias, float* big, float* ws,
                  hipStream_t st) {
    if (jvae_convt2_x3_ok(g.N, g.Cs, g.Ws, g.Cb)) return jvae_convt2_x3(small, w, bias, big, g.N, g.Cs, g.Ws, g.Cb, ws, st);
    const float* wp = packed_f32(w, ws, g.Cs, g.Cb, 1, 0, st);
    if (!wp) return JVAE_EINVAL;
    return jvae_convt2(small, wp, bias, big, g.N, g.Cs, g.Ws, g.Cb, st);
}

// Transposed convolution of a 1x1 input with no padding (imager.0 of deconv32: 64 x 1 x 1 -> 64 x 8 x 8): the
// output IS the product x[n][ci] . w[ci][(co,kh,kw)] and the dgrad its transpose: plain GEMMs, no fold pass.
inline bool point_input(const ConvGeom& g) {
    return g.Hs == 1 && g.Ws == 1 && g.P == 0 && g.S == 1 && g.Hb == g.KH && g.Wb == g.KW;
}

// wgrad: role swap when the folded side has very few channels (Conv 32->3): see conv_wgrad_mfma.hip
inline bool wgrad_swap(const ConvGeom& g) { return g.S == 1 && g.Cs < 16 && g.Cb >= 16 && g.Hs == g.Hb; }
inline bool wgrad_fast(const ConvGeom& g) {
    if (!is5(g)) return false;
    if (wgrad_swap(g)) return jvae_conv5_wgrad_ok(g.Cb, g.Hb, g.Wb, g.Cs, g.Hs, g.Ws, 1, 4 - g.P);
    return jvae_conv5_wgrad_ok(g.Cs, g.Hs, g.Ws, g.Cb, g.Hb, g.Wb, g.S, g.P);
}
inline size_t wgrad_ws_floats(const ConvGeom& g) {
    if (wgrad_swap(g)) return jvae_conv5_wgrad_ws_floats(g.N, g.Cb, g.Cs, 1, g.Wb);
    return jvae_conv5_wgrad_ws_floats(g.N, g.Cs, g.Cb, g.S, g.Ws);
}

}  // namespace

size_t jvae_conv_ws(const ConvGeom& g, int transposed) {
    (void)transposed;
    size_t a = jvae_conv_generic_ws(g);
    size_t b = is5(g) ? 4 * jvae_conv5_pack_floats(g.Cb, g.Cs) : 0;
    size_t c = wgrad_fast(g) ? 4 * wgrad_ws_floats(g) : 0;
    if (b > a) a = b;
    if (c > a) a = c;
    if (transposed && point_input(g)) {
        const size_t d = 4 * (size_t)16 * g.N * g.Cs;                              // dgrad: K pieces of dx
        const size_t e = 4 * (size_t)8 * g.Cs * g.Cb * g.KH * g.KW;                // wgrad: K pieces of dW
        if (d > a) a = d;
        if (e > a) a = e;
    }
    const size_t e = jvae_channel_sum_ws_bytes(g.Cb > g.Cs ? g.Cb : g.Cs);
    return a > e ? a : e;
}

// Can the forward of this layer emit per-workgroup BatchNorm partial sums, and how many per channel at most?
int jvae_conv_stats_splits(const ConvGeom& g, int transposed) {
    if (!transposed) {
        if (jvae_conv5_smallco_ok(g.Cb, g.Hb, g.Wb, g.Cs, g.KH, g.KW, g.S, g.P)) return 0;
        return fold_fwd_fast(g) ? jvae_conv5_fwd_max_splits(g.N, g.Ws) : 0;
    }
    if (point_input(g)) return 0;
    if (fold_bwd_fast_s1(g)) return jvae_conv5_fwd_max_splits(g.N, g.Wb);
    if (fold_bwd_fast_s2(g)) return jvae_conv5_fwd_max_splits(g.N, g.Ws);
    return 0;
}

// Forward AND weight gradient of this layer can apply a deferred BatchNorm to the layer input while staging it.
bool jvae_conv_affine_ok(const ConvGeom& g, int transposed) {
    if (!wgrad_fast(g)) return false;
    if (!transposed) return jvae_conv5_smallco_ok(g.Cb, g.Hb, g.Wb, g.Cs, g.KH, g.KW, g.S, g.P) || fold_fwd_fast(g);
    if (point_input(g)) return false;
    return fold_bwd_fast_s1(g) || fold_bwd_fast_s2(g);
}

int jvae_conv_fwd(const ConvGeom& g, int transposed, const float* x, const float* w, const float* bias, float* y,
                  float* ws, size_t ws_bytes, hipStream_t st, float* stats, int* nsplit, const InAff* aff) {
    if (nsplit) *nsplit = 0;
    if (!transposed) {
        if (jvae_conv5_smallco_ok(g.Cb, g.Hb, g.Wb, g.Cs, g.KH, g.KW, g.S, g.P))
            return jvae_conv5_smallco(x, w, bias, y, g.N, g.Cb, g.Wb, g.Cs, st, aff);
        if (fold_fwd_fast(g) && ws_bytes >= 4 * jvae_conv5_pack_floats(g.Cb, g.Cs))
            return jvae_conv5_fwd(x, w, 0, 0, bias, y, g.N, g.Cb, g.Hb, g.Wb, g.Cs, g.Ws, g.S, g.P, ws, st, stats, nsplit, aff);
        if (aff) return JVAE_ENOTSUP;
        return jvae_fold_fwd(g, x, w, bias, y, ws, ws_bytes, st);
    }
    if (aff && point_input(g)) return JVAE_ENOTSUP;
    if (point_input(g)) {
        const int cols = g.Cb * g.KH * g.KW;
        return jvae_gemm_launch_ex(g.N, cols, g.Cs, 1, x, g.Cs, 1, 0, w, cols, 1, 0, y, cols, 1, 0,
                                   bias, bias ? 1 : 0, g.KH * g.KW, 0, 1, st);
    }
    if (fold_bwd_fast_s1(g) && ws_bytes >= 4 * jvae_conv5_pack_floats(g.Cs, g.Cb))
        return jvae_conv5_fwd(x, w, 1, 1, bias, y, g.N, g.Cs, g.Hs, g.Ws, g.Cb, g.Wb, 1, 4 - g.P, ws, st, stats, nsplit, aff);
    if (fold_bwd_fast_s2(g) && ws_bytes >= 4 * jvae_conv5_pack_floats(g.Cs, g.Cb)) {
        if (jvae_convt2_x3_ok(g.N, g.Cs, g.Ws, g.Cb))
            return jvae_convt2_x3(x, w, bias, y, g.N, g.Cs, g.Ws, g.Cb, ws, st, stats, nsplit, aff);
        const float* wp = packed_f32(w, ws, g.Cs, g.Cb, 1, 0, st);
        if (!wp) return JVAE_EINVAL;
        return jvae_convt2(x, wp, bias, y, g.N, g.Cs, g.Ws, g.Cb, st, stats, nsplit, aff);
    }
    if (aff) return JVAE_ENOTSUP;
    return jvae_fold_bwd(g, x, w, bias, y, ws, ws_bytes, st);
}

int jvae_conv_dgrad(const ConvGeom& g, int transposed, const float* dy, const float* w, float* dx,
                    float* ws, size_t ws_bytes, hipStream_t st) {
    if (!transposed) {
        if (fold_bwd_fast_s1(g) && ws_bytes >= 4 * jvae_conv5_pack_floats(g.Cs, g.Cb))
            return jvae_conv5_fwd(dy, w, 1, 1, nullptr, dx, g.N, g.Cs, g.Hs, g.Ws, g.Cb, g.Wb, 1, 4 - g.P, ws, st);
        if (fold_bwd_fast_s2(g) && ws_bytes >= 4 * jvae_conv5_pack_floats(g.Cs, g.Cb))
            return run_t2(g, dy, w, nullptr, dx, ws, st);
        return jvae_fold_bwd(g, dy, w, nullptr, dx, ws, ws_bytes, st);
    }
    if (point_input(g)) {
        const int cols = g.Cb * g.KH * g.KW;        // dx[n][ci] = sum_j dy[n][j] w[ci][j]: few tiles, long K -> K pieces,
        const long outf = (long)g.N * g.Cs;         // stored side by side and folded in a fixed order (deterministic)
        if (ws_bytes < 4 * (size_t)(16 * outf)) return JVAE_EWORKSPACE;
        int S = 0;
        int rc = jvae_gemm_launch_part(g.N, g.Cs, cols, 1, dy, cols, 1, 0, w, 1, cols, 0, ws, g.Cs, 1, 0, outf, 16, &S, st);
        if (rc) return rc;
        return jvae_splitk_fold(ws, nullptr, dx, S, outf, g.Cs, 0, 0, st);
    }
    if (fold_fwd_fast(g) && ws_bytes >= 4 * jvae_conv5_pack_floats(g.Cb, g.Cs))
        return jvae_conv5_fwd(dy, w, 0, 0, nullptr, dx, g.N, g.Cb, g.Hb, g.Wb, g.Cs, g.Ws, g.S, g.P, ws, st);
    return jvae_fold_fwd(g, dy, w, nullptr, dx, ws, ws_bytes, st);
}

int jvae_conv_wgrad(const ConvGeom& g, int transposed, const float* x, const float* dy, float* dw,
                    float* ws, size_t ws_bytes, hipStream_t st, const InAff* aff) {
    const float* big = transposed ? dy : x;       // unfolded side
    const float* small = transposed ? x : dy;     // folded side
    // the deferred BatchNorm belongs to the layer INPUT x: the big side of a convolution, the small side of a transposed one
    const InAff* aff_big = transposed ? nullptr : aff;
    const InAff* aff_small = transposed ? aff : nullptr;
    if (wgrad_fast(g) && ws_bytes >= 4 * wgrad_ws_floats(g)) {
        // the generic entry point zeroed dw (or holds the value to accumulate onto): always accumulate here
        if (wgrad_swap(g))      // roles swapped: ps = big, q = small
            return jvae_conv5_wgrad(big, small, dw, 1, 1, g.N, g.Cb, g.Wb, g.Cs, 1, 4 - g.P, ws, st, aff_big, aff_small);
        return jvae_conv5_wgrad(small, big, dw, 1, 0, g.N, g.Cs, g.Ws, g.Cb, g.S, g.P, ws, st, aff_small, aff_big);
    }
    if (aff) return JVAE_ENOTSUP;
    if (transposed && point_input(g)) {
        // ConvTranspose2d of a 1x1 input (imager.0): unfolding the kxk output at its single position is the identity, so
        // dW[ci][j] (+)= sum_n x[n][ci] dy[n][j] is a plain product of the two tensors as they lie in memory (the generic path
        // copied dy into a col buffer first and ran 64 workgroups over K = N: 60 us of a 4 ms step).  Few tiles, long K:
        // K pieces stored side by side, folded onto dw in a fixed order (deterministic).
        const int cols = g.Cb * g.KH * g.KW;
        const long outf = (long)g.Cs * cols;
        int want = (int)(1024 / ((long)cdiv(g.Cs, 64) * cdiv(cols, 64)));
        want = want < 1 ? 1 : (want > 8 ? 8 : want);
        if (ws_bytes < 4 * (size_t)(want * outf)) return JVAE_EWORKSPACE;
        int S = 0;
        int rc = jvae_gemm_launch_part(g.Cs, cols, g.N, 1, small, 1, g.Cs, 0, big, cols, 1, 0, ws, cols, 1, 0, outf, want, &S, st);
        if (rc) return rc;
        return jvae_splitk_fold(ws, nullptr, dw, S, outf, cols, 0, 1, st);
    }
    return jvae_fold_wgrad(g, big, small, dw, ws, ws_bytes, st);
}
