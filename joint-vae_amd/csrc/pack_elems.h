// The three weight re-pack mappings of the 5x5 convolution kernels, one definition each: used by the per-layer pack kernels
// (conv_mfma.hip, conv_x3.hip, conv_b8.hip) and by the batched refresh of the pack cache (pack_cache.hip).
// Source w: PyTorch layout [o][c][tap] (swap: [c][o][tap] = ConvTranspose2d / role swap); flip: tap -> 24 - tap.
#pragma once
#include "common.h"
#include "conv_x3.h"

enum JvaePackKind { JVAE_PACK_F32 = 0, JVAE_PACK_X3 = 1, JVAE_PACK_B8 = 2, JVAE_PACK_X3S = 3, JVAE_PACK_SCI = 4, JVAE_PACK_T2S = 5 };

// tap pairs per K step of the 16x16x32 split-bf16 layout: 25 taps = 12 pairs + tap 24 with an all-zero partner, + one all-zero
// pair so that every staging group of the kernel (2 pairs) is complete
#define JVAE_X3S_PAIRS 14

__host__ __device__ __forceinline__ int jvae_pack_op(int O) { return (O + 31) / 32 * 32; }

// number of pack elements (= loop trips of the element functions below) and bytes of the packed form
__host__ __device__ __forceinline__ long jvae_pack_elems(int kind, int C, int O) {
    const int OP = jvae_pack_op(O);
    if (kind == JVAE_PACK_F32) return (long)C * 25 * OP;
    if (kind == JVAE_PACK_SCI) return (long)((O + 7) / 8) * C * 25 * 8;
    if (kind == JVAE_PACK_X3S) return (long)((C + 15) / 16) * JVAE_X3S_PAIRS * 4 * OP * 8;
    if (kind == JVAE_PACK_T2S) return (long)((C + 31) / 32) * 25 * 4 * OP * 8;
    return (long)((C + 15) / 16) * 25 * 2 * OP * 8;
}
__host__ __device__ __forceinline__ size_t jvae_pack_bytes(int kind, int C, int O) {
    const long n = jvae_pack_elems(kind, C, O);
    return (size_t)n * ((kind == JVAE_PACK_F32 || kind == JVAE_PACK_SCI) ? 4 : ((kind == JVAE_PACK_X3 || kind == JVAE_PACK_X3S || kind == JVAE_PACK_T2S) ? 6 : 2));
}

__device__ __forceinline__ float jvae_pack_src(const float* __restrict__ w, int C, int O, int c, int o, int tap, int swap, int flip) {
    const int st = flip ? 24 - tap : tap;
    if (c >= C || o >= O) return 0.f;
    return swap ? w[((long)c * O + o) * 25 + st] : w[((long)o * C + c) * 25 + st];
}

// fp32 operand of conv_mfma.hip / conv_t2_mfma.hip: Wp[c][tap][o] (o < OP, zero for o >= O)
__device__ __forceinline__ void jvae_pack_f32_elem(const float* __restrict__ w, float* __restrict__ wp, long i,
                                                   int C, int O, int swap, int flip) {
    const int OP = jvae_pack_op(O);
    const int o = (int)(i % OP), tap = (int)((i / OP) % 25), c = (int)(i / ((long)OP * 25));
    wp[i] = jvae_pack_src(w, C, O, c, o, tap, swap, flip);
}

// fp32 operand of conv_smallco.hip's few-input-channel kernel: Wp[o / 8][c][tap][o % 8] (the 40 weights of one (channel, kernel
// row) and channel group are 160 contiguous bytes: five s_load_dwordx8), zero for o >= O
__device__ __forceinline__ void jvae_pack_sci_elem(const float* __restrict__ w, float* __restrict__ wp, long i,
                                                   int C, int O, int swap, int flip) {
    const int o8 = (int)(i % 8);
    long t = i / 8;
    const int tap = (int)(t % 25); t /= 25;
    const int c = (int)(t % C), g = (int)(t / C);
    wp[i] = jvae_pack_src(w, C, O, c, g * 8 + o8, tap, swap, flip);
}

// i -> (kb, tap, half, o, ci) of the 16-byte-unit layouts (8 channels of one (tap, o) per unit)
__device__ __forceinline__ void jvae_pack_unit_index(long i, int OP, int* kb, int* tap, int* half, int* o, int* ci) {
    *ci = (int)(i % 8);
    long t = i / 8;
    *o = (int)(t % OP); t /= OP;
    *half = (int)(t % 2); t /= 2;
    *tap = (int)(t % 25);
    *kb = (int)(t / 25);
}

// split-bf16 operand of conv_x3.hip / conv_t2_x3.hip:
// Wp[(kb*5 + kh)][(plane*5 + kw)*2 + half][o][ci] = plane(W[o][c = kb*16 + half*8 + ci][tap = kh*5 + kw])
__device__ __forceinline__ void jvae_pack_x3_elem(const float* __restrict__ w, __bf16* __restrict__ wp, long i,
                                                  int C, int O, int swap, int flip) {
    const int OP = jvae_pack_op(O);
    int kb, tap, half, o, ci;
    jvae_pack_unit_index(i, OP, &kb, &tap, &half, &o, &ci);
    const int kh = tap / 5, kw = tap % 5;
    const float v = jvae_pack_src(w, C, O, kb * 16 + half * 8 + ci, o, tap, swap, flip);
    __bf16 s[3];
    x3_split(v, s[0], s[1], s[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        wp[((((long)(kb * 5 + kh) * 30 + (pl * 5 + kw) * 2 + half) * OP) + o) * 8 + ci] = s[pl];
}

// split-bf16 operand of conv_x3.hip's 16x16x32 form (K index of one MFMA = 2 taps x 16 channels):
// Wp[kb][pair][plane][kq][o][ci] = plane(W[o][c = kb*16 + (kq&1)*8 + ci][tap = 2*pair + (kq>>1)]), zero for tap >= 25
__device__ __forceinline__ void jvae_pack_x3s_elem(const float* __restrict__ w, __bf16* __restrict__ wp, long i,
                                                   int C, int O, int swap, int flip) {
    const int OP = jvae_pack_op(O);
    const int ci = (int)(i % 8);
    long t = i / 8;
    const int o = (int)(t % OP); t /= OP;
    const int kq = (int)(t % 4); t /= 4;
    const int pair = (int)(t % JVAE_X3S_PAIRS);
    const int kb = (int)(t / JVAE_X3S_PAIRS);
    const int tap = 2 * pair + (kq >> 1);
    const float v = tap < 25 ? jvae_pack_src(w, C, O, kb * 16 + (kq & 1) * 8 + ci, o, tap, swap, flip) : 0.f;
    __bf16 s[3];
    x3_split(v, s[0], s[1], s[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        wp[(((((long)kb * JVAE_X3S_PAIRS + pair) * 3 + pl) * 4 + kq) * OP + o) * 8 + ci] = s[pl];
}

// Tap sequence of the 4-phase stride-2 transposed kernel's 16x16x32 form (conv_t2_x3.hip, round 5): the 25 taps sorted by the
// patch POSITION (dh, dw) = ((kh & 1) + 2 - kh) / 2, ((kw & 1) + 2 - kw) / 2 they read - the 25 taps touch only the 3 x 3
// neighbourhood of a small-grid pixel, so consecutive taps of the sequence share their patch fragments.  tap = kh * 5 + kw.
__host__ __device__ __forceinline__ int jvae_t2s_tap(int t) {
    constexpr int SEQ[25] = {0, 1, 5, 6,  2, 3, 7, 8,  4, 9,  10, 11, 15, 16,  12, 13, 17, 18,  14, 19,  20, 21,  22, 23,  24};
    return SEQ[t];
}

// split-bf16 operand of conv_t2_x3.hip's 16x16x32 form (K index of one MFMA = 32 channels of ONE tap):
// Wp[kb][t][plane][kq][o][ci] = plane(W[o][c = kb*32 + kq*8 + ci][tap = jvae_t2s_tap(t)])
__device__ __forceinline__ void jvae_pack_t2s_elem(const float* __restrict__ w, __bf16* __restrict__ wp, long i,
                                                   int C, int O, int swap, int flip) {
    const int OP = jvae_pack_op(O);
    const int ci = (int)(i % 8);
    long t = i / 8;
    const int o = (int)(t % OP); t /= OP;
    const int kq = (int)(t % 4); t /= 4;
    const int ts = (int)(t % 25);
    const int kb = (int)(t / 25);
    const float v = jvae_pack_src(w, C, O, kb * 32 + kq * 8 + ci, o, jvae_t2s_tap(ts), swap, flip);
    __bf16 s[3];
    x3_split(v, s[0], s[1], s[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        wp[(((((long)kb * 25 + ts) * 3 + pl) * 4 + kq) * OP + o) * 8 + ci] = s[pl];
}

// bf16 operand of conv_b8.hip / conv_t2_b8.hip: Wp[kb][tap][half][o][ci] = bf16(W[o][c = kb*16 + half*8 + ci][tap])
__device__ __forceinline__ void jvae_pack_b8_elem(const float* __restrict__ w, __bf16* __restrict__ wp, long i,
                                                  int C, int O, int swap, int flip) {
    const int OP = jvae_pack_op(O);
    int kb, tap, half, o, ci;
    jvae_pack_unit_index(i, OP, &kb, &tap, &half, &o, &ci);
    wp[i] = (__bf16)jvae_pack_src(w, C, O, kb * 16 + half * 8 + ci, o, tap, swap, flip);
}

// pack_cache.hip.  Returns the cache slot holding the packed form of (kind, w, C, O, swap, flip), or nullptr when the cache
// is off / disarmed / full (the caller then packs into its own workspace, as without a cache).  *fresh = false: the slot is
// new - the caller must launch the pack into it on `st` now; later steps find it refreshed by jvae_pack_cache_begin.
void* jvae_pack_cache_get(int kind, const float* w, int C, int O, int swap, int flip, bool* fresh);
