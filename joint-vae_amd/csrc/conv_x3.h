// Shared pieces of the split-bf16 ("x3") convolution kernels (conv_x3.hip, conv_t2_x3.hip).
#pragma once
#include "common.h"

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int x3_u32x4 __attribute__((ext_vector_type(4)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));

// v = h + m + l exactly (round-to-nearest-even at every step; the residuals are exact in fp32): 8 + 8 + 8 significand
// bits, bf16 has fp32's exponent range.  Products of two such terms are exact in fp32.
__device__ __forceinline__ void x3_split(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    float r = v - (float)h;          // exact
    m = (__bf16)r;
    r -= (float)m;                   // exact
    l = (__bf16)r;
}

// The same split for TWO values at once (round 4), the results as packed bf16 pairs (element 0 in the low half): one
// v_cvt_pk_bf16_f32 converts both, the residuals are one v_pk_add_f32 - 9 vector instructions for two values where the scalar form
// compiled to 15 (single conversions with a wasted second operand, then re-packing conversions).  Identical arithmetic: same bits.
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_split2(x3_f32x2 v, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3_bf16x2));
    x3_f32x2 r = v - x3_f32x2{__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};     // exact
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x3_bf16x2));
    r = r - x3_f32x2{__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};              // exact
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x3_bf16x2));
}
