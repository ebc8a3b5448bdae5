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
