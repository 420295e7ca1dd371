// Three-way bf16 split of fp32 operands for the opt-in `conv_bf3` paths (csrc/conv_bf3.hip, the expand stage of
// csrc/mbconv_rows.hip): x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), round to nearest even;
// the two subtractions are exact in fp32 and the three pieces carry 24 significand bits.
#pragma once
#include <hip/hip_runtime.h>

namespace orbit {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float bf3_f32x2 __attribute__((ext_vector_type(2)));
typedef float bf3_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: a in the low half, b in the high half
    const bf3_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// four fp32 values -> three planes of four bf16 (two dwords each)
__device__ __forceinline__ void split3(const bf3_f32x4 v, uint2& p0, uint2& p1, uint2& p2) {
    p0.x = pk_bf16(v[0], v[1]), p0.y = pk_bf16(v[2], v[3]);
    const float r0 = v[0] - bf_lo(p0.x), r1 = v[1] - bf_hi(p0.x), r2 = v[2] - bf_lo(p0.y), r3 = v[3] - bf_hi(p0.y);
    p1.x = pk_bf16(r0, r1), p1.y = pk_bf16(r2, r3);
    p2.x = pk_bf16(r0 - bf_lo(p1.x), r1 - bf_hi(p1.x)), p2.y = pk_bf16(r2 - bf_lo(p1.y), r3 - bf_hi(p1.y));
}

// eight bf16 of one plane from two split quads (the second may be absent: zeros)
__device__ __forceinline__ bf16x8 bf3_frag(uint2 lo, uint2 hi) {
    const uint4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}

}  // namespace orbit
