// bf3_split.h -- the exact three-term bf16 split of an fp32 value (the arithmetic every split-bf16 contraction kernel of this
// library is built on: igemm_bf3.h explains why six bf16 products reproduce an fp32 product).
#pragma once
#include "hab_common.h"

namespace hab {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (x0, x1) = t1 + t2 + t3 exactly, element-wise, each term a bf16: t1 = rn(x), t2 = rn(x - t1), t3 = x - t1 - t2 (rn = v_cvt_pk_bf16_f32,
// round to nearest even; both residuals are exact in fp32 and the last one has at most 8 significant bits).  Returns the three PACKED
// pairs (low half = element 0).  4.5 VALU instructions per element, packing included.
__device__ __forceinline__ void bf3_split2(float x0, float x1, unsigned& w1, unsigned& w2, unsigned& w3) {
    f32x2 v; v[0] = x0; v[1] = x1;
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r;
    r[0] = v[0] - __uint_as_float(w1 << 16); r[1] = v[1] - __uint_as_float(w1 & 0xffff0000u);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 q;
    q[0] = r[0] - __uint_as_float(w2 << 16); q[1] = r[1] - __uint_as_float(w2 & 0xffff0000u);
    w3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

}  // namespace hab
