// bf3_planes.h -- the "pl32" operand format of the DMA-staged split-bf16 contractions (igemm_pl.h) and the exact 3-term split itself.
//
// Round 2 measured every split-bf16 kernel VALU-bound on the consumer side (profiles/r02_c2_sq_counters.txt: ~11 VALU per MFMA,
// matrix pipe busy 0.26-0.30): an fp32 activation is re-split into its three bf16 terms every time it is staged -- 4x per element
// by a 4x4/2 convolution, 9x by a 3x3 one, again by the weight gradient, 4.5 VALU instructions each time -- and weights are re-split
// by every workgroup of every launch although they change once per optimiser step.  Here the split is done ONCE, by the producer:
// a tensor that will be a contraction operand is stored as its three bf16 planes and consumers copy the planes global -> LDS with
// `buffer_load_dwordx4 ... lds` (no VGPR round trip, no VALU, zero padding from the buffer range check).
//
// Layout "pl32": the logical fp32 array X[0 .. n), n % 32 == 0 (NHWC activations with C % 32 == 0, packed weight matrices with
// K % 32 == 0), is cut into groups of 32 consecutive elements; group g is stored as 192 contiguous bytes
//      plane 0: rn16(x)            32 bf16   (bytes   0 ..  63)
//      plane 1: rn16(x - p0)       32 bf16   (bytes  64 .. 127)
//      plane 2: x - p0 - p1        32 bf16   (bytes 128 .. 191)        x = p0 + p1 + p2 exactly
// so the byte offset of element group e (e % 32 == 0) is 6 e = 1.5 x its fp32 byte offset: every tile / row / filter-tap offset
// the fp32 DMA functors of problems.h compute (all multiples of 128 bytes) carries over by `x + (x >> 1)`.  A k-tile of BK = 32
// reduction elements of one operand row is one group: three 64-byte rows of an LDS image, 4 DMA lanes each.
#pragma once
#include "hab_common.h"

namespace hab {

typedef unsigned short pl16;  // one bf16 bit pattern
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int PL_GROUP = 32;         // elements per group
constexpr int PL_GROUP_U16 = 96;     // pl16 units per group (3 planes x 32)
// pl16 index of plane `p` of logical element `idx`
__host__ __device__ inline size_t pl_index(size_t idx, int p) { return (idx >> 5) * PL_GROUP_U16 + (size_t)p * 32 + (idx & 31); }
// pl16 units of a tensor of n logical elements (n % 32 == 0)
__host__ __device__ inline size_t pl_units(size_t n) { return (n >> 5) * PL_GROUP_U16; }

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

// Four consecutive logical elements idx .. idx+3 (idx % 4 == 0) -> 8 bytes in each of the three planes of their group.
// Host build (tests/hostcheck executes the epilogue functors on the CPU): software round-to-nearest-even, same arithmetic.
__host__ __device__ inline void pl_store4(pl16* __restrict__ base, size_t idx, const f32x4 v) {
    pl16* dst = base + pl_index(idx, 0);
#ifdef __HIP_DEVICE_COMPILE__
    unsigned a1, a2, a3, b1, b2, b3;
    bf3_split2(v[0], v[1], a1, a2, a3);
    bf3_split2(v[2], v[3], b1, b2, b3);
    u32x2 w1, w2, w3;
    w1[0] = a1; w1[1] = b1; w2[0] = a2; w2[1] = b2; w3[0] = a3; w3[1] = b3;
    *reinterpret_cast<u32x2*>(dst) = w1;
    *reinterpret_cast<u32x2*>(dst + 32) = w2;
    *reinterpret_cast<u32x2*>(dst + 64) = w3;
#else
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        for (int p = 0; p < 3; ++p) {
            unsigned bits;
            __builtin_memcpy(&bits, &x, 4);
            const unsigned r = (bits + 0x7fffu + ((bits >> 16) & 1u)) >> 16;
            dst[p * 32 + e] = (pl16)r;
            const unsigned back = r << 16;
            float t;
            __builtin_memcpy(&t, &back, 4);
            x -= t;
        }
    }
#endif
}

// p0 + p1 + p2 of four consecutive elements (exact: the sum is the fp32 value that was split)
__host__ __device__ inline f32x4 pl_load4(const pl16* __restrict__ base, size_t idx) {
    const pl16* src = base + pl_index(idx, 0);
    f32x4 v;
    for (int e = 0; e < 4; ++e) {
        float s = 0.f;
        for (int p = 2; p >= 0; --p) {  // smallest term first: every partial sum is exact
            const unsigned bits = (unsigned)src[p * 32 + e] << 16;
            float t;
            __builtin_memcpy(&t, &bits, 4);
            s += t;
        }
        v[e] = s;
    }
    return v;
}
// (x > 0) of four consecutive elements from plane 0 alone: rn16 keeps sign and zero-ness of every normal fp32
__host__ __device__ inline void pl_positive4(const pl16* __restrict__ base, size_t idx, bool (&pos)[4]) {
    const u32x2 w = *reinterpret_cast<const u32x2*>(base + pl_index(idx, 0));
    for (int e = 0; e < 4; ++e) {
        const unsigned h = (w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
        pos[e] = (h & 0x8000u) == 0 && (h & 0x7fffu) != 0;
    }
}

}  // namespace hab
