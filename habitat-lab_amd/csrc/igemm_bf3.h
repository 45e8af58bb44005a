// igemm_bf3.h -- the r-contiguous x r-contiguous contractions (convolution forward / data gradient, Linear forward) on the bf16
// matrix pipe with fp32-equivalent arithmetic.
//
// gfx950 issues v_mfma_f32_32x32x2_f32 at the fp32 VECTOR rate (157 TFLOP/s chip peak) and -- measured in round 1
// (tools/ubench/mfma_valu_overlap.hip) -- that instruction shares the SIMD's issue with every other instruction, so a kernel's time is
// MFMA time + everything else.  v_mfma_f32_32x32x16_bf16 runs on the matrix pipe proper: 16x the rate, and VALU / LDS / address work
// of other waves issues beside it.  An fp32 operand is split EXACTLY into three bf16 terms by rounding to nearest
//      a1 = rn16(a),  a2 = rn16(a - a1),  a3 = a - a1 - a2      (|a2| <= 2^-9 |a|, |a3| <= 2^-17 |a|, a = a1 + a2 + a3 exactly)
// and the product a*b is accumulated from the six partial products of weight >= 2^-18 relative
//      a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2)
// (each product exact in the fp32 accumulator; the three dropped terms a2 b3 + a3 b2 + a3 b3 are <= 2^-25 |a b| and of either sign --
// below the rounding of the fp32 product itself.  Truncation instead of rounding would leave all dropped terms with the sign of a*b: a
// bias that grows with K instead of averaging out; measured 12x the fp32 path's error on ResNet18 gradients at 4096 frames).  Six bf16
// MFMAs of K = 16 replace eight fp32 MFMAs of K = 2: 6/16 of the matrix-pipe time, which no longer blocks the rest of the instruction
// stream.  The split costs 4.5 VALU instructions per staged element (v_cvt_pk_bf16_f32 packs as it rounds), paid once per element when
// the tile is written to LDS.
//
// LDS image: three planes per operand, [rows][BK + 8] bf16 (80-byte row pitch: the 16-byte fragment reads of 16 lanes fall on 16
// distinct 4-bank windows).  Lane l of a wave supplies A[i = l & 31][k = 8 (l >> 5) .. +7] of a K = 16 step: one ds_read_b128 per
// plane and operand tile.  The accumulator layout equals v_mfma_f32_32x32x2_f32's, so the problems' epilogues are shared with
// igemm.h unchanged (incl. the transposed-accumulator vector epilogue).
#pragma once
#include "igemm.h"
#include "bf3_split.h"

namespace hab {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BF3_BKP = IGEMM_BK + 8;  // bf16 elements per LDS row

// the exact 3-term split (bf3_split2) lives in bf3_split.h
// scalar form: the three bf16 bit patterns in the low 16 bits
__device__ __forceinline__ void bf3_split(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
    unsigned w1, w2, w3;
    bf3_split2(x, 0.f, w1, w2, w3);
    h1 = w1 & 0xffffu; h2 = w2 & 0xffffu; h3 = w3 & 0xffffu;
}
// packs the bf16 halves (upper 16 bits) of two fp32 patterns whose low halves are zero or ignorable: low half = first element
__device__ __forceinline__ unsigned bf3_pack(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// four consecutive-k fp32 values -> 8 bytes (4 bf16) in each of the three planes
__device__ __forceinline__ void bf3_store4(const f32x4 v, unsigned short* p1, unsigned short* p2, unsigned short* p3) {
    unsigned a1, a2, a3, b1, b2, b3;
    bf3_split2(v[0], v[1], a1, a2, a3);
    bf3_split2(v[2], v[3], b1, b2, b3);
    u32x2 w1, w2, w3;
    w1[0] = a1; w1[1] = b1; w2[0] = a2; w2[1] = b2; w3[0] = a3; w3[1] = b3;
    *reinterpret_cast<u32x2*>(p1) = w1;
    *reinterpret_cast<u32x2*>(p2) = w2;
    *reinterpret_cast<u32x2*>(p3) = w3;
}

// k-run of an i/j-contiguous operand: a staging unit is KV consecutive rows x R consecutive k, gathered as R row-vectors and
// transposed in registers (the split pairs neighbours along k, so the packed bf16 pairs come out k-contiguous for free).
constexpr int bf3_run(int rows, int kv, int bk, int nt) {
    int r = bk * rows / (kv * nt);
    int p = 2;
    while (p * 2 <= r && p < 8) p *= 2;
    return p;
}

template <class P, int TM, int TN, int WM, int WN>
struct IgemmBf3Cfg {
    static constexpr int NT = WM * WN * 64;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = IGEMM_BK;
    static constexpr int KV = AKv<P>::value;
    static constexpr int A_PLANE = BM * BF3_BKP, B_PLANE = BN * BF3_BKP;  // bf16 elements
    static constexpr int RA = bf3_run(BM, KV, BK, NT), RB = bf3_run(BN, 4, BK, NT);
    static constexpr int AKQ = P::A_RC ? BK / KV : BK / RA;  // units per row (RC) / k-runs per k-tile (IC)
    static constexpr int BKQ = P::B_RC ? BK / 4 : BK / RB;
    static constexpr int A_TOTAL = P::A_RC ? BM * BK / KV : (BM / KV) * AKQ;
    static constexpr int B_TOTAL = P::B_RC ? BN * BK / 4 : (BN / 4) * BKQ;
    static constexpr int A_UNITS = (A_TOTAL + NT - 1) / NT, B_UNITS = (B_TOTAL + NT - 1) / NT;
    static constexpr int A_RAWS = P::A_RC ? 1 : RA, B_RAWS = P::B_RC ? 1 : RB;
    // both operands i/j-contiguous (weight gradients): the per-k gather keys (pixel decode: two divisions + 64-bit offsets, ~30 VALU
    // each) are the same for every row unit, so each k-tile's BK keys are computed ONCE per workgroup into LDS (two buffers) instead
    // of R times per thread
    // (measured: +5..13 % on the N >= 64 weight gradients; the 32-column tiles lose 6 % to the extra LDS traffic and stay as they were)
    static constexpr bool KSH = !P::A_RC && !P::B_RC && (TN * WN >= 2);
    static constexpr size_t PLANES_BYTES = (size_t)3 * (A_PLANE + B_PLANE) * 2;
    static constexpr size_t KEYS_BYTES = KSH ? (size_t)2 * BK * (sizeof(typename P::AKey) + sizeof(typename P::BKey)) : 0;
    static constexpr size_t LDS_BYTES = PLANES_BYTES + KEYS_BYTES;
    static_assert(!P::A_RC || NT % (BK / KV) == 0, "A unit mapping");
    static_assert(!P::B_RC || NT % (BK / 4) == 0, "B unit mapping");
    static_assert(BM % KV == 0, "A rows per unit");
};

template <int R>
__device__ __forceinline__ void bf3_store_run(const float (&x)[R], unsigned short* p1, unsigned short* p2, unsigned short* p3) {
    unsigned w1[R / 2], w2[R / 2], w3[R / 2];
#pragma unroll
    for (int q = 0; q < R / 2; ++q) bf3_split2(x[2 * q], x[2 * q + 1], w1[q], w2[q], w3[q]);
    if constexpr (R == 2) {
        *reinterpret_cast<unsigned*>(p1) = w1[0]; *reinterpret_cast<unsigned*>(p2) = w2[0]; *reinterpret_cast<unsigned*>(p3) = w3[0];
    } else if constexpr (R == 4) {
        u32x2 a, b, c;
        a[0] = w1[0]; a[1] = w1[1]; b[0] = w2[0]; b[1] = w2[1]; c[0] = w3[0]; c[1] = w3[1];
        *reinterpret_cast<u32x2*>(p1) = a; *reinterpret_cast<u32x2*>(p2) = b; *reinterpret_cast<u32x2*>(p3) = c;
    } else {
        static_assert(R == 8, "k-run");
        u32x4 a, b, c;
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = w1[q]; b[q] = w2[q]; c[q] = w3[q]; }
        *reinterpret_cast<u32x4*>(p1) = a; *reinterpret_cast<u32x4*>(p2) = b; *reinterpret_cast<u32x4*>(p3) = c;
    }
}

template <class P, int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64) igemm_bf3_kernel(const P p, const int k_per_split, float* __restrict__ partial, const int sign_schedule, const int nsplit, const int ablate) {
    using Cfg = IgemmBf3Cfg<P, TM, TN, WM, WN>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, KV = Cfg::KV;
    constexpr int A_UNITS = Cfg::A_UNITS, B_UNITS = Cfg::B_UNITS, A_TOTAL = Cfg::A_TOTAL, B_TOTAL = Cfg::B_TOTAL;
    constexpr int AKQ = Cfg::AKQ, BKQ = Cfg::BKQ, RA = Cfg::RA, RB = Cfg::RB;

    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* As = smem16;                        // planes 0..2: [BM][BF3_BKP]
    unsigned short* Bs = smem16 + 3 * Cfg::A_PLANE;     // planes 0..2: [BN][BF3_BKP]

    typename P::AKey* akeys = reinterpret_cast<typename P::AKey*>(reinterpret_cast<char*>(smem16) + Cfg::PLANES_BYTES);  // [2][BK]
    typename P::BKey* bkeys = reinterpret_cast<typename P::BKey*>(akeys + 2 * BK);                                       // [2][BK]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

    const int nt_m = cdiv(p.M, BM), nt_n = cdiv(p.N, BN);
    const int ntiles = nt_m * nt_n;
    int tile, kz;
    if (nsplit == 1) {  // XCD-contiguous runs of M-tiles (igemm.h)
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        kz = 0;
    } else {
        // Split-K (weight gradients: a handful of output tiles, hundreds of K slices).  All output tiles of ONE K slice read the same
        // rows of both operands (the taps of a filter are shifted views of the same pixels), so they are placed on the same XCD, in
        // consecutive dispatch slots: workgroups are dealt round-robin over the 8 XCDs, hence slice kz -> XCD kz % 8.  With the
        // slices of a tile spread over all XCDs instead, every private L2 fetched the operands again: 2.5-5.6x the algorithmic HBM
        // bytes, at ~4 TB/s -- the weight gradients were HBM-bound on re-reads (profiles/r02_c2_hbm_traffic_before_xcd_slices.txt).
        if (nsplit >= 16) {
            const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
            tile = slot % ntiles;
            kz = (slot / ntiles) * 8 + xcd;
            if (kz >= nsplit) return;
        } else {  // few slices: grouping would leave XCDs idle
            tile = blockIdx.x % ntiles;
            kz = blockIdx.x / ntiles;
        }
    }
    // Which coordinate runs fastest inside an XCD's contiguous run of tiles decides which operand that XCD's L2 fetches once: with
    // tile_n fastest an XCD covers a few rows of M-tiles and ALL column tiles (every XCD reads all of B, 1/8 of A: right for the
    // convolutions, M = pixels >> N = channels); when the COLUMN operand is the large one -- the visual fc's data and weight gradients:
    // 4 row tiles x 196 column tiles over a 51 MB weight matrix / 205 MB of activations -- tile_m runs fastest, so the row tiles that
    // share a column block sit on one XCD in consecutive dispatch slots (round 4 counters: 1.69x / 3.5x the algorithmic HBM bytes
    // with every column block fetched by four XCDs).  A bijection either way; the sign schedule is keyed by tile coordinates.
    const bool m_fastest = nsplit == 1 && nt_m < nt_n;
    const int tile_n = m_fastest ? tile / nt_m : tile % nt_n, tile_m = m_fastest ? tile % nt_m : tile / nt_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int k_begin = kz * k_per_split;
    const int k_end = min(p.K, k_begin + k_per_split);
    const int ntk = max(0, cdiv(k_end - k_begin, BK));

    // unit u = t + NT*j.   RC operand: row u / AKQ, k-offset (u % AKQ) * KV.   IC operand: k-run u % AKQ (fastest: the lanes of an
    // LDS write group then cover whole rows of the image -> conflict-free), rows (u / AKQ) * KV .. +KV-1.
    typename P::ACtx actx[A_UNITS];
    typename P::BCtx bctx[B_UNITS];
#pragma unroll
    for (int j = 0; j < A_UNITS; ++j) {
        const int u = t + NT * j;
        actx[j] = P::A_RC ? p.a_ctx(m0 + u / AKQ) : p.a_ctx(m0 + (u / AKQ) * KV);
    }
#pragma unroll
    for (int j = 0; j < B_UNITS; ++j) {
        const int u = t + NT * j;
        bctx[j] = P::B_RC ? p.b_ctx(n0 + (u >> 3)) : p.b_ctx(n0 + (u / BKQ) * 4);
    }
    constexpr bool CS = ColsumB<P>::value;
    bool has_cs = false;
    if constexpr (CS) has_cs = (p.colsum != nullptr);
    const bool do_cs = has_cs && (tile_m == 0);
    const int MP = p.M + (has_cs ? 1 : 0);
    f32x4 cs[B_UNITS];
#pragma unroll
    for (int j = 0; j < B_UNITS; ++j) cs[j] = zero4();

    typename P::ARaw araw[A_UNITS][Cfg::A_RAWS];
    typename P::BRaw braw[B_UNITS][Cfg::B_RAWS];
    auto a_k = [&](int kt, int j, int r) {
        const int u = t + NT * j;
        return k_begin + kt * BK + (P::A_RC ? (u % AKQ) * KV : (u % AKQ) * RA + r);
    };
    auto b_k = [&](int kt, int j, int r) {
        const int u = t + NT * j;
        return k_begin + kt * BK + (P::B_RC ? (u & 7) * 4 : (u % BKQ) * RB + r);
    };
    auto make_keys = [&](int kt) {  // Cfg::KSH: keys of k-tile kt into key buffer kt & 1 (visible after the next barrier)
        if constexpr (Cfg::KSH) {
            if (t < BK) {
                const typename P::KCtx kc = p.k_ctx(k_begin + kt * BK, k_end);
                akeys[(kt & 1) * BK + t] = p.a_key(kc, k_begin + kt * BK + t, k_end);
                bkeys[(kt & 1) * BK + t] = p.b_key(kc, k_begin + kt * BK + t, k_end);
            }
        }
    };
    auto fetch = [&](int kt) {
        if (ablate & 1) return;  // development (HAB_BF3_ABLATE, tools/ablate_layers.sh): no operand gathers
        const typename P::KCtx kc = p.k_ctx(k_begin + kt * BK, k_end);
        if constexpr (P::A_RC) {
            const typename P::AKey ak = p.a_key(kc, a_k(kt, 0, 0), k_end);
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j)
                if (A_TOTAL % NT == 0 || t + NT * j < A_TOTAL) araw[j][0] = p.a_fetch(actx[j], kc, ak);
        } else {
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                // NT % AKQ == 0 is not required: the key depends on the unit
#pragma unroll
                for (int j = 0; j < A_UNITS; ++j)
                    if (A_TOTAL % NT == 0 || t + NT * j < A_TOTAL) {
                        if constexpr (Cfg::KSH) araw[j][r] = p.a_fetch(actx[j], kc, akeys[(kt & 1) * BK + ((t + NT * j) % AKQ) * RA + r]);
                        else araw[j][r] = p.a_fetch(actx[j], kc, p.a_key(kc, a_k(kt, j, r), k_end));
                    }
            }
        }
        if constexpr (P::B_RC) {
            const typename P::BKey bk = p.b_key(kc, b_k(kt, 0, 0), k_end);
#pragma unroll
            for (int j = 0; j < B_UNITS; ++j)
                if (B_TOTAL % NT == 0 || t + NT * j < B_TOTAL) braw[j][0] = p.b_fetch(bctx[j], kc, bk);
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < B_UNITS; ++j)
                    if (B_TOTAL % NT == 0 || t + NT * j < B_TOTAL) {
                        if constexpr (Cfg::KSH) braw[j][r] = p.b_fetch(bctx[j], kc, bkeys[(kt & 1) * BK + ((t + NT * j) % BKQ) * RB + r]);
                        else braw[j][r] = p.b_fetch(bctx[j], kc, p.b_key(kc, b_k(kt, j, r), k_end));
                    }
        }
    };
    auto stage = [&](int kt, const unsigned sgn) {  // registers -> split -> three bf16 planes; sgn = 0x80000000: B enters negated
        if (ablate & 8) return;  // development: no split, no LDS writes
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            const int u = t + NT * j;
            if (A_TOTAL % NT == 0 || u < A_TOTAL) {
                if constexpr (P::A_RC) {
                    f32x4 v[KV / 4];
                    p.a_cvt(actx[j], araw[j][0], a_k(kt, j, 0), k_end, v);
                    unsigned short* dst = As + (u / AKQ) * BF3_BKP + (u % AKQ) * KV;
#pragma unroll
                    for (int q = 0; q < KV / 4; ++q) bf3_store4(v[q], dst + 4 * q, dst + Cfg::A_PLANE + 4 * q, dst + 2 * Cfg::A_PLANE + 4 * q);
                } else {
                    f32x4 v[RA][KV / 4];
#pragma unroll
                    for (int r = 0; r < RA; ++r) p.a_cvt(actx[j], araw[j][r], a_k(kt, j, r), k_end, v[r]);
                    unsigned short* dst = As + ((u / AKQ) * KV) * BF3_BKP + (u % AKQ) * RA;
#pragma unroll
                    for (int e = 0; e < KV; ++e) {
                        float x[RA];
#pragma unroll
                        for (int r = 0; r < RA; ++r) x[r] = v[r][e >> 2][e & 3];
                        bf3_store_run<RA>(x, dst + e * BF3_BKP, dst + Cfg::A_PLANE + e * BF3_BKP, dst + 2 * Cfg::A_PLANE + e * BF3_BKP);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) {
            const int u = t + NT * j;
            if (B_TOTAL % NT == 0 || u < B_TOTAL) {
                if constexpr (P::B_RC) {
                    const f32x4 v = p.b_cvt(braw[j][0]);
                    unsigned short* dst = Bs + (u >> 3) * BF3_BKP + (u & 7) * 4;
                    f32x4 vs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vs[e] = __uint_as_float(__float_as_uint(v[e]) ^ sgn);
                    bf3_store4(vs, dst, dst + Cfg::B_PLANE, dst + 2 * Cfg::B_PLANE);
                } else {
                    f32x4 v[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        v[r] = p.b_cvt(braw[j][r]);
                        if constexpr (CS) {
                            if (do_cs) cs[j] += v[r];
                        }
                    }
                    unsigned short* dst = Bs + ((u / BKQ) * 4) * BF3_BKP + (u % BKQ) * RB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x[RB];
#pragma unroll
                        for (int r = 0; r < RB; ++r) x[r] = __uint_as_float(__float_as_uint(v[r][e]) ^ sgn);
                        bf3_store_run<RB>(x, dst + e * BF3_BKP, dst + Cfg::B_PLANE + e * BF3_BKP, dst + 2 * Cfg::B_PLANE + e * BF3_BKP);
                    }
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    // Sign schedule.  The bf16 MFMA aligns its 16 products and the accumulator in a ~32-bit window and TRUNCATES what falls below it
    // (two's complement: towards -infinity) before the fp32 result is rounded -- measured: every output is low by ~0.0025 ulp per
    // instruction, whatever its sign.  Invisible per element (<= 0.3 ulp for K ~ 300), but coherent: a bias gradient that sums 10^7
    // such outputs was off by 7e-5 relative.  Accumulating the NEGATED sum (B enters negated, the accumulator is negated before the
    // epilogue) makes the truncation push the other way, so output tiles (checkerboard) and split-K slices alternate the
    // convention: the residual per-output bias is no longer coherent across the tensor or across the slabs, and sums over it
    // average out like ordinary rounding noise.  Cost: one sign flip per B element while it is staged and one pass over the
    // accumulator, in every second workgroup.
    const bool flip_all = sign_schedule && (((tile_m + tile_n + kz) & 1) != 0);
    const unsigned sgn = flip_all ? 0x80000000u : 0u;
    auto negate_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = -acc[i][j][v];
    };
    if (ntk > 0) {
        if constexpr (Cfg::KSH) { make_keys(0); __syncthreads(); }
        fetch(0);
    }
    for (int kt = 0; kt < ntk; ++kt) {
        if (kt + 1 < ntk) make_keys(kt + 1);  // other key buffer: read by fetch(kt + 1) behind the barrier below
        stage(kt, sgn);
        __syncthreads();
        if (kt + 1 < ntk) fetch(kt + 1);
        if (!(ablate & 2))  // development: no fragment reads, no MFMAs
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned short* src = As + ((wm * TM + i) * 32 + li) * BF3_BKP + c * 16 + hi * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(src + pl * Cfg::A_PLANE);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned short* src = Bs + ((wn * TN + j) * 32 + li) * BF3_BKP + c * 16 + hi * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(src + pl * Cfg::B_PLANE);
            }
            // six partial products, smallest weight first
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if constexpr (EpiV4<P>::value)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][PB[q]], af[i][PA[q]], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[q]], bf[j][PB[q]], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    if (flip_all) negate_acc();
    if ((ablate & 4) && acc[0][0][0] != 12345.f) return;  // development: no output traffic

    // ---- epilogue: identical to igemm_kernel's (same accumulator layout) ----
    const bool split = nsplit > 1;
    if constexpr (CS) {
        if (do_cs) {  // block-uniform: column c = the BKQ units of column quad c >> 2, summed in unit order through LDS
            float* red = reinterpret_cast<float*>(smem16);
#pragma unroll
            for (int j = 0; j < B_UNITS; ++j)
                if (B_TOTAL % NT == 0 || t + NT * j < B_TOTAL) *reinterpret_cast<f32x4*>(red + (size_t)(t + NT * j) * 4) = cs[j];
            __syncthreads();
            if (t < BN && n0 + t < p.N) {
                float s_ = 0.f;
                for (int q = 0; q < BKQ; ++q) s_ += red[((t >> 2) * BKQ + q) * 4 + (t & 3)];
                if (split)
                    partial[((size_t)kz * MP + p.M) * p.N + n0 + t] = s_;
                else
                    p.store_colsum(n0 + t, s_);
            }
        }
    }
    if constexpr (EpiV4<P>::value) {
        if (split)
            igemm_partial_v4<P, TM, TN>(p, acc, partial + (size_t)kz * MP * p.N, m0 + wm * TM * 32, n0 + wn * TN * 32, li, hi);
        else
            igemm_epilogue_v4<P, TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, li, hi);
        return;
    }
    if (split) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = m0 + (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
                    if (row < p.M && col < p.N) partial[((size_t)kz * MP + row) * p.N + col] = acc[i][j][v];
                }
            }
        return;
    }
    typename P::EpiCol ecol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) ecol[j] = p.epi_col(n0 + (wn * TN + j) * 32 + li);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typename P::EpiRow erow[4];
            typename P::EpiAux eaux[4][TN];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                erow[q] = p.epi_row(m0 + (wm * TM + i) * 32 + q + 8 * g + 4 * hi);
#pragma unroll
                for (int j = 0; j < TN; ++j) eaux[q][j] = p.epi_fetch(erow[q], ecol[j]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j) p.epi_store(erow[q], ecol[j], eaux[q][j], acc[i][j][g * 4 + q]);
        }
}

template <class P, int TM, int TN, int WM, int WN>
inline int igemm_bf3_launch(const P& p, float* ws, size_t ws_floats, int target_blocks, hipStream_t stream) {
    using Cfg = IgemmBf3Cfg<P, TM, TN, WM, WN>;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return HAB_ERR_ARG;
    const IgemmPlan pl = igemm_plan(Cfg::BM, Cfg::BN, p.M, p.N, p.K, target_blocks, 4096, ws ? ws_floats : 0);
    auto kern = igemm_bf3_kernel<P, TM, TN, WM, WN>;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = (Cfg::LDS_BYTES > 64 * 1024) ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) : hipSuccess;
    if (attr_err != hipSuccess) return (int)attr_err;
    const int ntiles = cdiv(p.M, Cfg::BM) * cdiv(p.N, Cfg::BN);
    const int grid = pl.splits >= 16 ? ntiles * ((pl.splits + 7) / 8 * 8) : ntiles * pl.splits;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");  // development: measure the cost of the sign schedule
    static const int ablate = hab_env_int("HAB_BF3_ABLATE", 0);
    kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, pl.k_per_split, ws, sign_schedule, pl.splits, ablate);
    HAB_LAUNCH_CHECK();
    if (pl.splits > 1) {
        igemm_splitk_reduce<P>(p, ws, pl.splits, stream);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

}  // namespace hab
