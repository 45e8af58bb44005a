// igemm_bf3_ws.h -- wave-specialised form of igemm_bf3_kernel (EXPERIMENT: built and ISA-checked offline, never run; see
// tools/experiments/README.md).
//
// igemm_bf3_kernel alternates, in every wave, a staging phase (gather -> exact three-way bf16 split -> LDS: ~10 VALU instructions per
// MFMA, profiles/r02_layers_sq_counters.txt) and an MFMA phase, with two barriers per k-tile; the matrix pipe is busy ~0.3 of the time
// and the only overlap of the two phases comes from a second workgroup on the CU that happens to be in the other phase.  Here the
// roles are given to different waves of ONE workgroup:
//   * PW producer waves (threads 0 .. 64 PW - 1) gather k-tile kt + 2 into registers, split k-tile kt + 1 and write it to LDS image
//     (kt + 1) & 1 -- no MFMA, no fragment reads;
//   * WM x WN consumer waves read the fragments of image kt & 1 and issue the 6 TM TN MFMAs of each 16-deep k-group -- no VALU
//     beyond address arithmetic.
// The hardware places wave w of a workgroup on SIMD w % 4, so with PW = 4 and four consumer waves every SIMD holds one producer and
// one consumer: the bf16 MFMA runs on the matrix pipe while the other wave's VALU / LDS / VMEM instructions issue beside it (measured
// for this instruction class in round 1: tools/ubench/mfma_valu_overlap.hip).  One barrier per k-tile: the image written during step
// kt is the one NOT being read, and the gather registers are double-buffered so that a load issued in step kt is consumed in step
// kt + 1 (a whole k-tile of latency hiding instead of the length of an MFMA phase).  Arithmetic, operand order, sign schedule and
// epilogues are those of igemm_bf3_kernel: results are bit-identical for the same tile shape.
#pragma once
#include "igemm_bf3.h"

namespace hab {

template <class P, int TM, int TN, int WM, int WN, int PW>
struct IgemmBf3WsCfg {
    static constexpr int NC = WM * WN * 64, NP = PW * 64, NT = NC + NP;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = IGEMM_BK;
    static constexpr int KV = AKv<P>::value;
    static constexpr int A_PLANE = BM * BF3_BKP, B_PLANE = BN * BF3_BKP;  // bf16 elements
    static constexpr int IMG = 3 * (A_PLANE + B_PLANE);                   // one LDS image (bf16 elements)
    static constexpr int RA = bf3_run(BM, KV, BK, NP), RB = bf3_run(BN, 4, BK, NP);
    static constexpr int AKQ = P::A_RC ? BK / KV : BK / RA;
    static constexpr int BKQ = P::B_RC ? BK / 4 : BK / RB;
    static constexpr int A_TOTAL = P::A_RC ? BM * BK / KV : (BM / KV) * AKQ;
    static constexpr int B_TOTAL = P::B_RC ? BN * BK / 4 : (BN / 4) * BKQ;
    static constexpr int A_UNITS = (A_TOTAL + NP - 1) / NP, B_UNITS = (B_TOTAL + NP - 1) / NP;
    static constexpr int A_RAWS = P::A_RC ? 1 : RA, B_RAWS = P::B_RC ? 1 : RB;
    static constexpr bool KSH = !P::A_RC && !P::B_RC && (TN * WN >= 2);  // shared gather keys (igemm_bf3.h); THREE buffers here
    static constexpr size_t PLANES_BYTES = (size_t)2 * IMG * 2;
    static constexpr size_t KEYS_BYTES = KSH ? (size_t)3 * BK * (sizeof(typename P::AKey) + sizeof(typename P::BKey)) : 0;
    static constexpr size_t LDS_BYTES = PLANES_BYTES + KEYS_BYTES;
    static_assert(!P::A_RC || NP % (BK / KV) == 0, "A unit mapping");
    static_assert(!P::B_RC || NP % (BK / 4) == 0, "B unit mapping");
    static_assert(BM % KV == 0, "A rows per unit");
    static_assert(BN <= NP && BK <= NP, "column-sum / key threads");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <class P, int TM, int TN, int WM, int WN, int PW>
__global__ void __launch_bounds__((WM * WN + PW) * 64)
igemm_bf3_ws_kernel(const P p, const int k_per_split, float* __restrict__ partial, const int sign_schedule, const int nsplit) {
    using Cfg = IgemmBf3WsCfg<P, TM, TN, WM, WN, PW>;
    constexpr int NP = Cfg::NP, BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, KV = Cfg::KV;
    constexpr int A_UNITS = Cfg::A_UNITS, B_UNITS = Cfg::B_UNITS, A_TOTAL = Cfg::A_TOTAL, B_TOTAL = Cfg::B_TOTAL;
    constexpr int AKQ = Cfg::AKQ, BKQ = Cfg::BKQ, RA = Cfg::RA, RB = Cfg::RB;

    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    typename P::AKey* akeys = reinterpret_cast<typename P::AKey*>(reinterpret_cast<char*>(smem16) + Cfg::PLANES_BYTES);  // [3][BK]
    typename P::BKey* bkeys = reinterpret_cast<typename P::BKey*>(akeys + 3 * BK);                                       // [3][BK]

    const int t = threadIdx.x;  // producers: t < NP (the staging unit index of igemm_bf3.h with NT = NP)
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // scalar: the role branches below are uniform branches
    const bool producer = wave < PW;
    const int cw = producer ? 0 : wave - PW;
    const int wm = cw / WN, wn = cw % WN;
    const int li = lane & 31, hi = lane >> 5;

    const int nt_m = cdiv(p.M, BM), nt_n = cdiv(p.N, BN);
    const int ntiles = nt_m * nt_n;
    int tile, kz;
    if (nsplit == 1) {
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        kz = 0;
    } else if (nsplit >= 16) {  // all output tiles of a K slice on one XCD (igemm_bf3.h)
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        tile = slot % ntiles;
        kz = (slot / ntiles) * 8 + xcd;
        if (kz >= nsplit) return;
    } else {
        tile = blockIdx.x % ntiles;
        kz = blockIdx.x / ntiles;
    }
    const bool m_fastest = nsplit == 1 && nt_m < nt_n;  // the large operand is the column one: see igemm_bf3.h
    const int tile_n = m_fastest ? tile / nt_m : tile % nt_n, tile_m = m_fastest ? tile % nt_m : tile / nt_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int k_begin = kz * k_per_split;
    const int k_end = min(p.K, k_begin + k_per_split);
    const int ntk = max(0, cdiv(k_end - k_begin, BK));

    constexpr bool CS = ColsumB<P>::value;
    bool has_cs = false;
    if constexpr (CS) has_cs = (p.colsum != nullptr);
    const bool do_cs = has_cs && (tile_m == 0);
    const int MP = p.M + (has_cs ? 1 : 0);
    const bool flip_all = sign_schedule && (((tile_m + tile_n + kz) & 1) != 0);
    const unsigned sgn = flip_all ? 0x80000000u : 0u;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // ------------------------------------------------------------------ producer waves: their own loop, so that the register
    // allocation is the maximum of the two roles, not the sum; the barriers below pair one-to-one with the consumers' (same count
    // on both sides for every ntk; `wave` is scalar, the branch is not a divergence)
    if (producer) {
        typename P::ACtx actx[A_UNITS];
        typename P::BCtx bctx[B_UNITS];
        f32x4 cs[B_UNITS];
        typename P::ARaw araw[2][A_UNITS][Cfg::A_RAWS];
        typename P::BRaw braw[2][B_UNITS][Cfg::B_RAWS];
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            const int u = t + NP * j;
            actx[j] = P::A_RC ? p.a_ctx(m0 + u / AKQ) : p.a_ctx(m0 + (u / AKQ) * KV);
        }
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) {
            const int u = t + NP * j;
            bctx[j] = P::B_RC ? p.b_ctx(n0 + (u >> 3)) : p.b_ctx(n0 + (u / BKQ) * 4);
            cs[j] = zero4();
        }
        auto a_k = [&](int kt, int j, int r) {
            const int u = t + NP * j;
            return k_begin + kt * BK + (P::A_RC ? (u % AKQ) * KV : (u % AKQ) * RA + r);
        };
        auto b_k = [&](int kt, int j, int r) {
            const int u = t + NP * j;
            return k_begin + kt * BK + (P::B_RC ? (u & 7) * 4 : (u % BKQ) * RB + r);
        };
        auto make_keys = [&](int kt, int kb) {  // Cfg::KSH: keys of k-tile kt into key buffer kb
            if constexpr (Cfg::KSH) {
                if (t < BK) {
                    const typename P::KCtx kc = p.k_ctx(k_begin + kt * BK, k_end);
                    akeys[kb * BK + t] = p.a_key(kc, k_begin + kt * BK + t, k_end);
                    bkeys[kb * BK + t] = p.b_key(kc, k_begin + kt * BK + t, k_end);
                }
            }
        };
        // gather of k-tile kt into register set S (keys, when shared, from key buffer kb)
        auto fetch = [&](auto s_, int kt, int kb) {
            constexpr int S = decltype(s_)::value;
            const typename P::KCtx kc = p.k_ctx(k_begin + kt * BK, k_end);
            if constexpr (P::A_RC) {
                const typename P::AKey ak = p.a_key(kc, a_k(kt, 0, 0), k_end);
#pragma unroll
                for (int j = 0; j < A_UNITS; ++j)
                    if (A_TOTAL % NP == 0 || t + NP * j < A_TOTAL) araw[S][j][0] = p.a_fetch(actx[j], kc, ak);
            } else {
#pragma unroll
                for (int r = 0; r < RA; ++r)
#pragma unroll
                    for (int j = 0; j < A_UNITS; ++j)
                        if (A_TOTAL % NP == 0 || t + NP * j < A_TOTAL) {
                            if constexpr (Cfg::KSH) araw[S][j][r] = p.a_fetch(actx[j], kc, akeys[kb * BK + ((t + NP * j) % AKQ) * RA + r]);
                            else araw[S][j][r] = p.a_fetch(actx[j], kc, p.a_key(kc, a_k(kt, j, r), k_end));
                        }
            }
            if constexpr (P::B_RC) {
                const typename P::BKey bk = p.b_key(kc, b_k(kt, 0, 0), k_end);
#pragma unroll
                for (int j = 0; j < B_UNITS; ++j)
                    if (B_TOTAL % NP == 0 || t + NP * j < B_TOTAL) braw[S][j][0] = p.b_fetch(bctx[j], kc, bk);
            } else {
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int j = 0; j < B_UNITS; ++j)
                        if (B_TOTAL % NP == 0 || t + NP * j < B_TOTAL) {
                            if constexpr (Cfg::KSH) braw[S][j][r] = p.b_fetch(bctx[j], kc, bkeys[kb * BK + ((t + NP * j) % BKQ) * RB + r]);
                            else braw[S][j][r] = p.b_fetch(bctx[j], kc, p.b_key(kc, b_k(kt, j, r), k_end));
                        }
            }
        };
        // register set S (k-tile kt) -> split -> LDS image S  (set and image of a k-tile have the same parity: kt & 1)
        auto stage = [&](auto s_, int kt) {
            constexpr int S = decltype(s_)::value;
            unsigned short* As = smem16 + S * Cfg::IMG;
            unsigned short* Bs = As + 3 * Cfg::A_PLANE;
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j) {
                const int u = t + NP * j;
                if (A_TOTAL % NP == 0 || u < A_TOTAL) {
                    if constexpr (P::A_RC) {
                        f32x4 v[KV / 4];
                        p.a_cvt(actx[j], araw[S][j][0], a_k(kt, j, 0), k_end, v);
                        unsigned short* dst = As + (u / AKQ) * BF3_BKP + (u % AKQ) * KV;
#pragma unroll
                        for (int q = 0; q < KV / 4; ++q) bf3_store4(v[q], dst + 4 * q, dst + Cfg::A_PLANE + 4 * q, dst + 2 * Cfg::A_PLANE + 4 * q);
                    } else {
                        f32x4 v[RA][KV / 4];
#pragma unroll
                        for (int r = 0; r < RA; ++r) p.a_cvt(actx[j], araw[S][j][r], a_k(kt, j, r), k_end, v[r]);
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
                const int u = t + NP * j;
                if (B_TOTAL % NP == 0 || u < B_TOTAL) {
                    if constexpr (P::B_RC) {
                        const f32x4 v = p.b_cvt(braw[S][j][0]);
                        unsigned short* dst = Bs + (u >> 3) * BF3_BKP + (u & 7) * 4;
                        f32x4 vs;
#pragma unroll
                        for (int e = 0; e < 4; ++e) vs[e] = __uint_as_float(__float_as_uint(v[e]) ^ sgn);
                        bf3_store4(vs, dst, dst + Cfg::B_PLANE, dst + 2 * Cfg::B_PLANE);
                    } else {
                        f32x4 v[RB];
#pragma unroll
                        for (int r = 0; r < RB; ++r) {
                            v[r] = p.b_cvt(braw[S][j][r]);
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

        // step kt: keys(kt + 3) -> key buffer kt % 3 (held keys(kt): last read by gather(kt), two barriers ago)
        //          gather(kt + 2) -> register set kt & 1 (emptied by stage(kt) in the previous step)
        //          stage(kt + 1)  -> image (kt + 1) & 1 (last read by the consumers in step kt - 1)
        if constexpr (Cfg::KSH) {
            for (int q = 0; q < 3 && q < ntk; ++q) make_keys(q, q);
            __syncthreads();  // barrier K
        }
        if (ntk > 0) {
            fetch(S0(), 0, 0);
            if (ntk > 1) fetch(S1(), 1, 1);
            stage(S0(), 0);
        }
        __syncthreads();  // barrier 0
        int k3 = 0;  // kt % 3
        auto step = [&](auto s_, int kt) {
            constexpr int S = decltype(s_)::value;
            using SO = std::integral_constant<int, 1 - S>;
            if (kt + 3 < ntk) make_keys(kt + 3, k3);
            if (kt + 2 < ntk) fetch(s_, kt + 2, k3 == 0 ? 2 : k3 - 1);  // key buffer (kt + 2) % 3
            if (kt + 1 < ntk) stage(SO(), kt + 1);
            k3 = (k3 == 2) ? 0 : k3 + 1;
        };
        // steady state: the gather is UNCONDITIONAL.  With `if (kt + 2 < ntk) gather` the compiler's s_waitcnt pass has to merge the
        // path without new loads into the path with them, and the split of k-tile kt + 1 then waits for vmcnt(7..0) -- i.e. for the
        // loads issued a few instructions earlier -- instead of vmcnt(15..8) (seen in the ISA of the first version)
        int kt = 0;
        for (; kt + 3 < ntk; kt += 2) {
            // sched_barrier: without it the scheduler hoists the round / split arithmetic of the registers just requested above the
            // s_barrier (register-only work), and the wave then waits for those loads BEFORE the barrier (seen in the ISA)
            if constexpr (Cfg::KSH) make_keys(kt + 3, k3);
            fetch(S0(), kt + 2, k3 == 0 ? 2 : k3 - 1);
            __builtin_amdgcn_sched_barrier(0);
            stage(S1(), kt + 1);
            k3 = (k3 == 2) ? 0 : k3 + 1;
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();  // barrier 2 kt + 1
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 4 < ntk) make_keys(kt + 4, k3);
            fetch(S1(), kt + 3, k3 == 0 ? 2 : k3 - 1);
            __builtin_amdgcn_sched_barrier(0);
            stage(S0(), kt + 2);
            k3 = (k3 == 2) ? 0 : k3 + 1;
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();  // barrier 2 kt + 2
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; kt < ntk; kt += 2) {  // the last one to three k-tiles
            step(S0(), kt);
            __syncthreads();
            if (kt + 1 < ntk) step(S1(), kt + 1);
            __syncthreads();
        }
        if constexpr (CS) {
            if (do_cs) {  // block-uniform: column c = the BKQ units of column quad c >> 2, summed in unit order through LDS
                float* red = reinterpret_cast<float*>(smem16);
#pragma unroll
                for (int j = 0; j < B_UNITS; ++j)
                    if (B_TOTAL % NP == 0 || t + NP * j < B_TOTAL) *reinterpret_cast<f32x4*>(red + (size_t)(t + NP * j) * 4) = cs[j];
                __syncthreads();  // barrier C
                if (t < BN && n0 + t < p.N) {
                    float s_ = 0.f;
                    for (int q = 0; q < BKQ; ++q) s_ += red[((t >> 2) * BKQ + q) * 4 + (t & 3)];
                    if (nsplit > 1)
                        partial[((size_t)kz * MP + p.M) * p.N + n0 + t] = s_;
                    else
                        p.store_colsum(n0 + t, s_);
                }
            }
        }
        return;
    }

    // ------------------------------------------------------------------ consumer waves
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;
    // all fragment reads of the k-tile are issued up front (BK / 16 k-groups x (TM + TN) x 3 planes); the first MFMAs wait for the
    // first k-group only (LDS returns in order), the reads of the later groups land underneath them
    auto mfma_tile = [&](auto s_) {
        constexpr int S = decltype(s_)::value;
        const unsigned short* As = smem16 + S * Cfg::IMG;
        const unsigned short* Bs = As + 3 * Cfg::A_PLANE;
        bf16x8 af[BK / 16][TM][3], bf[BK / 16][TN][3];
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned short* src = As + ((wm * TM + i) * 32 + li) * BF3_BKP + c * 16 + hi * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[c][i][pl] = *reinterpret_cast<const bf16x8*>(src + pl * Cfg::A_PLANE);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned short* src = Bs + ((wn * TN + j) * 32 + li) * BF3_BKP + c * 16 + hi * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[c][j][pl] = *reinterpret_cast<const bf16x8*>(src + pl * Cfg::B_PLANE);
            }
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // six partial products, smallest weight first
#pragma unroll
        for (int c = 0; c < BK / 16; ++c)
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if constexpr (EpiV4<P>::value)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[c][j][PB[q]], af[c][i][PA[q]], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c][i][PA[q]], bf[c][j][PB[q]], acc[i][j], 0, 0, 0);
    };

    if constexpr (Cfg::KSH) __syncthreads();  // barrier K
    __syncthreads();                           // barrier 0: image 0 holds k-tile 0
    int kt = 0;
    for (; kt + 3 < ntk; kt += 2) {  // same skeleton as the producers' loops
        mfma_tile(S0());
        __syncthreads();  // barrier 2 kt + 1
        mfma_tile(S1());
        __syncthreads();  // barrier 2 kt + 2
    }
    for (; kt < ntk; kt += 2) {
        mfma_tile(S0());
        __syncthreads();
        if (kt + 1 < ntk) mfma_tile(S1());
        __syncthreads();
    }
    const bool split = nsplit > 1;
    if constexpr (CS) {
        if (do_cs) __syncthreads();  // barrier C
    }
    // ------------------------------------------------------------------ epilogue (igemm_bf3_kernel's)
    if (flip_all) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = -acc[i][j][v];
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

template <class P, int TM, int TN, int WM, int WN, int PW>
inline int igemm_bf3_ws_launch(const P& p, float* ws, size_t ws_floats, int target_blocks, hipStream_t stream) {
    using Cfg = IgemmBf3WsCfg<P, TM, TN, WM, WN, PW>;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return HAB_ERR_ARG;
    const IgemmPlan pl = igemm_plan(Cfg::BM, Cfg::BN, p.M, p.N, p.K, target_blocks, 4096, ws ? ws_floats : 0);
    auto kern = igemm_bf3_ws_kernel<P, TM, TN, WM, WN, PW>;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = (Cfg::LDS_BYTES > 64 * 1024) ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) : hipSuccess;
    if (attr_err != hipSuccess) return (int)attr_err;
    const int ntiles = cdiv(p.M, Cfg::BM) * cdiv(p.N, Cfg::BN);
    const int grid = pl.splits >= 16 ? ntiles * ((pl.splits + 7) / 8 * 8) : ntiles * pl.splits;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, pl.k_per_split, ws, sign_schedule, pl.splits);
    HAB_LAUNCH_CHECK();
    if (pl.splits > 1) {
        igemm_splitk_reduce<P>(p, ws, pl.splits, stream);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

}  // namespace hab
