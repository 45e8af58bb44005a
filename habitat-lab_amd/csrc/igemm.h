// igemm.h -- the one tiled fp32-MFMA contraction kernel of the library (gfx950).
//
//   C[i][j] = sum_r P(i, r) * Q(r, j)          i < M, j < N, r < K
//
// Every dense contraction on the PPO hot path is an instance: conv fwd / dgrad / wgrad as implicit
// GEMMs over NHWC activations, Linear fwd / dgrad / wgrad, RNN input projections.  A problem type
// `Prob` (problems.h) supplies the operand gathers and the epilogue; the tile machinery (LDS
// staging, MFMA fragments, split-K) is shared.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD, 157 TFLOP/s chip peak).  Lane l of a
// wave supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; D: col = l&31,
// row = (v&3) + 8*(v>>2) + 4*(l>>5) for accumulator register v (cdna_hip_programming.md section 3).
// The k order inside a K-tile is free, so lane half `hi` consumes k = 8c + 4hi + s at MFMA step
// (c, s): with an r-contiguous LDS image that is ONE ds_read_b128 per operand per 4 MFMAs.
//
// Operand forms (per problem, compile time):
//   *_RC = true : operand is r-contiguous in memory  -> LDS image [rows][BK+4], ds_read_b128
//   *_RC = false: operand is i/j-contiguous in memory -> LDS image [BK][rows+4], ds_read_b32
// A gather "unit" is KV consecutive elements along the contiguous direction (KV = 4: one 16-byte
// load; the observation-ingest problems use KV = 16: four filter taps from 12 bytes of uint8 rgb +
// 16 bytes of depth).
//
// Pipeline.  The tile is SINGLE-buffered in LDS (41 KB for the 256x32 tile) so that 3-5 workgroups
// are resident per CU: while one workgroup sits in its barriers / address math / epilogue, the MFMA
// pipe of each SIMD is fed by the waves of the others (fp32 MFMA is 64 cycles per instruction, one
// ready wave per SIMD saturates it).  Global loads are split into `fetch` (branch-free: clamped
// address + validity bit, issued right after the barrier, in flight across the MFMAs of the current
// tile) and `cvt` (zero-fill / uint8 scaling, applied when the registers are written to LDS).
// The epilogue pre-loads per-column data (bias) once and batches the optional per-element loads
// (residual add, ReLU mask, accumulate) four rows at a time in front of the stores.
//
// Block -> tile mapping is XCD-aware: blocks are dispatched round-robin over the 8 XCDs, so tile
// ids are remapped to give each XCD (private 4 MiB L2) a contiguous run of M-tiles, which keeps the
// im2col halo re-reads of neighbouring tiles in one L2.
#pragma once
#include <type_traits>

#include "hab_common.h"

namespace hab {

constexpr int IGEMM_BK = 32;
// Row pitch padding of the i/j-contiguous LDS image [BK][rows + pad].  (pad 8 would put the two lane halves of a fragment read,
// which are 4 k rows apart, on disjoint banks; measured: no difference end to end -- the ds_read_b32 stream is issue-bound, not
// bank-bound -- so the smaller image stays.)
constexpr int IGEMM_ICPAD = 4;

// Optional fused column sums of the B operand (bias gradients ride along with the weight-gradient
// contraction: db[j] = sum_r Q(r, j)).  A problem opts in with `static constexpr bool COLSUM_B = true`,
// a nullable `float* colsum` member and `store_colsum(j, v)`; B must be in the j-contiguous form.
template <class P, class = void>
struct ColsumB : std::false_type {};
template <class P>
struct ColsumB<P, std::void_t<decltype(P::COLSUM_B)>> : std::bool_constant<P::COLSUM_B && !P::B_RC> {};

// Gather-unit width along the contiguous direction of the A operand (default 4 floats).
template <class P, class = void>
struct AKv : std::integral_constant<int, 4> {};
template <class P>
struct AKv<P, std::void_t<decltype(P::A_KV)>> : std::integral_constant<int, P::A_KV> {};

// Vector epilogue.  A problem whose output is n-contiguous (NHWC activations / activation gradients) opts in with
// `static constexpr bool EPI_VEC4 = true` and epi_col4 / epi_fetch4 / epi_store4.  The kernels then issue the MFMAs with the
// operands swapped, which transposes the accumulator: lane l holds output row m = l & 31 and the FOUR CONSECUTIVE columns
// n = 8*(v >> 2) + 4*(l >> 5) + (v & 3) per register quad, so the epilogue is one 16-byte load per optional operand and one
// 16-byte store per quad (4 per 32x32 tile and lane instead of 16 scalar ones) and one row decode per tile instead of 16.
// fp32 MFMA shares the issue port with everything else on the SIMD, so epilogue instructions are MFMA time lost.
template <class P, class = void>
struct EpiV4 : std::false_type {};
template <class P>
struct EpiV4<P, std::void_t<decltype(P::EPI_VEC4)>> : std::bool_constant<P::EPI_VEC4> {};

template <class P, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue_v4(const P& p, const f32x16 (&acc)[TM][TN], int mbase, int nbase, int li, int hi) {
    typename P::EpiCol4 ecol[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) ecol[j][g] = p.epi_col4(nbase + j * 32 + 8 * g + 4 * hi);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const typename P::EpiRow erow = p.epi_row(mbase + i * 32 + li);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            typename P::EpiAux4 aux[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) aux[g] = p.epi_fetch4(erow, ecol[j][g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v[0] = acc[i][j][4 * g]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
                p.epi_store4(erow, ecol[j][g], aux[g], v);
            }
        }
    }
}

// split-K slab write of the transposed accumulator
template <class P, int TM, int TN>
__device__ __forceinline__ void igemm_partial_v4(const P& p, const f32x16 (&acc)[TM][TN], float* __restrict__ slab, int mbase, int nbase,
                                                 int li, int hi) {
    const bool vec = ((p.N & 3) == 0) && ((reinterpret_cast<uintptr_t>(slab) & 15) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = mbase + i * 32 + li;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = nbase + j * 32 + 8 * g + 4 * hi;
                float* dst = slab + (size_t)row * p.N + col;
                if (vec && col + 3 < p.N) {
                    f32x4 v;
                    v[0] = acc[i][j][4 * g]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < p.N) dst[e] = acc[i][j][4 * g + e];
                }
            }
    }
}

template <class P, int TM, int TN, int WM, int WN>
struct IgemmCfg {
    static constexpr int NT = WM * WN * 64;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = IGEMM_BK;
    static constexpr int LDK = BK + 4;
    static constexpr int KV = AKv<P>::value;
    static constexpr int A_TILE = P::A_RC ? BM * LDK : BK * (BM + IGEMM_ICPAD);
    static constexpr int B_TILE = P::B_RC ? BN * LDK : BK * (BN + IGEMM_ICPAD);
    static constexpr int A_TOTAL = BM * BK / KV, B_TOTAL = BN * BK / 4;  // gather units per tile
    static constexpr int A_UNITS = (A_TOTAL + NT - 1) / NT, B_UNITS = (B_TOTAL + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = (size_t)(A_TILE + B_TILE) * sizeof(float);
    // unit -> (row, k) decomposition must be thread-invariant in the fixed coordinate
    static_assert(P::A_RC ? (NT % (BK / KV) == 0) : (NT % (BM / KV) == 0), "A unit mapping");
    static_assert(P::B_RC ? (NT % (BK / 4) == 0) : (NT % (BN / 4) == 0), "B unit mapping");
    static_assert(NT * 4 <= A_TILE + B_TILE, "column-sum fold needs NT*4 floats of LDS");
};

template <class P, int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64) igemm_kernel(const P p, const int k_per_split, float* __restrict__ partial) {
    using Cfg = IgemmCfg<P, TM, TN, WM, WN>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LDK = Cfg::LDK, KV = Cfg::KV;
    constexpr int A_TILE = Cfg::A_TILE;
    constexpr int A_UNITS = Cfg::A_UNITS, B_UNITS = Cfg::B_UNITS, A_TOTAL = Cfg::A_TOTAL, B_TOTAL = Cfg::B_TOTAL;
    constexpr int AKQ = BK / KV;   // A units per row            (RC form)
    constexpr int AIQ = BM / KV;   // A units per k-row          (IC form)
    constexpr int BJQ = BN / 4;    // B units per k-row          (IC form)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + A_TILE;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

    // ---- XCD-aware tile remap (bijective for any tile count) ----
    const int nt_m = cdiv(p.M, BM), nt_n = cdiv(p.N, BN);
    const int ntiles = nt_m * nt_n;
    int tile;
    {
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = tile % nt_n, tile_m = tile / nt_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kz = blockIdx.z;
    const int k_begin = kz * k_per_split;
    const int k_end = min(p.K, k_begin + k_per_split);
    const int ntk = max(0, cdiv(k_end - k_begin, BK));

    // ---- staging contexts: unit u = t + NT*j ----
    //   RC: row = u / AKQ, k-offset = (u % AKQ) * KV   (fixed per thread)
    //   IC: k-row = u / AIQ, i-offset = (u % AIQ) * KV (fixed per thread)
    typename P::ACtx actx[A_UNITS];
    typename P::BCtx bctx[B_UNITS];
#pragma unroll
    for (int j = 0; j < A_UNITS; ++j) {
        const int u = t + NT * j;
        actx[j] = P::A_RC ? p.a_ctx(m0 + u / AKQ) : p.a_ctx(m0 + (u % AIQ) * KV);
    }
#pragma unroll
    for (int j = 0; j < B_UNITS; ++j) {
        const int u = t + NT * j;
        bctx[j] = P::B_RC ? p.b_ctx(n0 + (u >> 3)) : p.b_ctx(n0 + (u % BJQ) * 4);
    }
    constexpr bool CS = ColsumB<P>::value;
    bool has_cs = false;
    if constexpr (CS) has_cs = (p.colsum != nullptr);
    const bool do_cs = has_cs && (tile_m == 0);  // only the first row of tiles accumulates the column sums
    const int MP = p.M + (has_cs ? 1 : 0);       // rows of a split-K slab (the extra row carries the column sums)
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};

    typename P::ARaw araw[A_UNITS];
    typename P::BRaw braw[B_UNITS];
    auto a_k = [&](int kt, int j) {  // k coordinate of A unit j in K-tile kt
        const int u = t + NT * j;
        return k_begin + kt * BK + (P::A_RC ? (u % AKQ) * KV : u / AIQ);
    };
    auto b_k = [&](int kt, int j) {
        const int u = t + NT * j;
        return k_begin + kt * BK + (P::B_RC ? (u & 7) * 4 : u / BJQ);
    };
    auto fetch = [&](int kt) {
        const typename P::KCtx kc = p.k_ctx(k_begin + kt * BK, k_end);  // block-uniform
        // r-contiguous form: every unit of a thread has the same k -> one key (tap decode) per thread and K-tile
        typename P::AKey ak = p.a_key(kc, a_k(kt, 0), k_end);
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j)
            if (A_TOTAL % NT == 0 || t + NT * j < A_TOTAL) {
                if (!P::A_RC && j > 0) ak = p.a_key(kc, a_k(kt, j), k_end);
                araw[j] = p.a_fetch(actx[j], kc, ak);
            }
        typename P::BKey bk = p.b_key(kc, b_k(kt, 0), k_end);
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j)
            if (B_TOTAL % NT == 0 || t + NT * j < B_TOTAL) {
                if (!P::B_RC && j > 0) bk = p.b_key(kc, b_k(kt, j), k_end);
                braw[j] = p.b_fetch(bctx[j], kc, bk);
            }
    };
    auto stage = [&](int kt) {  // registers -> LDS (zero fill / conversion applied here)
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            const int u = t + NT * j;
            if (A_TOTAL % NT == 0 || u < A_TOTAL) {
                f32x4 v[KV / 4];
                p.a_cvt(actx[j], araw[j], a_k(kt, j), k_end, v);
                float* dst = P::A_RC ? As + (u / AKQ) * LDK + (u % AKQ) * KV : As + (u / AIQ) * (BM + IGEMM_ICPAD) + (u % AIQ) * KV;
#pragma unroll
                for (int q = 0; q < KV / 4; ++q) *reinterpret_cast<f32x4*>(dst + 4 * q) = v[q];
            }
        }
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) {
            const int u = t + NT * j;
            if (B_TOTAL % NT == 0 || u < B_TOTAL) {
                const f32x4 v = p.b_cvt(braw[j]);
                float* dst = P::B_RC ? Bs + (u >> 3) * LDK + (u & 7) * 4 : Bs + (u / BJQ) * (BN + IGEMM_ICPAD) + (u % BJQ) * 4;
                *reinterpret_cast<f32x4*>(dst) = v;
                if constexpr (CS) {
                    if (do_cs) cs += v;
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

    if (ntk > 0) fetch(0);
    for (int kt = 0; kt < ntk; ++kt) {
        stage(kt);
        __syncthreads();
        if (kt + 1 < ntk) fetch(kt + 1);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + li;
                if constexpr (P::A_RC) {
                    af[i] = *reinterpret_cast<const f32x4*>(As + row * LDK + c * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[i][s] = As[(c * 8 + hi * 4 + s) * (BM + IGEMM_ICPAD) + row];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = (wn * TN + j) * 32 + li;
                if constexpr (P::B_RC) {
                    bf[j] = *reinterpret_cast<const f32x4*>(Bs + col * LDK + c * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[j][s] = Bs[(c * 8 + hi * 4 + s) * (BN + IGEMM_ICPAD) + col];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if constexpr (EpiV4<P>::value)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ----
    const bool split = gridDim.z > 1;
    if constexpr (CS) {
        if (do_cs) {  // block-uniform: fold the per-thread partial column sums through LDS (the tile buffers are free now)
            float* red = smem;
            *reinterpret_cast<f32x4*>(red + t * 4) = cs;
            __syncthreads();
            if (t < BN && n0 + t < p.N) {
                // thread q holds the columns (q % BJQ)*4 .. +3  (unit mapping of the IC form)
                float s = 0.f;
                for (int q = (t >> 2); q < NT; q += BJQ) s += red[q * 4 + (t & 3)];
                if (split)
                    partial[((size_t)kz * MP + p.M) * p.N + n0 + t] = s;
                else
                    p.store_colsum(n0 + t, s);
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
        for (int g = 0; g < 4; ++g) {  // four rows at a time: optional loads first, then the stores
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

// Second pass of a split-K launch: fixed-order sum of the partial slabs, then the problem's epilogue.
// A workgroup owns 64 consecutive output elements; its four waves each sum a contiguous quarter of the slabs (loads issued eight
// at a time, added in slab order) and the quarters are combined in a fixed order through LDS.  Weight gradients have few output
// elements and hundreds of slabs (M*N = 8k .. 600k, K = 10^6 .. 10^7): one thread per element walking all slabs serially made
// this pass latency-bound at 30-40 workgroups.
template <class P, bool WIDE>
__global__ void __launch_bounds__(256) igemm_splitk_reduce_kernel(const P p, const float* __restrict__ partial, int splits) {
    __shared__ float red[4][64];
    int MP = p.M;
    if constexpr (ColsumB<P>::value) MP += (p.colsum != nullptr) ? 1 : 0;
    const size_t total = (size_t)MP * p.N;
    if constexpr (!WIDE) {  // few slabs, many elements (rollout-time forward convolutions): one thread per element
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            float s = 0.0f;
            for (int z = 0; z < splits; ++z) s += partial[(size_t)z * total + e];
            const int m = (int)(e / p.N), n = (int)(e % p.N);
            if constexpr (ColsumB<P>::value) {
                if (m == p.M) { p.store_colsum(n, s); continue; }
            }
            p.store(m, n, s);
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int chunk = (splits + 3) >> 2;
    const int z0 = ty * chunk, z1 = min(splits, z0 + chunk);
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < total; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + tx;
        float s = 0.0f;
        if (e < total) {
            const float* src = partial + e;
            int z = z0;
            for (; z + 8 <= z1; z += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(z + u) * total];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; z < z1; ++z) s += src[(size_t)z * total];
        }
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && e < total) {
            s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
            const int m = (int)(e / p.N), n = (int)(e % p.N);
            bool done = false;
            if constexpr (ColsumB<P>::value) {
                if (m == p.M) { p.store_colsum(n, s); done = true; }
            }
            if (!done) p.store(m, n, s);
        }
        __syncthreads();
    }
}

template <class P>
inline void igemm_splitk_reduce(const P& p, const float* ws, int splits, hipStream_t stream) {
    const long long total = (long long)(p.M + 1) * p.N;
    if (splits >= 16) {
        int blocks = (int)cdivl(total, 64);
        if (blocks > 8192) blocks = 8192;
        igemm_splitk_reduce_kernel<P, true><<<blocks, 256, 0, stream>>>(p, ws, splits);
    } else {
        int blocks = (int)cdivl(total, 256);
        if (blocks > 4096) blocks = 4096;
        igemm_splitk_reduce_kernel<P, false><<<blocks, 256, 0, stream>>>(p, ws, splits);
    }
}

struct IgemmPlan {
    int splits = 1;
    int k_per_split = 0;
    size_t partial_floats = 0;  // workspace needed when splits > 1
};

// Chooses a split-K factor so that the launch has roughly >= `target_blocks` workgroups.
inline IgemmPlan igemm_plan(int BM, int BN, int M, int N, int K, int target_blocks, int max_splits, size_t max_partial_floats) {
    IgemmPlan pl;
    const long long tiles = (long long)cdiv(M, BM) * cdiv(N, BN);
    int splits = 1;
    if (tiles < target_blocks) splits = (int)((target_blocks + tiles - 1) / tiles);
    const int ktiles = cdiv(K, IGEMM_BK);
    if (splits > ktiles) splits = ktiles;
    if (splits > max_splits) splits = max_splits;
    while (splits > 1 && (size_t)splits * (size_t)(M + 1) * (size_t)N > max_partial_floats) --splits;
    if (splits < 1) splits = 1;
    pl.k_per_split = cdiv(ktiles, splits) * IGEMM_BK;
    pl.splits = cdiv(K, pl.k_per_split);
    pl.partial_floats = pl.splits > 1 ? (size_t)pl.splits * (M + 1) * N : 0;
    return pl;
}

template <class P, int TM, int TN, int WM, int WN>
inline int igemm_launch(const P& p, float* ws, size_t ws_floats, int target_blocks, hipStream_t stream) {
    using Cfg = IgemmCfg<P, TM, TN, WM, WN>;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return HAB_ERR_ARG;
    const IgemmPlan pl = igemm_plan(Cfg::BM, Cfg::BN, p.M, p.N, p.K, target_blocks, 4096, ws ? ws_floats : 0);
    auto kern = igemm_kernel<P, TM, TN, WM, WN>;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = (Cfg::LDS_BYTES > 64 * 1024) ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) : hipSuccess;
    if (attr_err != hipSuccess) return (int)attr_err;
    dim3 grid(cdiv(p.M, Cfg::BM) * cdiv(p.N, Cfg::BN), 1, pl.splits);
    kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, pl.k_per_split, ws);
    HAB_LAUNCH_CHECK();
    if (pl.splits > 1) {
        igemm_splitk_reduce<P>(p, ws, pl.splits, stream);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

}  // namespace hab
