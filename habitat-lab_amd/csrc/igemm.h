// igemm.h -- the one tiled fp32-MFMA contraction kernel of the library (gfx950).
//
//   C[i][j] = sum_r P(i, r) * Q(r, j)          i < M, j < N, r < K
//
// Every dense contraction on the PPO hot path is an instance: conv fwd / dgrad / wgrad as implicit
// GEMMs over NHWC activations, Linear fwd / dgrad / wgrad, RNN input projections.  A problem type
// `Prob` supplies the operand gathers and the epilogue; the tile machinery (LDS staging, MFMA
// fragments, split-K) is shared.
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
// Staging is global -> registers -> LDS with the loads of tile t+1 issued before the MFMAs of
// tile t (one barrier per K-tile).  fp32 MFMA is slow enough (1/16 of bf16) that one wave per
// SIMD saturates it; VALU index math of the gathers hides under the 64-cycle MFMAs.
//
// Block -> tile mapping is XCD-aware: blocks are dispatched round-robin over the 8 XCDs, so tile
// ids are remapped to give each XCD (private 4 MiB L2) a contiguous run of M-tiles, which keeps the
// im2col halo re-reads of neighbouring tiles in one L2.
#pragma once
#include <type_traits>

#include "hab_common.h"

namespace hab {

template <class P>
struct IgemmLaunch;  // fwd

constexpr int IGEMM_BK = 32;

// Optional fused column sums of the B operand (bias gradients ride along with the weight-gradient
// contraction: db[j] = sum_r Q(r, j)).  A problem opts in with `static constexpr bool COLSUM_B = true`,
// a nullable `float* colsum` member and `store_colsum(j, v)`; B must be in the j-contiguous form.
template <class P, class = void>
struct ColsumB : std::false_type {};
template <class P>
struct ColsumB<P, std::void_t<decltype(P::COLSUM_B)>> : std::bool_constant<P::COLSUM_B && !P::B_RC> {};

template <class P, int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64) igemm_kernel(const P p, const int k_per_split, float* __restrict__ partial) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = IGEMM_BK;
    constexpr int LDK = BK + 4;
    constexpr int A_LD = P::A_RC ? LDK : (BM + 4);
    constexpr int B_LD = P::B_RC ? LDK : (BN + 4);
    constexpr int A_TILE = P::A_RC ? BM * LDK : BK * (BM + 4);
    constexpr int B_TILE = P::B_RC ? BN * LDK : BK * (BN + 4);
    constexpr int A_UNITS = BM * BK / 4 / NT;
    constexpr int B_UNITS = (BN * BK / 4 + NT - 1) / NT;
    constexpr bool B_PARTIAL = (BN * BK / 4) < NT;  // fewer units than threads
    static_assert(A_UNITS >= 1, "tile too small for the block");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;               // [2][A_TILE]
    float* Bs = smem + 2 * A_TILE;  // [2][B_TILE]

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

    // ---- staging contexts ----
    // RC: unit u = t + NT*j -> row = u / 8, kq = u % 8 (fixed per thread).  IC: r = u / (rows/4), i4 fixed.
    typename P::ACtx actx[A_UNITS];
    typename P::BCtx bctx[B_UNITS];
    if constexpr (P::A_RC) {
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) actx[j] = p.a_ctx(m0 + (t >> 3) + (NT >> 3) * j);
    } else {
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) actx[j] = p.a_ctx(m0 + (t % (BM / 4)) * 4);
    }
    if constexpr (P::B_RC) {
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) bctx[j] = p.b_ctx(n0 + (t >> 3) + (NT >> 3) * j);
    } else {
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) bctx[j] = p.b_ctx(n0 + (t % (BN / 4)) * 4);
    }
    const bool b_active = !B_PARTIAL || (t < BN * BK / 4);
    constexpr bool CS = ColsumB<P>::value;
    bool has_cs = false;
    if constexpr (CS) has_cs = (p.colsum != nullptr);
    const bool do_cs = has_cs && (tile_m == 0);  // only the first row of tiles accumulates the column sums
    const int MP = p.M + (has_cs ? 1 : 0);       // rows of a split-K slab (the extra row carries the column sums)
    f32x4 cs;
    cs[0] = 0.f; cs[1] = 0.f; cs[2] = 0.f; cs[3] = 0.f;

    f32x4 areg[A_UNITS], breg[B_UNITS];
    auto load_tile = [&](int kt) {
        const int k0 = k_begin + kt * BK;
        if constexpr (P::A_RC) {
            const int k = k0 + (t & 7) * 4;
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j) areg[j] = p.a_load(actx[j], k, k_end);
        } else {
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j) areg[j] = p.a_load(actx[j], k0 + t / (BM / 4) + (NT / (BM / 4)) * j, k_end);
        }
        if (b_active) {
            if constexpr (P::B_RC) {
                const int k = k0 + (t & 7) * 4;
#pragma unroll
                for (int j = 0; j < B_UNITS; ++j) breg[j] = p.b_load(bctx[j], k, k_end);
            } else {
#pragma unroll
                for (int j = 0; j < B_UNITS; ++j) breg[j] = p.b_load(bctx[j], k0 + t / (BN / 4) + (NT / (BN / 4)) * j, k_end);
                if constexpr (CS) {
                    if (do_cs) {
#pragma unroll
                        for (int j = 0; j < B_UNITS; ++j) cs += breg[j];
                    }
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * A_TILE;
        float* b = Bs + buf * B_TILE;
        if constexpr (P::A_RC) {
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j)
                *reinterpret_cast<f32x4*>(a + ((t >> 3) + (NT >> 3) * j) * LDK + (t & 7) * 4) = areg[j];
        } else {
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j)
                *reinterpret_cast<f32x4*>(a + (t / (BM / 4) + (NT / (BM / 4)) * j) * (BM + 4) + (t % (BM / 4)) * 4) = areg[j];
        }
        if (b_active) {
            if constexpr (P::B_RC) {
#pragma unroll
                for (int j = 0; j < B_UNITS; ++j)
                    *reinterpret_cast<f32x4*>(b + ((t >> 3) + (NT >> 3) * j) * LDK + (t & 7) * 4) = breg[j];
            } else {
#pragma unroll
                for (int j = 0; j < B_UNITS; ++j)
                    *reinterpret_cast<f32x4*>(b + (t / (BN / 4) + (NT / (BN / 4)) * j) * (BN + 4) + (t % (BN / 4)) * 4) = breg[j];
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

    if (ntk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

    for (int kt = 0; kt < ntk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntk) load_tile(kt + 1);
        const float* a = As + buf * A_TILE;
        const float* b = Bs + buf * B_TILE;
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + li;
                if constexpr (P::A_RC) {
                    af[i] = *reinterpret_cast<const f32x4*>(a + row * LDK + c * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[i][s] = a[(c * 8 + hi * 4 + s) * (BM + 4) + row];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = (wn * TN + j) * 32 + li;
                if constexpr (P::B_RC) {
                    bf[j] = *reinterpret_cast<const f32x4*>(b + col * LDK + c * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[j][s] = b[(c * 8 + hi * 4 + s) * (BN + 4) + col];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ntk) store_tile(buf ^ 1);
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
                constexpr int GR = BN / 4;  // threads t, t+GR, t+2GR, ... hold the same 4 columns
                float s = 0.f;
                for (int q = (t >> 2); q < NT; q += GR) s += red[q * 4 + (t & 3)];
                if (split)
                    partial[((size_t)kz * MP + p.M) * p.N + n0 + t] = s;
                else
                    p.store_colsum(n0 + t, s);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
                if (row < p.M && col < p.N) {
                    if (split)
                        partial[((size_t)kz * MP + row) * p.N + col] = acc[i][j][v];
                    else
                        p.store(row, col, acc[i][j][v]);
                }
            }
        }
}

// Second pass of a split-K launch: fixed-order sum of the partial slabs, then the problem's epilogue.
template <class P>
__global__ void __launch_bounds__(256) igemm_splitk_reduce_kernel(const P p, const float* __restrict__ partial, int splits) {
    int MP = p.M;
    if constexpr (ColsumB<P>::value) MP += (p.colsum != nullptr) ? 1 : 0;
    const size_t total = (size_t)MP * p.N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        float s = 0.0f;
        for (int z = 0; z < splits; ++z) s += partial[(size_t)z * total + e];
        const int m = (int)(e / p.N), n = (int)(e % p.N);
        if constexpr (ColsumB<P>::value) {
            if (m == p.M) { p.store_colsum(n, s); continue; }
        }
        p.store(m, n, s);
    }
}

struct IgemmPlan {
    int splits = 1;
    int k_per_split = 0;
    size_t partial_floats = 0;  // workspace needed when splits > 1
};

// Chooses a split-K factor so that the launch has roughly >= `target_blocks` workgroups.
template <int BM, int BN>
inline IgemmPlan igemm_plan(int M, int N, int K, int target_blocks, int max_splits, size_t max_partial_floats) {
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
inline int igemm_launch(const P& p, const IgemmPlan& pl, float* partial, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = IGEMM_BK;
    constexpr int A_TILE = P::A_RC ? BM * (BK + 4) : BK * (BM + 4);
    constexpr int B_TILE = P::B_RC ? BN * (BK + 4) : BK * (BN + 4);
    constexpr size_t LDS = (size_t)2 * (A_TILE + B_TILE) * sizeof(float);
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return HAB_ERR_ARG;
    if (pl.splits > 1 && !partial) return HAB_ERR_ARG;
    auto kern = igemm_kernel<P, TM, TN, WM, WN>;
    static bool attr_set = false;
    if (!attr_set && LDS > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), 1, pl.splits);
    kern<<<grid, WM * WN * 64, LDS, stream>>>(p, pl.k_per_split, partial);
    HAB_LAUNCH_CHECK();
    if (pl.splits > 1) {
        int blocks = (int)cdivl((long long)(p.M + 1) * p.N, 256);
        if (blocks > 4096) blocks = 4096;
        igemm_splitk_reduce_kernel<P><<<blocks, 256, 0, stream>>>(p, partial, pl.splits);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

}  // namespace hab
