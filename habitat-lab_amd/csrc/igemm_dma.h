// igemm_dma.h -- LDS-DMA staged variant of the fp32 MFMA contraction for r-contiguous conv problems (forward and data
// gradient with C, Cout multiples of 32).
//
// Why: on gfx950 an fp32 MFMA shares the SIMD's issue with every other instruction (tools/ubench/mfma_valu_overlap.hip),
// so each VALU / LDS-write instruction of the staging path is time taken from the MFMA pipe.  igemm_kernel stages
// global -> VGPR -> (select) -> ds_write_b128 with ~14 address instructions per 16-byte gather.  Here the gathers are
// `buffer_load_dwordx4 ... lds`: the data goes straight from L2 into LDS, no VGPR round trip, no ds_write, no select;
// zero padding comes from the buffer range check (invalid gathers get an out-of-range offset and the hardware writes 0).
// Per gather and K-tile the address work is: test one bit of a precomputed tap-validity mask, select the offset.  The filter
// tap of a K-tile is block-uniform (C % 32 == 0), so its byte offset travels in the instruction's scalar offset.
//
// LDS image: a DMA instruction writes lane l's 16 bytes at base + 16*l, i.e. 1 KB "chunks" = 8 rows x 8 k-quads, unpadded.
// Bank conflicts of the MFMA fragment reads (ds_read_b128, 16-lane groups {0-3,12-15,20-27} ...) are avoided by an XOR
// swizzle of the quad slot with g(row) = ((row >> 1) & 3) | (((row >> 4) & 1) << 2): within every 16-lane group the eight
// rows of either parity get eight distinct slots.
#pragma once
#include "igemm.h"
#include "problems.h"

namespace hab {

constexpr uint32_t DMA_OOB = 0x80000000u;  // voffset that fails the range check (records <= 0x7fffffff)

__device__ inline int dma_swz(int row) { return ((row >> 1) & 3) | (((row >> 4) & 1) << 2); }

template <class P, int TM, int TN, int WM, int WN, bool DB>
struct IgemmDmaCfg {
    static constexpr int NT = WM * WN * 64, NW = WM * WN;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = IGEMM_BK;
    static constexpr int A_CHUNKS = BM / 8, B_CHUNKS = BN / 8;  // 1 KB chunks, one DMA wave-instruction each
    static constexpr int A_UNITS = (A_CHUNKS + NW - 1) / NW, B_UNITS = (B_CHUNKS + NW - 1) / NW;
    static constexpr int TILE = (BM + BN) * BK;  // floats per buffer
    static constexpr size_t LDS_BYTES = (size_t)(DB ? 2 : 1) * TILE * sizeof(float);
};

template <class P, int TM, int TN, int WM, int WN, bool DB>
__global__ void __launch_bounds__(WM* WN * 64) igemm_dma_kernel(const P p, const int k_per_split, float* __restrict__ partial) {
    using Cfg = IgemmDmaCfg<P, TM, TN, WM, WN, DB>;
    constexpr int NW = Cfg::NW, BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int A_UNITS = Cfg::A_UNITS, B_UNITS = Cfg::B_UNITS, A_CHUNKS = Cfg::A_CHUNKS, B_CHUNKS = Cfg::B_CHUNKS;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    constexpr int TILE = Cfg::TILE;
    float* As = smem;              // [2][A_CHUNKS][64 lanes][4]   (buffer b at + b * TILE)
    float* Bs = smem + BM * BK;    // [2][B_CHUNKS][64][4]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

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

    // ---- DMA descriptors and per-gather offsets ----
    const DmaTile ta = p.dma_a_tile(m0), tb = p.dma_b_tile(n0);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ta.base), 0, (int)ta.records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tb.base), 0, (int)tb.records, 0x00020000);
    // unit j of wave w fills chunk j*NW + w; lane l -> row 8*chunk + (l >> 3), slot l & 7 holding k-quad (slot ^ swz(row))
    uint32_t avoff[A_UNITS], amask[A_UNITS], bvoff[B_UNITS];
#pragma unroll
    for (int j = 0; j < A_UNITS; ++j) {
        const int row = 8 * (j * NW + wave) + (lane >> 3);
        const int kq = (lane & 7) ^ dma_swz(row & 31);
        avoff[j] = p.dma_a_row(ta, m0 + row, amask[j]) + (uint32_t)kq * 16u;
        if (A_CHUNKS % NW != 0 && j * NW + wave >= A_CHUNKS) amask[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < B_UNITS; ++j) {
        const int row = 8 * (j * NW + wave) + (lane >> 3);
        const int kq = (lane & 7) ^ dma_swz(row & 31);
        uint32_t ok;
        bvoff[j] = p.dma_b_row(tb, n0 + row, ok) + (uint32_t)kq * 16u;
        if (!ok || (B_CHUNKS % NW != 0 && j * NW + wave >= B_CHUNKS)) bvoff[j] = DMA_OOB;
    }
    // fragment read offsets (floats) -- the image layout does not depend on the K-tile
    int aoff[TM][BK / 8], boff[TN][BK / 8];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const int row = (wm * TM + i) * 32 + li;
            aoff[i][c] = (row >> 3) * 256 + ((row & 7) * 8 + ((2 * c + hi) ^ dma_swz(li))) * 4;
        }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const int row = (wn * TN + j) * 32 + li;
            boff[j][c] = (row >> 3) * 256 + ((row & 7) * 8 + ((2 * c + hi) ^ dma_swz(li))) * 4;
        }

    auto issue = [&](int kt, int buf) {
        const int k0 = k_begin + kt * BK;
        int tap;
        uint32_t sa, sb;
        p.dma_tap(k0, tap, sa, sb);  // block-uniform: tap index, scalar byte offsets of the tap for A and B
        const uint32_t bit = 1u << tap;
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            if (A_CHUNKS % NW != 0 && j * NW + wave >= A_CHUNKS) continue;
            const uint32_t v = (amask[j] & bit) ? avoff[j] : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(As + buf * TILE + (j * NW + wave) * 256), 16, (int)v,
                                                     (int)sa, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) {
            if (B_CHUNKS % NW != 0 && j * NW + wave >= B_CHUNKS) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(Bs + buf * TILE + (j * NW + wave) * 256), 16,
                                                     (int)bvoff[j], (int)sb, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    // Double-buffered form (DB): the DMA of tile kt+1 is issued right after the barrier that publishes tile kt and lands while the MFMAs
    // of tile kt run; that barrier also orders the DMA behind every wave's reads of the buffer it overwrites (tile kt-1).
    auto compute = [&](const float* a, const float* b) {
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(a + aoff[i][c]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b + boff[j][c]);
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
    };
    if constexpr (!DB) {
        // Single buffer (N <= 64: 20-36 KB per workgroup, 4+ workgroups per CU hide the DMA latency better than a second buffer,
        // measured 96 vs 81 TFLOP/s on the 3x3 32->32 layer).
        for (int kt = 0; kt < ntk; ++kt) {
            issue(kt, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's DMA writes have landed
            __syncthreads();
            compute(As, Bs);
            __syncthreads();
        }
    }
    if (DB && ntk > 0) issue(0, 0);
    for (int kt = 0; DB && kt < ntk; kt += 2) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's DMA writes of tile kt have landed
        __syncthreads();
        if (kt + 1 < ntk) issue(kt + 1, 1);
        compute(As, Bs);
        if (kt + 1 < ntk) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (kt + 2 < ntk) issue(kt + 2, 0);
            compute(As + TILE, Bs + TILE);
        }
    }

    // ---- epilogue (as igemm_kernel) ----
    if constexpr (EpiV4<P>::value) {
        if (gridDim.z > 1)
            igemm_partial_v4<P, TM, TN>(p, acc, partial + (size_t)kz * p.M * p.N, m0 + wm * TM * 32, n0 + wn * TN * 32, li, hi);
        else
            igemm_epilogue_v4<P, TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, li, hi);
        return;
    }
    if (gridDim.z > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = m0 + (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
                    if (row < p.M && col < p.N) partial[((size_t)kz * p.M + row) * p.N + col] = acc[i][j][v];
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

template <class P, int TM, int TN, int WM, int WN, bool DB>
inline int igemm_dma_launch(const P& p, float* ws, size_t ws_floats, int target_blocks, hipStream_t stream) {
    using Cfg = IgemmDmaCfg<P, TM, TN, WM, WN, DB>;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return HAB_ERR_ARG;
    const IgemmPlan pl = igemm_plan(Cfg::BM, Cfg::BN, p.M, p.N, p.K, target_blocks, 4096, ws ? ws_floats : 0);
    auto kern = igemm_dma_kernel<P, TM, TN, WM, WN, DB>;
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
