// igemm_pl.h -- the r-contiguous x r-contiguous contractions (convolution forward / data gradient, Linear as a 1x1 convolution) on
// the bf16 matrix pipe with BOTH operands already split: activations and weights arrive as pl32 planes (bf3_planes.h) and are copied
// global -> LDS by `buffer_load_dwordx4 ... lds`.  Same arithmetic as igemm_bf3.h -- the same three bf16 terms per operand, the same six
// partial products per 16-deep step in the same order, fp32 accumulate, the same sign schedule and split-K plan -- so for one tile shape
// the results are bit-identical to igemm_bf3_kernel's; what is gone is the consumer-side work that bounded it (profiles/r02: ~11 VALU per
// MFMA, matrix pipe busy 0.26-0.30): no global -> VGPR gather, no split, no ds_write, no zero-fill select.  Per k-tile and wave what is
// left beside the MFMAs is: one bit test + select per DMA piece, the fragment reads, two barriers.
//
// LDS image (per buffer): A planes [3][BM rows][64 B], B planes [3][BN rows][64 B], unpadded.  One DMA wave-instruction writes 1 KB =
// 16 rows of ONE plane (lane l -> row l >> 2, 16-byte slot l & 3).  The MFMA fragment read is a ds_read_b128 by 16-lane groups
// {0-3,12-15,20-27}, ... (MI355X_MICROARCH.md, LDS): with 64-byte rows four consecutive rows share a 256-byte bank row, so slot s of row r
// holds k-chunk s ^ ((r >> 2) & 3): within every group the four rows of each residue mod 4 get four distinct slots -> conflict-free.
// (The swizzle is applied to the per-lane SOURCE address; the LDS destination of a DMA is lane-linear by construction.)
//
// Problem interface: the fp32 DMA functors of problems.h (dma_a_tile / dma_a_row / dma_tap / dma_b_tile / dma_b_row) + dma_a_origin() /
// dma_b_origin(); every byte offset they return is a multiple of 128 and maps to the planes by x -> 1.5 x (bf3_planes.h).
#pragma once
#include "igemm_dma.h"
#include "igemm_bf3.h"

namespace hab {

__device__ __forceinline__ int pl_swz(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ uint32_t pl_scale(uint32_t fp32_bytes) { return fp32_bytes + (fp32_bytes >> 1); }

template <class P, int TM, int TN, int WM, int WN, bool DB>
struct IgemmPlCfg {
    static constexpr int NT = WM * WN * 64, NW = WM * WN;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = IGEMM_BK;
    static_assert(BK == 32, "one pl32 group per operand row and k-tile");
    static constexpr int A_CHUNKS = BM / 16, B_CHUNKS = BN / 16;  // 16-row pieces: one DMA wave-instruction per piece and plane
    static constexpr int A_UNITS = (A_CHUNKS + NW - 1) / NW, B_UNITS = (B_CHUNKS + NW - 1) / NW;
    static constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;  // pl16 units
    static constexpr int TILE = 3 * (A_PLANE + B_PLANE);
    static constexpr size_t LDS_BYTES = (size_t)(DB ? 2 : 1) * TILE * sizeof(pl16);
};

template <class P, int TM, int TN, int WM, int WN, bool DB>
__global__ void __launch_bounds__(WM* WN * 64) igemm_pl_kernel(const P p, const pl16* __restrict__ apl, const pl16* __restrict__ bpl, const int k_per_split,
                                                               float* __restrict__ partial, const int sign_schedule, const int nsplit, const int ablate) {
    using Cfg = IgemmPlCfg<P, TM, TN, WM, WN, DB>;
    constexpr int NW = Cfg::NW, BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int A_UNITS = Cfg::A_UNITS, B_UNITS = Cfg::B_UNITS, A_CHUNKS = Cfg::A_CHUNKS, B_CHUNKS = Cfg::B_CHUNKS;
    constexpr int A_PLANE = Cfg::A_PLANE, B_PLANE = Cfg::B_PLANE, TILE = Cfg::TILE;
    extern __shared__ __attribute__((aligned(1024))) pl16 smem_pl[];
    pl16* As = smem_pl;                 // [buf][plane][BM][32]
    pl16* Bs = smem_pl + 3 * A_PLANE;   // [buf][plane][BN][32]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

    // ---- block -> (tile, K slice): as igemm_bf3_kernel ----
    const int nt_m = cdiv(p.M, BM), nt_n = cdiv(p.N, BN);
    const int ntiles = nt_m * nt_n;
    int tile, kz;
    if (nsplit == 1) {
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        kz = 0;
    } else if (nsplit >= 16) {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        tile = slot % ntiles;
        kz = (slot / ntiles) * 8 + xcd;
        if (kz >= nsplit) return;
    } else {
        tile = blockIdx.x % ntiles;
        kz = blockIdx.x / ntiles;
    }
    const int tile_n = tile % nt_n, tile_m = tile / nt_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = kz * k_per_split;
    const int k_end = min(p.K, k_begin + k_per_split);
    const int ntk = max(0, cdiv(k_end - k_begin, BK));

    // ---- DMA descriptors: the fp32 windows of problems.h, scaled to the planes (6 bytes per element) ----
    const DmaTile ta = p.dma_a_tile(m0), tb = p.dma_b_tile(n0);
    const long long ea = ta.base - p.dma_a_origin(), eb = tb.base - p.dma_b_origin();  // element offsets of the windows (multiples of 32)
    const uint64_t reca = (uint64_t)ta.records + (ta.records >> 1), recb = (uint64_t)tb.records + (tb.records >> 1);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(apl) + ea * 6), 0, (int)(reca > 0x7fffffffull ? 0x7fffffffull : reca), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(bpl) + eb * 6), 0, (int)(recb > 0x7fffffffull ? 0x7fffffffull : recb), 0x00020000);
    // unit j of wave w fills rows 16 (j NW + w) .. +15 of the three planes; lane l -> row + (l >> 2), slot l & 3 = k-chunk slot ^ swz(row)
    uint32_t avoff[A_UNITS], amask[A_UNITS], bvoff[B_UNITS];
#pragma unroll
    for (int j = 0; j < A_UNITS; ++j) {
        const int row = 16 * (j * NW + wave) + (lane >> 2);
        const int kc = (lane & 3) ^ pl_swz(row & 31);
        avoff[j] = pl_scale(p.dma_a_row(ta, m0 + row, amask[j])) + (uint32_t)kc * 16u;
        if (A_CHUNKS % NW != 0 && j * NW + wave >= A_CHUNKS) amask[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < B_UNITS; ++j) {
        const int row = 16 * (j * NW + wave) + (lane >> 2);
        const int kc = (lane & 3) ^ pl_swz(row & 31);
        uint32_t ok;
        bvoff[j] = pl_scale(p.dma_b_row(tb, n0 + row, ok)) + (uint32_t)kc * 16u;
        if (!ok || (B_CHUNKS % NW != 0 && j * NW + wave >= B_CHUNKS)) bvoff[j] = DMA_OOB;
    }
    // fragment read offsets (pl16 units inside a plane): lane (li, hi) reads row li, k-chunk 2 s + hi of 16-deep step s
    int aoff[TM][2], boff[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) aoff[i][s] = ((wm * TM + i) * 32 + li) * 32 + (((2 * s + hi) ^ pl_swz(li)) * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) boff[j][s] = ((wn * TN + j) * 32 + li) * 32 + (((2 * s + hi) ^ pl_swz(li)) * 8);

    auto issue = [&](int kt, int buf) {
        if (ablate & 1) return;  // development (HAB_PL_ABLATE): no operand traffic
        const int k0 = k_begin + kt * BK;
        int tap;
        uint32_t sa, sb;
        p.dma_tap(k0, tap, sa, sb);  // block-uniform: tap index, fp32 byte offsets of the tap (+ channel group) for A and B
        sa = pl_scale(sa);
        sb = pl_scale(sb);
        const uint32_t bit = 1u << tap;
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            if (A_CHUNKS % NW != 0 && j * NW + wave >= A_CHUNKS) continue;
            const uint32_t v = (amask[j] & bit) ? avoff[j] : DMA_OOB;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(As + buf * TILE + pl * A_PLANE + (j * NW + wave) * 512),
                                                         16, (int)v, (int)(sa + pl * 64u), 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) {
            if (B_CHUNKS % NW != 0 && j * NW + wave >= B_CHUNKS) continue;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(Bs + buf * TILE + pl * B_PLANE + (j * NW + wave) * 512),
                                                         16, (int)bvoff[j], (int)(sb + pl * 64u), 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    // sign schedule of igemm_bf3.h: every second (tile, slice) accumulates the negated sum -- here the weight fragments are negated in
    // registers (one v_xor per fragment dword, <= 1 per MFMA) since nothing touches the operand on its way into LDS
    const bool flip_all = sign_schedule && (((tile_m + tile_n + kz) & 1) != 0);
    const unsigned sgn = flip_all ? 0x80008000u : 0u;

    auto compute = [&](const pl16* a, const pl16* b) {
        if (ablate & 2) return;  // development: no fragment reads, no MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(a + pl * A_PLANE + aoff[i][s]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    u32x4 w = *reinterpret_cast<const u32x4*>(b + pl * B_PLANE + boff[j][s]);
                    w[0] ^= sgn; w[1] ^= sgn; w[2] ^= sgn; w[3] ^= sgn;
                    bf[j][pl] = __builtin_bit_cast(bf16x8, w);
                }
            // six partial products, smallest weight first (igemm_bf3.h)
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
    };

    if constexpr (!DB) {
        // single buffer: several workgroups per CU hide each other's DMA latency (igemm_dma.h measured this better than a second buffer
        // for narrow tiles)
        for (int kt = 0; kt < ntk; ++kt) {
            issue(kt, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's DMA writes have landed
            __syncthreads();
            compute(As, Bs);
            __syncthreads();
        }
    } else {
        // double buffer: the DMA of k-tile kt + 1 is issued right after the barrier that publishes k-tile kt and lands under its MFMAs;
        // that barrier also orders the DMA behind every wave's reads of the buffer it overwrites (k-tile kt - 1)
        if (ntk > 0) issue(0, 0);
        for (int kt = 0; kt < ntk; kt += 2) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
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
    }

    if (flip_all) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = -acc[i][j][v];
    }

    // ---- epilogue: the problem's own (same accumulator layout as igemm_kernel / igemm_bf3_kernel) ----
    if ((ablate & 4) && acc[0][0][0] != 12345.f) return;  // development: no output traffic
    const bool split = nsplit > 1;
    if constexpr (EpiV4<P>::value) {
        if (split)
            igemm_partial_v4<P, TM, TN>(p, acc, partial + (size_t)kz * p.M * p.N, m0 + wm * TM * 32, n0 + wn * TN * 32, li, hi);
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
inline int igemm_pl_launch(const P& p, const pl16* apl, const pl16* bpl, float* ws, size_t ws_floats, int target_blocks, hipStream_t stream) {
    using Cfg = IgemmPlCfg<P, TM, TN, WM, WN, DB>;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % 32) || !apl || !bpl) return HAB_ERR_ARG;
    const IgemmPlan pl = igemm_plan(Cfg::BM, Cfg::BN, p.M, p.N, p.K, target_blocks, 4096, ws ? ws_floats : 0);
    auto kern = igemm_pl_kernel<P, TM, TN, WM, WN, DB>;
    static bool attr_set = false;
    if (!attr_set && Cfg::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ntiles = cdiv(p.M, Cfg::BM) * cdiv(p.N, Cfg::BN);
    const int grid = pl.splits >= 16 ? ntiles * ((pl.splits + 7) / 8 * 8) : ntiles * pl.splits;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    static const int ablate = hab_env_int("HAB_PL_ABLATE", 0);
    kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, apl, bpl, pl.k_per_split, ws, sign_schedule, pl.splits, ablate);
    HAB_LAUNCH_CHECK();
    if (pl.splits > 1) {
        igemm_splitk_reduce<P>(p, ws, pl.splits, stream);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

}  // namespace hab
