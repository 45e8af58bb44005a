// igemm_dma_wgrad.h -- LDS-DMA staged weight-gradient contraction of a convolution (fp32 MFMA):
//   dW[(kh,kw,ci)][co] = sum_r X[pixel(r) + tap(kh,kw)][ci] * dY[r][co],      r = (img, ho, wo)
// Same reasoning as igemm_dma.h: every staging instruction is time taken from the fp32 MFMA issue slot, so the two operands
// go global -> LDS by `buffer_load_dwordx4 ... lds` with no VGPR round trip.  Both operands are k-major here:
//   A image [32 pixels][BM = (taps x channels)]   one DMA wave-instruction fills (part of) ONE pixel row: the pixel decode
//                                                 (img, ho, wo) is wave-uniform and runs on the scalar unit, its byte offset
//                                                 travels in the instruction's scalar offset; the per-lane part (filter tap,
//                                                 channel) is fixed for the whole launch.  Padding: out-of-image taps get an
//                                                 out-of-range offset (hardware writes 0); interior pixels skip the test.
//   B image [32 pixels][BN output channels]       contiguous rows of dY.
// MFMA fragments are read with ds_read_b32 along i / j (conflict-free, the compiler pairs them into ds_read2_b32).
#pragma once
#include "igemm_dma.h"

namespace hab {

template <int TM, int WM, int TN>
struct WgradDmaCfg {
    static constexpr int NT = WM * 64, BM = WM * TM * 32, BN = TN * 32, BK = IGEMM_BK;
    static constexpr int A_ROW_INSTR = (BM / 4 + 63) / 64;          // DMA instructions per pixel row of A
    static constexpr int A_INSTR = BK * A_ROW_INSTR;                // per K-tile
    static constexpr int B_ROWS_PER_INSTR = 64 / (BN / 4);          // pixel rows of B per DMA instruction
    static constexpr int B_INSTR = BK / B_ROWS_PER_INSTR;
    static constexpr size_t LDS_BYTES = (size_t)BK * (BM + BN) * sizeof(float);
};

template <int TM, int WM, int TN>
__global__ void __launch_bounds__(WM * 64) igemm_dma_wgrad_kernel(const ConvWgradProb p, const int k_per_split, float* __restrict__ partial) {
    using Cfg = WgradDmaCfg<TM, WM, TN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NW = WM;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    float* As = smem;             // [BK][BM]
    float* Bs = smem + BK * BM;   // [BK][BN]
    const ConvGeom& g = p.g;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int nt_n = cdiv(p.N, BN);
    const int tile_n = blockIdx.x % nt_n, tile_m = blockIdx.x / nt_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kz = blockIdx.z;
    const int k_begin = kz * k_per_split;
    const int k_end = min(p.K, k_begin + k_per_split);
    const int ntk = max(0, cdiv(k_end - k_begin, BK));

    // per-lane part of the A gathers: unit q of a pixel row covers i = m0 + 4*(64*q + lane) .. +3 (one tap, 4 channels)
    uint32_t alane[Cfg::A_ROW_INSTR];
    int akh[Cfg::A_ROW_INSTR], akw[Cfg::A_ROW_INSTR];
    bool aon[Cfg::A_ROW_INSTR];
#pragma unroll
    for (int q = 0; q < Cfg::A_ROW_INSTR; ++q) {
        const int u = 64 * q + lane, i = m0 + 4 * u;
        aon[q] = (u < BM / 4);
        int tap = 0, ci = 0, kh = 0, kw = 0;
        if (i < p.M) {
            g.dC.divmod(i, tap, ci);
            g.dKW.divmod(tap, kh, kw);
            alane[q] = (uint32_t)((kh * g.W + kw) * g.C + ci) * 4u;
        } else {
            alane[q] = DMA_OOB;  // rows beyond M: zeros
        }
        akh[q] = kh; akw[q] = kw;
    }
    // B gathers: instruction b of a K-tile covers pixel rows b*RPI .. +RPI-1; lane -> (row l / (BN/4), 4 channels)
    constexpr int BU = BN / 4, RPI = Cfg::B_ROWS_PER_INSTR;
    const int brow = lane / BU, bcol = n0 + (lane % BU) * 4;
    const uint32_t blane = (bcol < p.N) ? (uint32_t)(brow * p.N + bcol) * 4u : DMA_OOB;
    const size_t xshift = (size_t)g.pad * (g.W + 1) * g.C;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0,
                                                                        (int)dma_records((size_t)p.K * p.N * 4), 0x00020000);
    const bool padded = g.pad > 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    for (int kt = 0; kt < ntk; ++kt) {
        const int k0 = k_begin + kt * BK;
        // the K-tile's window of X starts at the image of its first pixel (32 pixels span <= 32 images of >= 1 pixel)
        const int img0 = g.dHoWo.div(k0);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x) + (size_t)img0 * g.H * g.W * g.C - xshift, 0,
            (int)dma_records(((size_t)(g.B - img0) * g.H * g.W * g.C + xshift) * 4), 0x00020000);
        // ---- A: pixel rows wave, wave + NW, ... ----
        for (int kr = wave; kr < BK; kr += NW) {
            const int r = k0 + kr;  // wave-uniform
            int img, rem, ho, wo;
            g.dHoWo.divmod(r < p.K ? r : p.K - 1, img, rem);
            g.dWo.divmod(rem, ho, wo);
            const int h0 = ho * g.stride - g.pad, w0 = wo * g.stride - g.pad;
            const uint32_t soff = (uint32_t)((((img - img0) * g.H + h0 + g.pad) * g.W + w0 + g.pad) * g.C) * 4u;
            const bool row_on = r < k_end;
            const bool interior = !padded || (h0 >= 0 && w0 >= 0 && h0 + g.KH <= g.H && w0 + g.KW <= g.W);
#pragma unroll
            for (int q = 0; q < Cfg::A_ROW_INSTR; ++q) {
                if (!aon[q]) continue;
                uint32_t v = alane[q];
                if (!row_on) v = DMA_OOB;
                else if (!interior) {
                    const bool ok = ((unsigned)(h0 + akh[q]) < (unsigned)g.H) & ((unsigned)(w0 + akw[q]) < (unsigned)g.W);
                    v = ok ? v : DMA_OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(As + kr * BM + q * 256), 16, (int)v,
                                                         (int)soff, 0, 0);
            }
        }
        // ---- B: dY rows ----
        for (int b = wave; b < Cfg::B_INSTR; b += NW) {
            const int r = k0 + b * RPI + brow;  // per lane
            const uint32_t v = (r < k_end) ? blane : DMA_OOB;
            const uint32_t soff = (uint32_t)(k0 + b * RPI) * (uint32_t)p.N * 4u;  // < 2 GiB: checked by the launcher
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(Bs + b * 256), 16, (int)v, (int)soff, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        __syncthreads();
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) af[i][s] = As[(c * 8 + hi * 4 + s) * BM + (wave * TM + i) * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[j][s] = Bs[(c * 8 + hi * 4 + s) * BN + j * 32 + li];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: OIHW scatter (or split-K slab) ----
    const bool split = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + j * 32 + li;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + (wave * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
                if (row < p.M && col < p.N) {
                    if (split) partial[((size_t)kz * p.M + row) * p.N + col] = acc[i][j][v];
                    else p.store(row, col, acc[i][j][v]);
                }
            }
        }
}

// B-operand soffset is 32-bit: the split's first dY row offset must stay below 2 GiB.
inline bool wgrad_dma_ok(const ConvWgradProb& p) {
    return p.colsum == nullptr && (p.g.C % 4 == 0) && (p.N % 4 == 0) && (size_t)p.K * p.N * 4 < 0x7fffffffull &&
           (size_t)p.g.H * p.g.W * p.g.C * 4 * 40 < 0x7fffffffull && p.g.Ho * p.g.Wo >= 1;
}

template <int TM, int WM, int TN>
inline int igemm_dma_wgrad_launch(const ConvWgradProb& p, float* ws, size_t ws_floats, int target_blocks, hipStream_t stream) {
    using Cfg = WgradDmaCfg<TM, WM, TN>;
    const IgemmPlan pl = igemm_plan(Cfg::BM, Cfg::BN, p.M, p.N, p.K, target_blocks, 4096, ws ? ws_floats : 0);
    auto kern = igemm_dma_wgrad_kernel<TM, WM, TN>;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = (Cfg::LDS_BYTES > 64 * 1024) ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) : hipSuccess;
    if (attr_err != hipSuccess) return (int)attr_err;
    dim3 grid(cdiv(p.M, Cfg::BM) * cdiv(p.N, Cfg::BN), 1, pl.splits);
    kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, pl.k_per_split, ws);
    HAB_LAUNCH_CHECK();
    if (pl.splits > 1) {
        igemm_splitk_reduce<ConvWgradProb>(p, ws, pl.splits, stream);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

}  // namespace hab
