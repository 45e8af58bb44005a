// obs_conv_bf3_ws.h -- wave-specialised form of obs_conv_bf3_kernel (EXPERIMENT: built and ISA-checked offline, never run; see
// tools/experiments/README.md).  SimpleCNN's first convolution with the observation ingest fused: the call site bench.py's roofline
// record is quoted on (HBM-bound by its algorithmic bytes, measured at 0.25 of the HBM peak: profiles/r02_c2_bench.json).
//
// What the counters of obs_conv_bf3_kernel say (profiles/r02_layers_sq_counters.txt): 305 VALU instructions and 30 MFMAs per wave and
// k-tile, the matrix pipe busy 0.26 of the time, the waves waiting 0.39 of theirs -- gather, convert, LDS write, barrier, MFMAs,
// barrier, one after the other in every wave, with ONE k-tile (~1000 clocks) of look-ahead on loads that come from HBM.  Here:
//   * 4 producer waves gather FOUR k-tiles ahead (the 8x8 RGB-D filter is exactly four 64-element k-tiles: register set = k-tile
//     index, the set loaded for tile j + 1 while tile j is converted), convert k-tile s + 1 and write it to LDS image (s + 1) & 1;
//   * 4 consumer waves (one per SIMD, beside one producer each) issue the 15 MFMAs of k-tile s on image s & 1 and, after the last
//     k-tile's barrier, store the tile -- the producers are already two k-tiles into the next one;
//   * the weight fragments of all four k-tiles (one sign) stay in the consumers' REGISTERS for the workgroup's whole life instead of
//     being copied to LDS and read back per k-tile; the sign schedule alternates per workgroup ((blockIdx.x >> 3) & 1: consecutive
//     tiles of an XCD's run);
//   * one barrier per k-tile.  128-row tiles (2 x 32 KB A images of LDS), one workgroup per CU (register-limited: 2 waves per SIMD).
// Arithmetic per output element is obs_conv_bf3_kernel's (same products, same order inside a k-tile); only which tiles accumulate the
// negated sum differs.
#pragma once
#include "obs_conv_bf3.h"

namespace hab {

struct ObsWsCfg {
    static constexpr int NT = 512, NP = 256, BM = 128, KT = 4;
    static constexpr int A_RGB = BM * OBF_RGBP, A_DEP = BM * OBF_DEPP;      // one plane (bf16 elements)
    static constexpr int IMG = A_RGB + 3 * A_DEP;                            // one A image
    static constexpr size_t LDS_BYTES = (size_t)2 * IMG * 2 + 32 * sizeof(float);  // two A images + the bias row
    static constexpr int A_UNITS = BM * 4 / NP;
    static_assert(A_UNITS * 64 == BM, "unit j of a producer thread is row (t >> 2) + 64 j");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__global__ void __launch_bounds__(512) obs_conv_bf3_ws_kernel(const ObsConvFwdProb p, const unsigned short* __restrict__ planes, const int NPAD) {
    using P = ObsConvFwdProb;
    using Cfg = ObsWsCfg;
    constexpr int BM = Cfg::BM, KT = Cfg::KT, A_UNITS = Cfg::A_UNITS;
    static_assert(EpiV4<P>::value, "transposed-accumulator epilogue");

    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // scalar: the role branch is a uniform branch
    const bool producer = wave < 4;
    const int li = lane & 31, hi = lane >> 5;

    const int ntiles = cdiv(p.M, BM);  // p.N <= 32: one column tile
    auto tile_of = [&](int vb) {       // XCD-contiguous runs of M-tiles (igemm.h)
        const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7, idx = vb >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    int vb = blockIdx.x;
    if (vb >= ntiles) return;  // whole workgroup, before any barrier
    const bool neg = ((blockIdx.x >> 3) & 1) != 0;

    // bias through LDS: loaded from global inside the tile loop it costs the consumers an L2 round trip per tile and, worse, makes the
    // loop's first fragment read wait for vmcnt(0) -- i.e. for the previous tile's STORES (seen in the ISA: the fragment registers
    // were the bias registers); held in registers it does not fit beside the 192 registers of weight fragments
    float* biasL = reinterpret_cast<float*>(smem16 + 2 * Cfg::IMG);
    if (t < 32) biasL[t] = (p.bias && t < p.N) ? p.bias[t] : 0.f;  // visible after barrier 0

    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    // ------------------------------------------------------------------ producer waves
    if (producer) {
        struct Rgb12 { uint32_t d0, d1, d2; };
        const uint8_t* rgb_ptr[A_UNITS];
        const float* dep_ptr[A_UNITS];
        Rgb12 a_rgb[KT][A_UNITS];   // register set = k-tile index
        f32x4 a_dep[KT][A_UNITS];
        int offk[KT];                // first of the unit's four taps, per k-tile (pad == 0: every tap of a valid pixel is inside)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            int kh, kw;
            p.g.dKW.divmod(kt * 16 + (t & 3) * 4, kh, kw);
            offk[kt] = kh * p.g.W + kw;
        }
        // frame -> arena row: a tile of 128 output pixels touches at most two frames (Ho * Wo >= 128), whose rows are loaded ONE TILE
        // AHEAD of the setup that uses them.  Loaded inside setup() they are followed by s_waitcnt vmcnt(0) -- the counter is in
        // order -- which drains the three k-tiles of gathers in flight once per tile (seen in the ISA of the first version).
        int idx_img0 = 0, idx_srow0 = 0, idx_srow1 = 0;
        auto load_idx = [&](int tile) {
            const int m0 = tile * BM;
            idx_img0 = __builtin_amdgcn_readfirstlane(p.g.dHoWo.div(m0 < p.M ? m0 : p.M - 1));
            idx_srow0 = p.obs.srow(idx_img0);
            idx_srow1 = p.obs.srow(idx_img0 + 1 < p.g.B ? idx_img0 + 1 : idx_img0);
        };
        auto next_tile = [&](int v) { return v + (int)gridDim.x < ntiles ? tile_of(v + gridDim.x) : tile_of(v); };
        auto setup = [&](int tile) {  // unit j = output pixel m0 + (t >> 2) + 64 j, taps 4 (t & 3) .. +3 of the k-tile; uses load_idx(tile)
            const int m0 = tile * BM;
            const int img0 = idx_img0, srow0 = idx_srow0, srow1 = idx_srow1;
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j) {
                int m = m0 + (t >> 2) + 64 * j;
                m = m < p.M ? m : p.M - 1;  // rows past M: clamped, their accumulators are never stored
                int img, rem, ho, wo;
                p.g.dHoWo.divmod(m, img, rem);
                p.g.dWo.divmod(rem, ho, wo);
                const size_t pix = ((size_t)(img == img0 ? srow0 : srow1) * p.g.H + ho * p.g.stride) * p.g.W + wo * p.g.stride;
                rgb_ptr[j] = p.obs.rgb + pix * 3;
                dep_ptr[j] = p.obs.depth + pix;
            }
        };
        auto fetch = [&](auto k_) {
            constexpr int S = decltype(k_)::value;
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j) {
                a_rgb[S][j] = *reinterpret_cast<const Rgb12*>(rgb_ptr[j] + offk[S] * 3);
                a_dep[S][j] = ld4(dep_ptr[j] + offk[S]);
            }
        };
        auto stage = [&](auto k_) {  // register set S -> A image S & 1
            constexpr int S = decltype(k_)::value;
            unsigned short* Argb = smem16 + (S & 1) * Cfg::IMG;
            unsigned short* Adep = Argb + Cfg::A_RGB;
#pragma unroll
            for (int j = 0; j < A_UNITS; ++j) {
                const int row = (t >> 2) + 64 * j, unit = t & 3;
                const unsigned d[3] = {a_rgb[S][j].d0, a_rgb[S][j].d1, a_rgb[S][j].d2};
                unsigned f[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) f[e] = __float_as_uint((float)((d[e >> 2] >> (8 * (e & 3))) & 0xffu));  // exact in bf16
                unsigned short* dst = Argb + row * OBF_RGBP + unit * 12;
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    u32x2 wv;
                    wv[0] = bf3_pack(f[4 * h], f[4 * h + 1]);
                    wv[1] = bf3_pack(f[4 * h + 2], f[4 * h + 3]);
                    *reinterpret_cast<u32x2*>(dst + 4 * h) = wv;
                }
                unsigned short* dd = Adep + row * OBF_DEPP + unit * 4;
                bf3_store4(a_dep[S][j], dd, dd + Cfg::A_DEP, dd + 2 * Cfg::A_DEP);
            }
        };
        load_idx(tile_of(vb));
        setup(tile_of(vb));
        fetch(K0());
        load_idx(next_tile(vb));  // same position relative to the gathers as inside the loop: the waits merge to vmcnt(12), not 0
        fetch(K1()); fetch(K2()); fetch(K3());
        stage(K0());
        __syncthreads();  // barrier 0
        for (;;) {
            // iteration (tile, kt): gather (next tile, kt) -> set kt (converted one iteration ago), convert (tile, kt + 1) -> other image.
            // The gather is unconditional (the last tile re-reads itself): a conditional one makes the s_waitcnt pass merge the
            // no-load path and wait for the loads just issued (see igemm_bf3_ws.h).  sched_barrier keeps the conversion of the NEW
            // registers from being hoisted above the s_barrier.
            const int vb2 = vb + gridDim.x;
            const bool more = vb2 < ntiles;
            setup(next_tile(vb));
            fetch(K0());
            load_idx(next_tile(more ? vb2 : vb));  // consumed by the next iteration's setup
            __builtin_amdgcn_sched_barrier(0);
            stage(K1());
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            fetch(K1());
            __builtin_amdgcn_sched_barrier(0);
            stage(K2());
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            fetch(K2());
            __builtin_amdgcn_sched_barrier(0);
            stage(K3());
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            fetch(K3());
            __builtin_amdgcn_sched_barrier(0);
            if (more) stage(K0());
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            if (!more) break;
            vb = vb2;
        }
        return;
    }

    // ------------------------------------------------------------------ consumer waves: rows (wave - 4) * 32 .. +31 of the tile
    const int wm = wave - 4;
    f32x16 acc[1][1];
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[0][0][v] = 0.0f;
    // The weight fragments of all four k-tiles live in REGISTERS for the workgroup's whole life (48 x 16 bytes per lane = 192 VGPRs;
    // a consumer wave may use 256: two waves per SIMD).  Through LDS they would be 12 of the 18 fragment reads of every k-tile:
    // 72 KB of LDS reads + 32 KB of A writes per k-tile and CU = ~820 clocks at 128 B/clock against 480 clocks of MFMA.
    // planes = rgb [3][KT][NPAD][48] then depth [3][KT][NPAD][16] (a second set holds -w); lane l supplies row l & 31, k 8 (l >> 5) .. +7
    bf16x8 br[KT][3][3], bd[KT][3];
    {
        const unsigned short* src_set = planes + (neg ? (size_t)3 * NPAD * p.K : 0);
        const size_t rgb_plane = (size_t)KT * NPAD * OBF_RGB, dep_plane = (size_t)KT * NPAD * OBF_DEP;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    br[kt][g][pl] = *reinterpret_cast<const bf16x8*>(src_set + pl * rgb_plane + ((size_t)kt * NPAD + li) * OBF_RGB + g * 16 + hi * 8);
                bd[kt][pl] = *reinterpret_cast<const bf16x8*>(src_set + 3 * rgb_plane + pl * dep_plane + ((size_t)kt * NPAD + li) * OBF_DEP + hi * 8);
            }
    }
    auto mfma_tile = [&](auto k_) {
        constexpr int S = decltype(k_)::value;
        const unsigned short* Argb = smem16 + (S & 1) * Cfg::IMG;
        const unsigned short* Adep = Argb + Cfg::A_RGB;
        bf16x8 ar[3], ad[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) ar[g] = *reinterpret_cast<const bf16x8*>(Argb + (wm * 32 + li) * OBF_RGBP + g * 16 + hi * 8);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ad[pl] = *reinterpret_cast<const bf16x8*>(Adep + pl * Cfg::A_DEP + (wm * 32 + li) * OBF_DEPP + hi * 8);
        // rgb: three k-groups, A exact in one plane, smallest weight plane first (obs_conv_bf3_kernel's order)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[S][g][pl], ar[g], acc[0][0], 0, 0, 0);
        // depth: one k-group, both operands split
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bd[S][PB[q]], ad[PA[q]], acc[0][0], 0, 0, 0);
    };
    // every weight fragment has landed BEFORE the loop: with loads still pending at the loop entry, the s_waitcnt pass keeps a
    // vmcnt(0) in front of the loop's first MFMA (its operand happened to be the last load issued), which on every later pass waits
    // for the previous tile's stores (seen in the ISA)
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();  // barrier 0: image 0 holds (first tile, k-tile 0)
    for (;;) {
        const int vb2 = vb + gridDim.x;
        const bool more = vb2 < ntiles;
        mfma_tile(K0());
        __syncthreads();
        mfma_tile(K1());
        __syncthreads();
        mfma_tile(K2());
        __syncthreads();
        mfma_tile(K3());
        __syncthreads();
        // behind the barrier: the producers are converting k-tile 1 of the next tile while this tile is stored
        if (neg) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[0][0][v] = -acc[0][0][v];
        }
        {   // HAB_BIAS_RELU_VEC4's vector path (N % 4 == 0: checked by the launcher) with the bias quad from LDS.  Transposed accumulator:
            // lane = output row li, register quad g = columns 8 g + 4 hi .. +3.  The element-wise tail path of epi_store4 must not be
            // instantiated here: its (never executed) bias loads target accumulator registers and put a vmcnt(0) -- a wait for the
            // previous tile's stores -- in front of the next tile's first MFMA (seen in the ISA).
            const typename P::EpiRow erow = p.epi_row(tile_of(vb) * BM + wm * 32 + li);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 8 * g + 4 * hi;
                f32x4 v;
                v[0] = acc[0][0][4 * g]; v[1] = acc[0][0][4 * g + 1]; v[2] = acc[0][0][4 * g + 2]; v[3] = acc[0][0][4 * g + 3];
                v += *reinterpret_cast<const f32x4*>(biasL + n);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (erow.ok && n + 3 < p.N) {
                    if (p.y) *reinterpret_cast<f32x4*>(p.y + erow.base + n) = v;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[0][0][v] = 0.0f;
        if (!more) break;
        vb = vb2;
    }
}

// returns HAB_OK, an error, or 1 when the problem / workspace does not fit this path (caller falls back to obs_conv_bf3_launch)
inline int obs_conv_bf3_ws_launch(const ObsConvFwdProb& p, float* ws, size_t ws_floats, hipStream_t stream) {
    using Cfg = ObsWsCfg;
    if (!p.quad || p.K != 64 * Cfg::KT || p.M <= 0 || p.N <= 0 || p.N > 32 || (p.N & 3) || p.g.Ho * p.g.Wo < Cfg::BM) return 1;
    const int NPAD = 32;
    const size_t plane_bytes = (size_t)2 * 3 * Cfg::KT * NPAD * OBF_BK * 2;  // +w and -w sets
    if (!ws || ws_floats * 4 < plane_bytes || (reinterpret_cast<uintptr_t>(ws) & 15)) return 1;
    unsigned short* planes = reinterpret_cast<unsigned short*>(ws);
    obs_conv_bf3_split_weights<<<cdiv(2 * NPAD * p.K, 256), 256, 0, stream>>>(p.w, p.N, p.K, NPAD, planes);
    HAB_LAUNCH_CHECK();
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(obs_conv_bf3_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (attr_err != hipSuccess) return (int)attr_err;
    const int ntiles = cdiv(p.M, Cfg::BM);
    const int grid = ntiles < 256 ? ntiles : 256;  // persistent: one workgroup per CU
    obs_conv_bf3_ws_kernel<<<(grid + 7) / 8 * 8, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, planes, NPAD);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
