// obs_conv_patch.h -- SimpleCNN's first convolution (8x8 / 4 on uint8 rgb + fp32 depth, 4 -> 32 channels, simple_cnn.py:68-74,139-156)
// with the OBSERVATION PATCH resident in LDS.  The dominant call site of the C2 cycle and bench.py's roofline record.
//
// Why.  The im2col forms (obs_conv_bf3.h and its producer / consumer variant) gather every input pixel once per output pixel that
// touches it: 4x (8x8 window, stride 4), as 12-byte rgb + 16-byte depth pieces -- 32 small loads per output pixel, each converted
// (uint8 -> bf16, depth -> 3 bf16 terms) again.  Measured: 2.0-2.2 TB/s of algorithmic traffic, 0.26-0.28 of the HBM roofline that
// bounds this layer, with the matrix pipe busy 0.26 (profiles/r02_c2_sq_counters.txt).  Here a workgroup owns TH = 4 output rows of ONE
// image: it reads the 4 TH + 4 = 20 input rows they need as whole rows (768 B of rgb + 1 KB of depth each: fully coalesced, each
// byte fetched once per tile -- 1.25x over all tiles), converts every element ONCE, and keeps the patch in LDS as bf16 images:
//      P0 [20][W][4]   (r, g, b, d0)   uint8 values are exact in bf16; d0 = rn16(depth)
//      P1 [20][W], P2 [20][W]          the two residual terms of the exact depth split (bf3_split.h)
// With k = (kw, c) inside a filter row, the A fragment of output pixel (ho, wo) and filter row kh is 64 CONTIGUOUS bytes of P0 at pixel
// (4 ho + kh, 4 wo): no im2col image exists anywhere.  The weights (32 x 256, three bf16 planes, 1/255 folded into the rgb columns; 59 KB)
// are LDS-resident for the life of the persistent workgroup.  The six partial products of the split scheme (igemm_bf3.h):
//      P0 x {w1, w2, w3}   covers rgb x w (exact operand, 3 products) and d0 x {w1, w2, w3}          48 MFMAs per 32 x 32 tile
//      P1 x {wd1, wd2},  P2 x wd1   the depth residuals against the depth-channel weights, two filter rows per 16-deep step   12 MFMAs
// = the 60 MFMAs of obs_conv_bf3.h, with one fragment read per three of them on the A side.
// Two groups of four waves alternate between a matrix interval and a memory interval (see the kernel).  Measured at 2048 frames:
// 0.80-0.84 ms against 0.96 (obs_conv_bf3_ws.h) and 1.0 (obs_conv_bf3.h); ablation (HAB_OCP_ABLATE): MFMAs alone 0.45, loads + conversion 0.29,
// stores 0.16 -- the two sides still overlap only partly.
#pragma once
#include <utility>

#include "igemm_bf3.h"

namespace hab {

constexpr int OCP_TH = 4;                     // output rows per tile
constexpr int OCP_R = 4 * OCP_TH + 4;         // input rows per tile
constexpr int OCP_KP = 256 + 8;               // pitch (bf16) of a weight row of the rgbd planes: 528 B -> 16 consecutive rows on 16 distinct 16-B slots
constexpr int OCP_DP = 64 + 8;                // pitch of a depth-residual weight row: 144 B

// weight images in the workspace, in LDS order:  rgbd [3][32][OCP_KP] (k = kh*32 + kw*4 + c, rgb columns scaled by 1/255), then
// dep [2][32][OCP_DP] (k = kh*8 + kw; planes w1, w2 of the depth-channel weights)
constexpr int OCP_W_ELEMS = 3 * 32 * OCP_KP + 2 * 32 * OCP_DP;
__global__ void obs_patch_split_weights(const float* __restrict__ w, int N, unsigned short* __restrict__ img) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 32 * 256) return;
    const int n = e >> 8, k = e & 255, c = k & 3;
    float v = n < N ? w[(size_t)n * 256 + k] : 0.f;
    if (c < 3) v *= HAB_RCP255;
    unsigned h1, h2, h3;
    bf3_split(v, h1, h2, h3);
    img[(0 * 32 + n) * OCP_KP + k] = (unsigned short)h1;
    img[(1 * 32 + n) * OCP_KP + k] = (unsigned short)h2;
    img[(2 * 32 + n) * OCP_KP + k] = (unsigned short)h3;
    if (c == 3) {
        unsigned short* d = img + 3 * 32 * OCP_KP;
        const int kd = k >> 2;  // kh*8 + kw
        d[(0 * 32 + n) * OCP_DP + kd] = (unsigned short)h1;
        d[(1 * 32 + n) * OCP_DP + kd] = (unsigned short)h2;
    }
    if (k < 8) {  // zero the row paddings once (never read by a fragment, kept defined)
        img[(0 * 32 + n) * OCP_KP + 256 + k] = 0; img[(1 * 32 + n) * OCP_KP + 256 + k] = 0; img[(2 * 32 + n) * OCP_KP + 256 + k] = 0;
        unsigned short* d = img + 3 * 32 * OCP_KP;
        d[(0 * 32 + n) * OCP_DP + 64 + k] = 0; d[(1 * 32 + n) * OCP_DP + 64 + k] = 0;
    }
}

template <class F, int... Is>
__device__ __forceinline__ void ocp_static_for_impl(F& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>()), ...); }
template <int N, class F>
__device__ __forceinline__ void ocp_static_for(F& f) { ocp_static_for_impl(f, std::make_integer_sequence<int, N>()); }

struct ObsPatchGeom {
    int W, H, Ho, Wo, tiles_per_img, ntiles;
    int units;       // gather units per tile: OCP_R rows x W / 4 pixel quads
    int patch_elems; // bf16 elements of one patch (P0 + P1 + P2), padded
};

// One workgroup = TWO groups of four waves that alternate roles in lock step (one s_barrier per interval):
//      interval p, group p & 1 ("matrix"):  issue the loads of its NEXT tile into registers, then the 120 MFMAs per wave of its current tile
//      the other group ("memory"):          store the previous tile (epilogue), then convert + write its next tile into its own patch buffer
// so that on every SIMD the MFMAs of one wave run beside the loads / conversion / stores of the other -- with a single role sequence per
// wave the three phases simply add up (ablation of the first version: 0.24 + 0.13 + 0.13 ms of 0.52 at 1024 frames).  Both groups share
// the resident weights: planes w1 / w2 (and the residual plane wd1) in LDS, plane w3 and the residual plane wd2 as fragments in registers (80 VGPRs), which is what
// lets two 60 KB patch buffers fit beside them.
template <int UPT>
__global__ void __launch_bounds__(512) obs_conv_patch_kernel(const ObsConvFwdProb p, const ObsPatchGeom gq, const unsigned short* __restrict__ wimg,
                                                              const int sign_schedule, const int ablate) {
    using P = ObsConvFwdProb;
    static_assert(EpiV4<P>::value, "transposed-accumulator epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int W = gq.W;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), group = wave >> 2, gw = wave & 3, tg = t & 255;  // wave-uniform: tile
    // indices, the rows[] indirection (a SCALAR load: a vector one puts a vmcnt(0) -- a wait for this wave's own stores -- in front of every tile) and the role branches stay on the scalar unit
    const int li = lane & 31, hi = lane >> 5;
    unsigned short* W12 = smem16;                                            // [2][32][OCP_KP]: planes w1, w2 of the rgbd weights
    unsigned short* WD1 = W12 + 2 * 32 * OCP_KP;                             // [32][OCP_DP]: plane w1 of the depth-residual weights
    float* biasL = reinterpret_cast<float*>(WD1 + 32 * OCP_DP);              // [32]
    unsigned short* P0 = WD1 + 32 * OCP_DP + 64 + group * gq.patch_elems;    // this group's patch: [OCP_R][W][4]
    unsigned short* P1 = P0 + OCP_R * W * 4;                                 // [OCP_R][W]
    unsigned short* P2 = P1 + OCP_R * W;                                     // [OCP_R][W]

    // ---- resident weights: w1 / w2 -> LDS (linear copy), w3 and the residual planes -> fragment registers ----
    for (int i = t; i < 2 * 32 * OCP_KP / 8; i += 512) reinterpret_cast<u32x4*>(W12)[i] = reinterpret_cast<const u32x4*>(wimg)[i];
    for (int i = t; i < 32 * OCP_DP / 8; i += 512) reinterpret_cast<u32x4*>(WD1)[i] = reinterpret_cast<const u32x4*>(wimg + 3 * 32 * OCP_KP)[i];
    if (t < 32) biasL[t] = (p.bias && t < p.N) ? p.bias[t] : 0.f;
    bf16x8 w3f[8][2], wd2f[4];
    {
        const unsigned short* w3 = wimg + 2 * 32 * OCP_KP + li * OCP_KP + hi * 8;
        const unsigned short* wd = wimg + 3 * 32 * OCP_KP + li * OCP_DP + hi * 8;
#pragma unroll
        for (int kh = 0; kh < 8; ++kh)
#pragma unroll
            for (int s = 0; s < 2; ++s) w3f[kh][s] = *reinterpret_cast<const bf16x8*>(w3 + kh * 32 + s * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) wd2f[j] = *reinterpret_cast<const bf16x8*>(wd + 32 * OCP_DP + j * 16);
    }

    auto tile_of = [&](int vb) {
        const int q = gq.ntiles >> 3, r = gq.ntiles & 7, xcd = vb & 7, idx = vb >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };

    // ---- patch gather: unit u = (row r, pixel quad x4) -> 12 B of rgb + 16 B of depth, both whole-row coalesced ----
    struct Rgb12 { uint32_t d0, d1, d2; };
    Rgb12 rg[UPT];
    f32x4 dp[UPT];
    const int qpr = W >> 2;  // quads per row
    // depth (20 registers) is loaded underneath the tile's own matrix interval, rgb (15) at the start of the memory interval, underneath
    // the stores of the previous tile: with both sets live across the MFMAs the kernel spilled -- and a spill of a loaded value is a wait
    // for the load, i.e. the whole HBM latency inside the matrix interval (first two-group version: 0.59 ms instead of 0.52)
    // (the frame -> arena row indirection is ONE dependent load per tile, not one per gather unit)
    auto tile_base = [&](int tile, int& y0) {
        const int img = tile / gq.tiles_per_img;
        y0 = 4 * (tile - img * gq.tiles_per_img) * OCP_TH;
        return (size_t)p.obs.srow(img) * gq.H;
    };
    auto unit_pix = [&](size_t frame, int y0, int j, bool& ok) {
        const int u = tg + 256 * j;
        const int r = u / qpr, x4 = u - r * qpr;
        const int y = y0 + r;
        ok = (u < gq.units) & (y < gq.H);
        return (frame + (ok ? y : 0)) * W + 4 * (ok ? x4 : 0);
    };
    auto fetch_depth = [&](int tile) {
        if (ablate & 1) return;  // development (HAB_OCP_ABLATE): no observation reads
        int y0;
        const size_t frame = tile_base(tile, y0);
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            bool ok;
            const size_t pix = unit_pix(frame, y0, j, ok);
            dp[j] = ld4(p.obs.depth + pix);
            if (!ok) dp[j] = zero4();
        }
    };
    auto fetch_rgb = [&](int tile) {
        if (ablate & 1) return;
        int y0;
        const size_t frame = tile_base(tile, y0);
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            bool ok;
            const size_t pix = unit_pix(frame, y0, j, ok);
            rg[j] = *reinterpret_cast<const Rgb12*>(p.obs.rgb + pix * 3);
            if (!ok) { rg[j].d0 = 0; rg[j].d1 = 0; rg[j].d2 = 0; }
        }
    };
    auto stage = [&]() {
        if (ablate & 8) return;  // development: no conversion, no patch writes
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            const int u = tg + 256 * j;
            if (u >= gq.units) continue;
            const unsigned d[3] = {rg[j].d0, rg[j].d1, rg[j].d2};
            unsigned f[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) f[e] = __float_as_uint((float)((d[e >> 2] >> (8 * (e & 3))) & 0xffu));  // exact in bf16 (upper 16 bits)
            unsigned a1, a2, a3, b1, b2, b3;  // depth pixels (0,1) and (2,3): packed bf16 pairs of the three terms
            bf3_split2(dp[j][0], dp[j][1], a1, a2, a3);
            bf3_split2(dp[j][2], dp[j][3], b1, b2, b3);
            u32x4 lo, hi4;  // pixels 0,1 and 2,3 of the quad: (r g)(b d0) each
            lo[0] = bf3_pack(f[0], f[1]);  lo[1] = bf3_pack(f[2], a1 << 16);
            lo[2] = bf3_pack(f[3], f[4]);  lo[3] = bf3_pack(f[5], a1 & 0xffff0000u);
            hi4[0] = bf3_pack(f[6], f[7]); hi4[1] = bf3_pack(f[8], b1 << 16);
            hi4[2] = bf3_pack(f[9], f[10]); hi4[3] = bf3_pack(f[11], b1 & 0xffff0000u);
            unsigned short* dst = P0 + (size_t)u * 16;  // u = r * (W/4) + x4 -> pixel (r, 4 x4): 16 bf16 per quad
            *reinterpret_cast<u32x4*>(dst) = lo;
            *reinterpret_cast<u32x4*>(dst + 8) = hi4;
            u32x2 r1, r2;
            r1[0] = a2; r1[1] = b2; r2[0] = a3; r2[1] = b3;
            *reinterpret_cast<u32x2*>(P1 + (size_t)u * 4) = r1;
            *reinterpret_cast<u32x2*>(P2 + (size_t)u * 4) = r2;
        }
    };

    // this wave's two 32-pixel row tiles: output row gw of its group's tile, columns 32 i + li
    const int a_px0 = (4 * gw) * W + 4 * li;  // patch pixel of filter row 0 for half 0; half 1: + 128 pixels
    f32x16 acc[2];
    // 20 steps of 16 reduction elements: steps 0..3 the depth residuals (smallest terms first; two filter rows per step, lane half hi takes
    // row 2 j + hi), steps 4..19 (r, g, b, d0) x {w3, w2, w1} (filter row kh = (st - 4) >> 1, half s = (st - 4) & 1).  The fragment reads of
    // step st + 1 are issued BEFORE the MFMAs of step st (two register sets): left to itself hipcc read each step's fragments right in
    // front of its MFMAs, and with one matrix wave per SIMD every step then sat out the LDS latency (ISA of the first versions).
    bf16x8 fr[2][5];
    auto load_step = [&](auto stc) {
        constexpr int st = decltype(stc)::value, set = st & 1;
        if constexpr (st < 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = a_px0 + (2 * st + hi) * W + 128 * i;
                u32x4 v1, v2;
                const u32x2 l1 = *reinterpret_cast<const u32x2*>(P1 + px), h1 = *reinterpret_cast<const u32x2*>(P1 + px + 4);
                const u32x2 l2 = *reinterpret_cast<const u32x2*>(P2 + px), h2 = *reinterpret_cast<const u32x2*>(P2 + px + 4);
                v1[0] = l1[0]; v1[1] = l1[1]; v1[2] = h1[0]; v1[3] = h1[1];
                v2[0] = l2[0]; v2[1] = l2[1]; v2[2] = h2[0]; v2[3] = h2[1];
                fr[set][i] = __builtin_bit_cast(bf16x8, v1);
                fr[set][2 + i] = __builtin_bit_cast(bf16x8, v2);
            }
            fr[set][4] = *reinterpret_cast<const bf16x8*>(WD1 + li * OCP_DP + (2 * st + hi) * 8);
        } else {
            constexpr int kh = (st - 4) >> 1, s_ = (st - 4) & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fr[set][i] = *reinterpret_cast<const bf16x8*>(P0 + (size_t)(a_px0 + kh * W + 128 * i) * 4 + (2 * s_ + hi) * 8);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fr[set][2 + pl] = *reinterpret_cast<const bf16x8*>(W12 + (pl * 32 + li) * OCP_KP + kh * 32 + s_ * 16 + hi * 8);
        }
    };
    auto flip = [&](bf16x8 f, const unsigned sgn) {  // sign schedule: the A fragments of odd tiles enter negated
        u32x4 v = __builtin_bit_cast(u32x4, f);
        v[0] ^= sgn; v[1] ^= sgn; v[2] ^= sgn; v[3] ^= sgn;
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mma_step = [&](auto stc, const unsigned sgn) {
        constexpr int st = decltype(stc)::value, set = st & 1;
        if constexpr (st < 4) {
            const bf16x8 a1[2] = {flip(fr[set][0], sgn), flip(fr[set][1], sgn)}, a2[2] = {flip(fr[set][2], sgn), flip(fr[set][3], sgn)};
            const bf16x8 wd1 = fr[set][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wd1, a2[i], acc[i], 0, 0, 0);       // d2 x wd1
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wd2f[st], a1[i], acc[i], 0, 0, 0);  // d1 x wd2
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wd1, a1[i], acc[i], 0, 0, 0);       // d1 x wd1
        } else {
            constexpr int kh = (st - 4) >> 1, s_ = (st - 4) & 1;
            const bf16x8 a[2] = {flip(fr[set][0], sgn), flip(fr[set][1], sgn)};
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3f[kh][s_], a[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[set][3], a[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[set][2], a[i], acc[i], 0, 0, 0);
        }
    };
    auto matrix_phase = [&](const unsigned sgn) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;
        if (ablate & 2) return;  // development: no fragment reads, no MFMAs
        load_step(std::integral_constant<int, 0>());
        auto body = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
            if constexpr (st + 1 < 20) load_step(std::integral_constant<int, st + 1>());
            mma_step(stc, sgn);
        };
        ocp_static_for<20>(body);
    };
    // Epilogue = HAB_BIAS_RELU_VEC4's vector path (N % 4 == 0: checked by the launcher), bias quads from LDS: lane = output pixel, register
    // quad g = channels 8 g + 4 hi .. +3.  The problem's generic epi_store4 carries an element-wise tail path whose (never executed) loads
    // and software bf16 rounding cost ~100 instructions, 11 spilled register pairs and a vmcnt(0) per quad here (seen in the ISA).
    auto store_tile = [&](int tile, bool neg) {
        if ((ablate & 4) && acc[0][0] != 12345.f) return;  // development: no output traffic
        const int img = tile / gq.tiles_per_img, ho = (tile - img * gq.tiles_per_img) * OCP_TH + gw;
        if (ho >= gq.Ho) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int wo = 32 * i + li;
            if (wo >= gq.Wo) continue;
            const size_t base = ((size_t)(img * gq.Ho + ho) * gq.Wo + wo) * p.N;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 8 * g + 4 * hi;
                if (n + 3 >= p.N) continue;  // N % 4 == 0: a quad is inside or outside
                f32x4 v;
                v[0] = acc[i][4 * g]; v[1] = acc[i][4 * g + 1]; v[2] = acc[i][4 * g + 2]; v[3] = acc[i][4 * g + 3];
                if (neg) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; v[3] = -v[3]; }
                v += *reinterpret_cast<const f32x4*>(biasL + n);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (p.y) *reinterpret_cast<f32x4*>(p.y + base + n) = v;
            }
        }
    };

    // ---- the two groups' tile sequences: virtual block b + group * G, step 2 G (G = gridDim.x, a multiple of 8: a workgroup stays on
    // its XCD's run of tiles, whose neighbours share halo rows in that XCD's L2) ----
    const int vstep = 2 * gridDim.x;
    const int vb0 = blockIdx.x + group * gridDim.x;
    const int n_mine = vb0 < gq.ntiles ? (gq.ntiles - vb0 + vstep - 1) / vstep : 0;
    const int n_first = (int)blockIdx.x < gq.ntiles ? (gq.ntiles - (int)blockIdx.x + vstep - 1) / vstep : 0;  // group 0's count >= group 1's
    if (n_first == 0) return;
    // prologue: both groups load their first tile; group 0 stages it (group 1 does so in interval 0, as the memory group)
    if (n_mine > 0) { fetch_depth(tile_of(vb0)); if (group == 0) fetch_rgb(tile_of(vb0)); }
    if (group == 0) stage();
    __syncthreads();
    for (int pI = 0; pI <= 2 * n_first; ++pI) {
        if ((pI & 1) == group) {
            // matrix interval of tile k = pI >> 1; the depth rows of the next tile are requested first and land underneath the MFMAs
            const int k = pI >> 1;
            if (k < n_mine) {
                if (k + 1 < n_mine) fetch_depth(tile_of(vb0 + (k + 1) * vstep));
                const int tile = tile_of(vb0 + k * vstep);
                matrix_phase((sign_schedule && (tile & 1)) ? 0x80008000u : 0u);
            }
        } else {
            // memory interval: request the rgb rows of tile ks (the one the next matrix interval computes), convert + write it, THEN store
            // tile kd (computed in the previous interval): vmcnt counts loads and stores together, so a conversion placed behind the stores
            // waits for their completion as well (measured at 2048 frames: 0.83 -> 0.80 ms).  Measured and rejected: rgb requested a whole
            // interval earlier (0.88: 15 more registers live across the MFMAs), s_setprio for the matrix group (no effect).
            const int kd = (pI - 1 - group) / 2;   // group 0: pI = 2 kd + 1; group 1: pI = 2 kd + 2
            const int ks = (pI + 1 - group) / 2;   // group 0: next matrix interval pI + 1 = 2 ks; group 1: pI + 1 = 2 ks + 1
            if (ks < n_mine) {                     // (group 0 never has a memory interval at pI = 0)
                fetch_rgb(tile_of(vb0 + ks * vstep));
                stage();
            }
            if (pI >= 1 + group && kd < n_mine) {
                const int tile_d = tile_of(vb0 + kd * vstep);
                store_tile(tile_d, sign_schedule && (tile_d & 1));
            }
        }
        __syncthreads();
    }
}

// The weight image alone (packed forward weights [N][8][8][4] -> img, OCP_W_ELEMS bf16): what a caller that keeps the weights fixed over
// many calls builds once.  1: this filter shape has no image.
inline int obs_conv_patch_weight_image(const float* w_packed, int N, int KH, int KW, int C, unsigned short* img, hipStream_t stream) {
    if (KH != 8 || KW != 8 || C != 4 || N > 32 || (N & 3) || !w_packed || !img || (reinterpret_cast<uintptr_t>(img) & 15)) return 1;
    obs_patch_split_weights<<<32, 256, 0, stream>>>(w_packed, N, img);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// returns HAB_OK, an error, or 1 when the problem does not fit this path (caller falls back to the im2col kernels)
// wimg_cached: the weight image of p.w built earlier by obs_conv_patch_weight_image (the engine keeps one per optimiser step) or null
// (built here into the workspace, one more launch per call)
inline int obs_conv_patch_launch(const ObsConvFwdProb& p, float* ws, size_t ws_floats, hipStream_t stream, const unsigned short* wimg_cached = nullptr) {
    const ConvGeom& g = p.g;
    if (!p.quad || g.KH != 8 || g.KW != 8 || g.stride != 4 || g.pad != 0 || p.N > 32 || (p.N & 3) || p.K != 256 || g.W > 256 || (g.W & 3) || g.Wo > 64 ||
        g.Wo < 1 || p.M <= 0)
        return 1;
    if (!wimg_cached && (!ws || ws_floats * 4 < (size_t)OCP_W_ELEMS * 2 || (reinterpret_cast<uintptr_t>(ws) & 15))) return 1;
    if (reinterpret_cast<uintptr_t>(wimg_cached) & 15) return 1;
    ObsPatchGeom gq;
    gq.W = g.W; gq.H = g.H; gq.Ho = g.Ho; gq.Wo = g.Wo;
    gq.tiles_per_img = cdiv(g.Ho, OCP_TH);
    gq.ntiles = g.B * gq.tiles_per_img;
    gq.units = OCP_R * (g.W >> 2);
    if (cdiv(gq.units, 256) > 5) return 1;
    // the idle lane (wo = 63) of the last filter rows reads up to 64 bytes past P0 / P1 / P2 of its group's patch: inside the padding
    gq.patch_elems = OCP_R * g.W * 6 + 64;
    const size_t lds = ((size_t)2 * 32 * OCP_KP + 32 * OCP_DP + 64 + (size_t)2 * gq.patch_elems) * 2;
    if (lds > 160 * 1024) return 1;
    const unsigned short* wimg = wimg_cached;
    if (!wimg) {
        unsigned short* built = reinterpret_cast<unsigned short*>(ws);
        obs_patch_split_weights<<<32, 256, 0, stream>>>(p.w, p.N, built);
        HAB_LAUNCH_CHECK();
        wimg = built;
    }
    auto kern = obs_conv_patch_kernel<5>;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    static const int ablate = hab_env_int("HAB_OCP_ABLATE", 0);
    const int pairs = (gq.ntiles + 1) / 2;
    const int grid = pairs < 256 ? (pairs + 7) / 8 * 8 : 256;
    kern<<<grid, 512, lds, stream>>>(p, gq, wimg, sign_schedule, ablate);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
