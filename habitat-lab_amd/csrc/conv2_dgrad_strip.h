// conv2_dgrad_strip.h -- data gradient of SimpleCNN's second convolution (4x4 / stride 2, 32 -> 64 channels; 63 x 63 <- 30 x 30 at 256^2
// observations, any input up to 63 x 63 through the runtime-geometry instantiation) with the
// dY STRIP resident in LDS and the FILTER slices resident in registers: the scheme of conv2_fwd_strip.h applied to the merged-stride-class
// form of the data gradient (problems.h ConvDgradMergedProb).
//
//      dX[h][w][ci] = (a1[h][w][ci] > 0) * sum over a, b in {0, 1}, co of  dY[h2 - a][w2 - b][co] * W[co][ph + 2a][pw + 2b][ci]
//      with h = 2 h2 + ph, w = 2 w2 + pw (dY outside 30 x 30 counts as zero; a1 = the convolution's input = conv1's ReLU output)
//
// A "cell" (h2, w2) holds the four stride classes (ph, pw) of 2 x 2 output pixels; all of them read the same four dY pixels.
//   * a workgroup owns R2 cell rows of one frame: the R2 + 1 dY rows are read from HBM once, split once, stored [pixel][64 channels] as
//     three bf16 planes (128-byte pixel rows, 33 pixel columns: one zero column left, two right), 16-byte chunks XOR-swizzled by
//     (pixel >> 1) & 7 (16 consecutive pixels of a read phase cover all 64 banks);
//   * wave w of 8 owns tap (a, b) = w >> 1 of row class ph = w & 1: its filter slice -- 4 k-steps (64 co) x 2 column classes pw x 3
//     planes = 96 VGPRs -- stays in registers; per cell row (32 cells = one MFMA tile) it reads 12 fragments and issues 48 MFMAs
//     (operands swapped: a lane ends up with 4 consecutive input channels of one output pixel);
//   * the partial tiles meet in LDS; the four taps of a row class are summed in tap order, the ReLU mask is applied, one coalesced
//     16-byte store per value quad (two adjacent output pixels per 16 threads).
#pragma once
#include "igemm_bf3.h"
#include "ops.h"

namespace hab {

struct C2dArgs {
    const float* dy;    // [B][30][30][64]
    const float* wd;    // packed filter [32 ci][4][4][64 co] (the engine's data-gradient layout)
    const float* mask;  // [B][63][63][32] or null: dX *= (mask > 0)
    float* dx;          // [B][63][63][32]
    int B;
    int strips, items;
    int sign_schedule;
    int H, W, Ho, Wo;   // runtime-geometry instantiation (RT): dX is H x W (<= 63 x 63), dY is Ho x Wo
};

template <int R2>
struct C2dCfg {
    static constexpr int H = 63, Ho = 30, Wo = 30, NT = 512, CELLS = 32;
    static constexpr int YRS = R2 + 1, YCOLS = 33, YPIX = YRS * YCOLS;
    static constexpr int YU = YRS * Wo * 16, YPT = (YU + NT - 1) / NT;   // staging units: 4 channels of one dY pixel
    static constexpr int Y_PLANE = YPIX * 64;                            // bf16 elements
    static constexpr int RED_LD = 68;
    static constexpr size_t Y_BYTES = ((size_t)3 * Y_PLANE * 2 + 15) / 16 * 16, RED_BYTES = (size_t)8 * 32 * RED_LD * 4;
    static constexpr size_t LDS_BYTES = Y_BYTES + RED_BYTES;
    static_assert(CELLS % R2 == 0, "");
};

// RT = false: the benchmark geometry, every index a compile-time constant; RT = true: H, W, Ho, Wo from the arguments (W <= 63: one
// 32-cell tile per cell row, the LDS image keeps its 33 pixel columns -- the columns beyond Wo stay zero).
template <int R2, bool RT>
__global__ void __launch_bounds__(512) conv2_dgrad_strip_kernel(const C2dArgs a) {
    using Cfg = C2dCfg<R2>;
    constexpr int NT = Cfg::NT;
    const int H = RT ? a.H : Cfg::H, W = RT ? a.W : Cfg::H, Ho = RT ? a.Ho : Cfg::Ho, Wo = RT ? a.Wo : Cfg::Wo;
    const int YU = Cfg::YRS * Wo * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* ys = smem16;                                                             // [plane][YPIX][64], swizzled
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem16) + Cfg::Y_BYTES);    // [8][32][RED_LD]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tap = wave >> 1, ph = wave & 1, ta = tap >> 1, tb = tap & 1;
    const int li = lane & 31, hi = lane >> 5;

    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int per_xcd = (a.items + 7) >> 3, per_wg = (per_xcd + wg_per_xcd - 1) / wg_per_xcd;
    const int xcd_end = min(a.items, (xcd + 1) * per_xcd);
    const int first = min(xcd_end, xcd * per_xcd + jw * per_wg), last = min(xcd_end, first + per_wg);
    if (first >= last) return;

    // border columns (and everything else until staged) read as zero
    for (int i = t; i < (int)(Cfg::Y_BYTES / 16); i += NT) reinterpret_cast<u32x4*>(smem16)[i] = u32x4{0u, 0u, 0u, 0u};

    const bool flip = a.sign_schedule && (blockIdx.x & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;

    // ---- this wave's filter slice: k-step j = output channels 16 j .. +15; lane = (input channel li, 8 output channels) per column class ----
    bf16x8 bw[4][2][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
            const int kh = ph + 2 * ta, kw = pw + 2 * tb, co = j * 16 + hi * 8;
            const float* src = a.wd + ((size_t)(li * 4 + kh) * 4 + kw) * 64 + co;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
            unsigned p[3][4];
            bf3_split2(v0[0], v0[1], p[0][0], p[1][0], p[2][0]);
            bf3_split2(v0[2], v0[3], p[0][1], p[1][1], p[2][1]);
            bf3_split2(v1[0], v1[1], p[0][2], p[1][2], p[2][2]);
            bf3_split2(v1[2], v1[3], p[0][3], p[1][3], p[2][3]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bw[j][pw][pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0] ^ sgn2, p[pl][1] ^ sgn2, p[pl][2] ^ sgn2, p[pl][3] ^ sgn2});
        }

    f32x4 yr[Cfg::YPT];
    int pf_h20 = 0;  // first cell row of the strip whose loads are in yr
    auto fetch = [&](int item) {  // issues the loads only; rows outside the 30 x 30 image load the tensor's first bytes (zeroed at stage)
        const int img = item / a.strips, h20 = (item - img * a.strips) * R2;
        pf_h20 = h20;
#pragma unroll
        for (int j = 0; j < Cfg::YPT; ++j) {
            const int u = t + j * NT;
            const int r = u / (Wo * 16), row = h20 - 1 + r;
            const bool ok = (RT ? u < YU : (Cfg::YU % NT == 0 || u < Cfg::YU)) && (unsigned)row < (unsigned)Ho;
            const size_t off = ok ? ((size_t)img * Ho + row) * (size_t)(Wo * 64) + (size_t)(u - r * (Wo * 16)) * 4 : 0;
            yr[j] = *reinterpret_cast<const f32x4*>(a.dy + off);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < Cfg::YPT; ++j) {
            const int u = t + j * NT;
            if (RT ? u >= YU : (Cfg::YU % NT != 0 && u >= Cfg::YU)) continue;
            const int r = u / (Wo * 16), rem = u - r * (Wo * 16), w = rem >> 4, c4 = rem & 15;
            const int pixidx = r * Cfg::YCOLS + w + 1;
            unsigned short* dst = ys + pixidx * 64 + (((c4 >> 1) ^ ((pixidx >> 1) & 7)) << 3) + (c4 & 1) * 4;
            const bool in_image = (unsigned)(pf_h20 - 1 + r) < (unsigned)Ho;
            bf3_store4(in_image ? yr[j] : f32x4{0.f, 0.f, 0.f, 0.f}, dst, dst + Cfg::Y_PLANE, dst + 2 * Cfg::Y_PLANE);
        }
    };

    __syncthreads();  // the zero fill is complete
    fetch(first);
    for (int item = first; item < last; ++item) {
        stage();
        __syncthreads();
        if (item + 1 < last) fetch(item + 1);
        const int img = item / a.strips, h20 = (item - img * a.strips) * R2;
#pragma unroll 1
        for (int tr = 0; tr < R2; ++tr) {  // one cell row = one 32-cell tile
            // dY pixel of cell li under this wave's tap: row (h2 - a) - (h20 - 1) = tr + 1 - ta, column (w2 - b) + 1 = li + 1 - tb
            const int pixidx = (tr + 1 - ta) * Cfg::YCOLS + li + 1 - tb;
            const int swz = (pixidx >> 1) & 7;
            // ReLU-mask quads of this tile row, requested BEFORE its MFMAs: the epilogue sits behind two barriers, and a load issued
            // there was a full HBM round trip in front of every store (one workgroup per CU: nothing else to run meanwhile)
            f32x4 mk[2];
            if (a.mask) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int idx = t + q * NT, rph = idx >> 9, cell = (idx >> 4) & 31, cq = idx & 15;
                    const int h = 2 * (h20 + tr) + rph, w = 2 * cell + (cq >> 3);
                    const bool ok = h < H && w < W;
                    mk[q] = *reinterpret_cast<const f32x4*>(a.mask + (ok ? (((size_t)img * H + h) * W + w) * 32 + (cq & 7) * 4 : (size_t)0));
                }
            }
            f32x16 acc[2];
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[pw][v] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned short* src = ys + pixidx * 64 + (((2 * j + hi) ^ swz) << 3);
                bf16x8 yf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) yf[pl] = *reinterpret_cast<const bf16x8*>(src + pl * Cfg::Y_PLANE);
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest weight first (A: dY, B: filter)
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int pw = 0; pw < 2; ++pw)  // operands swapped: D[m = input channel][n = cell]
                        acc[pw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][pw][PB[q]], yf[PA[q]], acc[pw], 0, 0, 0);
            }
            if (tr > 0) __syncthreads();  // the previous tile's reduction has read `red`
            float* mine = red + (size_t)(wave * 32 + li) * Cfg::RED_LD;
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(mine + pw * 32 + 8 * g + 4 * hi) =
                        f32x4{acc[pw][4 * g], acc[pw][4 * g + 1], acc[pw][4 * g + 2], acc[pw][4 * g + 3]};
            __syncthreads();
            // ---- per row class: the four taps in tap order, ReLU mask, store.  Value quad idx: row class, cell, (pw, 4 channels) ----
            const int h2 = h20 + tr;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = t + q * NT, rph = idx >> 9, cell = (idx >> 4) & 31, cq = idx & 15;
                f32x4 s = *reinterpret_cast<const f32x4*>(red + (size_t)(rph * 32 + cell) * Cfg::RED_LD + cq * 4);
#pragma unroll
                for (int t4 = 1; t4 < 4; ++t4)
                    s += *reinterpret_cast<const f32x4*>(red + (size_t)((2 * t4 + rph) * 32 + cell) * Cfg::RED_LD + cq * 4);
                if (flip) s = -s;
                const int h = 2 * h2 + rph, w = 2 * cell + (cq >> 3);
                if (h < H && w < W) {
                    const size_t off = (((size_t)img * H + h) * W + w) * 32 + (cq & 7) * 4;
                    if (a.mask) {
                        const f32x4 m = mk[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[e] = m[e] > 0.f ? s[e] : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(a.dx + off) = s;
                }
            }
        }
        __syncthreads();  // every wave is done with this strip's image (and with `red`)
    }
}

inline bool conv2_dgrad_strip_covers(const ConvDesc& d) {
    return d.KH == 4 && d.KW == 4 && d.stride == 2 && d.pad == 0 && d.C == 32 && d.Cout == 64 && d.H >= 4 && d.W >= 4 && d.H <= 63 && d.W <= 63;
}

// 1: shape not covered.
inline int conv2_dgrad_strip(const ConvDesc& d, const float* dy, const float* wd, const float* mask, const float* add, float* dx,
                             hipStream_t stream) {
    if (!conv2_dgrad_strip_covers(d)) return 1;
    if (add || d.B < 16 || !dy || !wd || !dx) return 1;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(wd) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(mask)) & 15) return 1;
    constexpr int R2 = 2;
    using Cfg = C2dCfg<R2>;
    C2dArgs a;
    a.dy = dy; a.wd = wd; a.mask = mask; a.dx = dx; a.B = d.B;
    a.H = d.H; a.W = d.W; a.Ho = (d.H - 4) / 2 + 1; a.Wo = (d.W - 4) / 2 + 1;
    a.strips = ((d.H + 1) / 2 + R2 - 1) / R2;  // cell rows = ceil(H / 2)
    if ((long long)d.B * a.strips > 0x7fffffffLL) return 1;
    a.items = d.B * a.strips;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = [] {
        const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dgrad_strip_kernel<R2, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dgrad_strip_kernel<R2, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        return e0 != hipSuccess ? e0 : e1;
    }();
    if (attr_err != hipSuccess) return (int)attr_err;
    int grid = 256;
    while (grid > 8 && grid > a.items) grid -= 8;
    if (d.H == 63 && d.W == 63) conv2_dgrad_strip_kernel<R2, false><<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(a);
    else conv2_dgrad_strip_kernel<R2, true><<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
