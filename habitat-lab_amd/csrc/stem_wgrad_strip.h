// stem_wgrad_strip.h -- weight gradient of the GroupNorm-ResNet stem convolution (7x7 / stride 2 / padding 3, 4 input channels -> 32;
// habitat_baselines/rl/ddppo/policy/resnet.py:207-219 `conv1`, reached by total_loss.backward() of rl/ppo/ppo.py:253) with both
// operands resident in LDS in their natural layouts.  Counterpart of stem_conv_strip.h.
//
//      dW[co][ci][kh][kw] = sum over (img, ho, wo) of  dY[img][ho][wo][co] * X[img][2 ho + kh - 3][2 wo + kw - 3][ci]
//
// Why: as an implicit-GEMM weight gradient (igemm_bf3.h, both operands k-strided, register transposes, every x element re-gathered
// 49 / 4 times) this was the slowest single kernel of the ResNet18 learner: 3.7 ms per 4096 frames (57 TFLOP/s-eq), 7 % of a minibatch.
// Here, as in wgrad3x3_bf3.h:
//   * a (persistent) workgroup takes strips of 4 output rows of one frame: the 13 x rows under them (3 zero columns either side) and the
//     4 dY rows are read from HBM once with 16-byte loads, split once (exact 3-term split, igemm_bf3.h) and kept pixel-major --
//     x as [row][column][4 channels] (8 bytes per pixel), dY as [pixel][32 channels] (64 bytes per pixel);
//   * the contraction index is the PIXEL, 16 consecutive wo of one output row per MFMA step; fragments are built by the LDS transpose
//     read ds_read_b64_tr_b16: a 16-lane group passes 16 chunk addresses -- for dY (pixel p, channel quad q), for x (pixel p, TAP q):
//     the x chunk of output pixel wo + p and tap kw0 + q is the 4 channels of input column 2 (wo + p) + kw0 + q -- and lane 4 q' + c
//     receives element c of chunk q' of the four pixels: exactly the k-contiguous fragment of reduction slot (kw0 + q', c).  No
//     im2col, no register transposes, every address aligned;
//   * wave w owns filter row kh = w (7 waves): a 32 x 32 tile [co][(kw, ci) slot, kw = 7 zero-weight padding] of dW accumulated
//     over every strip of the workgroup's life -- no reduction across waves; per k-step 12 transpose reads and 6 MFMAs;
//   * one slab [7][32][32] per workgroup at the end; stem_wgrad_reduce_kernel sums the slabs in slab order (deterministic) into the
//     OIHW gradient, dropping the padding slots.  The next strip's global loads are in flight during the current strip's MFMAs.
// Sign schedule as everywhere on the split path: every second workgroup accumulates the negated sum (dY enters negated).
#pragma once
#include "bf3_split.h"

namespace hab {

typedef short swg_v4s __attribute__((ext_vector_type(4)));
typedef __bf16 swg_bf16x8 __attribute__((ext_vector_type(8)));

struct StemWgArgs {
    const float* x;    // [B][H][W][4]
    const float* dy;   // [B][Ho][Wo][32]
    float* slabs;      // [gridDim.x][7][32][32]
    int B, H, W, Ho, Wo, PW;
    int strips, items;
    int sign_schedule;
    const float* norm;  // null, or the RunningMeanAndVar affine applied to x while staging (stem_conv_strip.h: StemArgs::norm)
};

constexpr int SWG_TH = 4, SWG_XROWS = 2 * SWG_TH + 5, SWG_NT = 448;
constexpr int SWG_XPT = 4, SWG_YPT = 5;  // register-prefetched 16-byte units per thread at the largest covered width (Wo = 64)

__device__ __forceinline__ swg_v4s swg_tr_read(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) swg_v4s*)p);
}
__device__ __forceinline__ swg_bf16x8 swg_join(const swg_v4s lo, const swg_v4s hi) {
    return __builtin_bit_cast(swg_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ void __launch_bounds__(SWG_NT) stem_wgrad_strip_kernel(const StemWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short swg_sm[];
    const int xplane = SWG_XROWS * a.PW * 4;      // bf16 elements per plane of the x strip
    const int yplane = SWG_TH * a.Wo * 32;        // ... of the dY strip
    unsigned short* xs = swg_sm;                  // [3][13][PW][4]
    unsigned short* ys = swg_sm + 3 * xplane;     // [3][TH * Wo][32]
    const int t = threadIdx.x, lane = t & 63;
    const int kh = __builtin_amdgcn_readfirstlane(t >> 6);  // this wave's filter row
    const int xunits = SWG_XROWS * a.PW, yunits = SWG_TH * a.Wo * 8;

    // this workgroup's strips: XCD x takes the x-th eighth of the items, its workgroups contiguous pieces of it (conv2_fwd_strip.h)
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int per_xcd = (a.items + 7) >> 3, per_wg = (per_xcd + wg_per_xcd - 1) / wg_per_xcd;
    const int xcd_end = min(a.items, (xcd + 1) * per_xcd);
    const int first = min(xcd_end, xcd * per_xcd + jw * per_wg), last = min(xcd_end, first + per_wg);

    const bool flip = a.sign_schedule && (blockIdx.x & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;

    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;

    f32x4 xr[SWG_XPT], yr[SWG_YPT];
    f32x4 nb = {1.f, 1.f, 1.f, 1.f}, na = {0.f, 0.f, 0.f, 0.f};
    if (a.norm) { nb = *reinterpret_cast<const f32x4*>(a.norm); na = *reinterpret_cast<const f32x4*>(a.norm + 8); }
    int pf_ho0 = 0;
    // fetch only ISSUES the loads (out-of-image pixels load the tensor's first bytes): the zeros go in at stage()
    auto fetch = [&](int item) {
        const int img = item / a.strips, ho0 = (item - img * a.strips) * SWG_TH;
        pf_ho0 = ho0;
        const float* xb = a.x + (size_t)img * a.H * a.W * 4;
#pragma unroll
        for (int j = 0; j < SWG_XPT; ++j) {
            const int u = t + j * SWG_NT;
            const int r = u / a.PW, c = u - r * a.PW;
            const int hin = 2 * ho0 - 3 + r, win = c - 3;
            const bool ok = (u < xunits) & ((unsigned)hin < (unsigned)a.H) & ((unsigned)win < (unsigned)a.W);
            xr[j] = *reinterpret_cast<const f32x4*>(xb + (ok ? ((size_t)hin * a.W + win) * 4 : 0));
        }
        const float* yb = a.dy + ((size_t)img * a.Ho + ho0) * (size_t)(a.Wo * 32);
        const int yvalid = min(SWG_TH, a.Ho - ho0) * a.Wo * 8;  // units of the rows that exist
#pragma unroll
        for (int j = 0; j < SWG_YPT; ++j) {
            const int u = t + j * SWG_NT;
            yr[j] = *reinterpret_cast<const f32x4*>(yb + (u < yvalid ? (size_t)u * 4 : 0));
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < SWG_XPT; ++j) {
            const int u = t + j * SWG_NT;
            if (u >= xunits) continue;
            const int r = u / a.PW, c = u - r * a.PW;
            const int hin = 2 * pf_ho0 - 3 + r, win = c - 3;
            const bool ok = ((unsigned)hin < (unsigned)a.H) & ((unsigned)win < (unsigned)a.W);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                v = xr[j];
                if (a.norm) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaf(v[q], nb[q], na[q]);
                }
            }
            unsigned a1, a2, a3, b1, b2, b3;
            bf3_split2(v[0], v[1], a1, a2, a3);
            bf3_split2(v[2], v[3], b1, b2, b3);
            unsigned short* dst = xs + (size_t)u * 4;
            *reinterpret_cast<u32x2*>(dst) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(dst + xplane) = u32x2{a2, b2};
            *reinterpret_cast<u32x2*>(dst + 2 * xplane) = u32x2{a3, b3};
        }
        const int yvalid = min(SWG_TH, a.Ho - pf_ho0) * a.Wo * 8;
#pragma unroll
        for (int j = 0; j < SWG_YPT; ++j) {
            const int u = t + j * SWG_NT;
            if (u >= yunits) continue;
            const f32x4 v = u < yvalid ? yr[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned a1, a2, a3, b1, b2, b3;
            bf3_split2(v[0], v[1], a1, a2, a3);
            bf3_split2(v[2], v[3], b1, b2, b3);
            unsigned short* dst = ys + (size_t)u * 4;  // unit u = (pixel u / 8, channel quad u % 8): [pixel][32 channels]
            *reinterpret_cast<u32x2*>(dst) = u32x2{a1 ^ sgn2, b1 ^ sgn2};
            *reinterpret_cast<u32x2*>(dst + yplane) = u32x2{a2 ^ sgn2, b2 ^ sgn2};
            *reinterpret_cast<u32x2*>(dst + 2 * yplane) = u32x2{a3 ^ sgn2, b3 ^ sgn2};
        }
    };

    // lane constants of the transpose reads: 16-lane group (lane >> 4) & 1 serves rows 16 .. 31 of its operand tile, k-block lane >> 5;
    // lane i of a group passes the chunk of pixel (i >> 2) of the group's four and of quad / tap (i & 3)
    const int i16 = lane & 15, gg = (lane >> 4) & 1, kblk = lane >> 5;
    const int p4 = i16 >> 2, q4 = i16 & 3;
    const int ychunk = gg * 16 + q4 * 4;   // dY: channels 16 gg + 4 q .. +3 inside the pixel's 32-channel row
    const int xtap = gg * 4 + q4;          // x: tap kw = 4 gg + q (kw = 7: the zero-weight padding slot)
    const int nsteps = SWG_TH * a.Wo / 16;

    auto compute = [&]() {
        if (kh >= 7) return;
        for (int s = 0; s < nsteps; ++s) {
            int yoff[2], xoff[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int pix = 4 * (4 * s + 2 * kblk + r) + p4;  // this lane's pixel of the strip (row-major over TH x Wo)
                const int hol = pix / a.Wo, wo = pix - hol * a.Wo;
                yoff[r] = pix * 32 + ychunk;
                xoff[r] = ((2 * hol + kh) * a.PW + 2 * wo + xtap) * 4;
            }
            swg_bf16x8 af[3], bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                af[pl] = swg_join(swg_tr_read(ys + pl * yplane + yoff[0]), swg_tr_read(ys + pl * yplane + yoff[1]));
                bf[pl] = swg_join(swg_tr_read(xs + pl * xplane + xoff[0]), swg_tr_read(xs + pl * xplane + xoff[1]));
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest partial product first
#pragma unroll
            for (int q6 = 0; q6 < 6; ++q6) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[q6]], bf[PB[q6]], acc, 0, 0, 0);
        }
    };

    if (first < last) fetch(first);
    for (int item = first; item < last; ++item) {
        stage();
        __syncthreads();
        if (item + 1 < last) fetch(item + 1);
        compute();
        __syncthreads();  // every wave is done with this strip's images
    }
    // ---- slab [kh][slot][co]: lane (slot n = lane & 31) holds co = (v & 3) + 8 (v >> 2) + 4 (lane >> 5) in acc[v] ----
    if (kh < 7) {
        float* o = a.slabs + ((size_t)blockIdx.x * 7 + kh) * 1024 + (size_t)(lane & 31) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 sv = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (flip) sv = -sv;
            *reinterpret_cast<f32x4*>(o + 8 * g) = sv;
        }
    }
}

// dW[co][ci][kh][kw] (OIHW, creal input channels) = sum over the slabs, in slab order, of slab[kh][kw * 4 + ci][co]
__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dw, int creal) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 32 * creal * 49;
    if (e >= total) return;
    const int kw = e % 7, kh = (e / 7) % 7, ci = (e / 49) % creal, co = e / (49 * creal);
    const float* p = slabs + ((size_t)kh * 32 + kw * 4 + ci) * 32 + co;
    float s = 0.f;
    for (int k = 0; k < nslabs; ++k) s += p[(size_t)k * 7 * 1024];
    dw[e] = s;
}

inline bool stem_wgrad_strip_covers(int H, int W, int C, int Cout, int KH, int KW, int stride, int pad) {
    if (!(KH == 7 && KW == 7 && stride == 2 && pad == 3 && C == 4 && Cout == 32) || H < 1 || W < 1) return false;
    const int Wo = (W + 6 - 7) / 2 + 1;
    if (Wo > 64 || (Wo & 15)) return false;  // a 16-pixel k-step lies inside one output row
    const int PW = (2 * Wo + 6 + 1) & ~1;
    if (SWG_XROWS * PW > SWG_XPT * SWG_NT || SWG_TH * Wo * 8 > SWG_YPT * SWG_NT) return false;
    return (size_t)3 * SWG_XROWS * PW * 8 + (size_t)3 * SWG_TH * Wo * 64 <= 160 * 1024;
}

// 1: geometry not covered (or the scratch is too small).  ws: >= 256 * 7 * 1024 floats.
inline int stem_wgrad_strip(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int creal, float* ws, size_t ws_floats,
                            hipStream_t stream, const float* norm = nullptr) {
    if (!x || !dy || !dw_oihw || B <= 0 || creal < 1 || creal > 4 || (reinterpret_cast<uintptr_t>(norm) & 15)) return HAB_ERR_ARG;
    if (!stem_wgrad_strip_covers(H, W, 4, 32, 7, 7, 2, 3) || !ws) return 1;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ws)) & 15) return 1;
    StemWgArgs a;
    a.x = x; a.dy = dy; a.slabs = ws; a.B = B; a.H = H; a.W = W; a.norm = norm;
    a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1;
    a.PW = (2 * a.Wo + 6 + 1) & ~1;
    a.strips = (a.Ho + SWG_TH - 1) / SWG_TH;
    if ((long long)B * a.strips > 0x7fffffffLL) return 1;
    a.items = B * a.strips;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    int grid = 256;
    while (grid > 8 && grid > a.items) grid -= 8;
    if ((size_t)grid * 7 * 1024 > ws_floats) return 1;
    const size_t lds = (size_t)3 * SWG_XROWS * a.PW * 8 + (size_t)3 * SWG_TH * a.Wo * 64;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_strip_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    stem_wgrad_strip_kernel<<<grid, SWG_NT, lds, stream>>>(a);
    HAB_LAUNCH_CHECK();
    const int total = 32 * creal * 49;
    stem_wgrad_reduce_kernel<<<(total + 255) / 256, 256, 0, stream>>>(ws, grid, dw_oihw, creal);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
