// conv3x3_strip.h -- 3x3 / stride 1 / padding 1 convolution with 32 input and 32 output channels (ResNet layer1:
// habitat_baselines/rl/ddppo/policy/resnet.py:19-34 conv3x3 inside BasicBlock :37-69), forward AND data gradient, with the INPUT
// STRIP resident in LDS and TWO pixel tiles per wave.
//
// Why: conv_patch_bf3.h keeps the input patch in LDS too, but a wave there owns ONE 32-pixel tile: per 6 MFMAs it reads 3 activation
// and 3 weight fragments from LDS (1 KB per MFMA); with 8 waves per CU that is 128 bytes per clock -- the LDS peak: the N = 32 layers
// run at 113 TFLOP/s-eq (0.27 of the split ceiling) with the matrix pipe half idle, 0.62-0.70 ms per layer1 convolution at 4096 frames,
// eight of them per minibatch.  Here:
//   * a workgroup owns a strip of 16 output rows x the full width (<= 32) of one frame: the 18 input rows (+ a zero column either
//     side) are read once with 16-byte loads, split once (exact 3-term split, igemm_bf3.h) into three bf16 planes [row][column][32 ch],
//     80-byte pixel pitch (the 16-byte fragment reads of 16 consecutive pixels hit all 64 banks);
//   * wave w owns output rows 2 w and 2 w + 1: two 32 x 32 accumulators that share every weight fragment -- 6 activation reads per 12
//     MFMAs, half the LDS traffic per MFMA; the MFMAs of the two tiles alternate (two short rounding chains, as in conv_patch_bf3.h);
//   * the filter never enters LDS: its three bf16 planes in MFMA fragment order (split once per optimiser step, conv3x3_strip_weights;
//     the data gradient uses the taps flipped and the channel roles swapped) are read from L1 / L2, 1 KB contiguous per fragment,
//     one k-step ahead in registers;
//   * operands swapped so that a lane ends with 4 consecutive channels of one pixel: 16-byte stores, 16-byte mask / residual loads;
//   * persistent workgroups, the next strip's global loads in flight during the MFMAs.
// Data-gradient epilogue as in problems.h: dx = (acc + add) * (mask > 0).  Sign schedule: every second workgroup negated.
#pragma once
#include "bf3_split.h"

namespace hab {

typedef __bf16 c3s_bf16x8 __attribute__((ext_vector_type(8)));

struct C3sArgs {
    const float* x;            // [B][H][W][32]
    const unsigned short* wq;  // [3 planes][18 k-steps][64 lanes][8] bf16 (conv3x3_strip_weights)
    const float* mask;         // optional [B][H][W][32]: y = mask > 0 ? y : 0   (ReLU mask of a data gradient)
    const float* add;          // optional [B][H][W][32]: y += add, before the mask  (gradient of the residual branch)
    float* y;                  // [B][H][W][32]
    int B, H, W;
    int strips, items;
    int sign_schedule;
};

constexpr int C3S_TH = 16, C3S_ROWS = C3S_TH + 2, C3S_PW = 34, C3S_PIX = 40, C3S_NT = 512;
constexpr int C3S_PLANE = C3S_ROWS * C3S_PW * C3S_PIX;                    // bf16 elements per plane
constexpr size_t C3S_LDS_BYTES = (size_t)3 * C3S_PLANE * 2;               // 146 880
constexpr int C3S_XPT = (C3S_ROWS * 32 * 8 + C3S_NT - 1) / C3S_NT;        // 16-byte units of a strip per thread (W = 32): 9

__global__ void __launch_bounds__(C3S_NT) conv3x3_strip_kernel(const C3sArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short c3s_sm[];
    unsigned short* xs = c3s_sm;  // [3][18][34][40]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, hi = lane >> 5;

    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int per_xcd = (a.items + 7) >> 3, per_wg = (per_xcd + wg_per_xcd - 1) / wg_per_xcd;
    const int xcd_end = min(a.items, (xcd + 1) * per_xcd);
    const int first = min(xcd_end, xcd * per_xcd + jw * per_wg), last = min(xcd_end, first + per_wg);
    if (first >= last) return;

    const bool flip = a.sign_schedule && (blockIdx.x & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;

    // zero columns (input columns -1 and >= W) of every image row: written once, never touched by stage()
    for (int e = t; e < 3 * C3S_ROWS * C3S_PW * 4; e += C3S_NT) {
        const int q = e & 3, px = e >> 2;  // 16-byte chunk q (of the first 4 = 32 channels) of image pixel px
        const int c = px % C3S_PW;
        if (c == 0 || c > a.W) *reinterpret_cast<u32x4*>(xs + (size_t)px * C3S_PIX + q * 8) = u32x4{0u, 0u, 0u, 0u};
    }

    const int units = C3S_ROWS * a.W * 8;  // 16-byte units of a strip: (row, column, channel quad)
    f32x4 xr[C3S_XPT];
    int pf_ho0 = 0;
    auto fetch = [&](int item) {  // issues the loads only (rows outside the image read the frame's first bytes; zeroed in stage)
        const int img = item / a.strips, ho0 = (item - img * a.strips) * C3S_TH;
        pf_ho0 = ho0;
        const float* xb = a.x + (size_t)img * a.H * a.W * 32;
#pragma unroll
        for (int j = 0; j < C3S_XPT; ++j) {
            const int u = t + j * C3S_NT;
            const int r = u / (a.W * 8);
            const int hin = ho0 - 1 + r;
            const bool ok = (u < units) & ((unsigned)hin < (unsigned)a.H);
            xr[j] = *reinterpret_cast<const f32x4*>(xb + (ok ? ((size_t)hin * a.W * 32 + (size_t)(u - r * a.W * 8) * 4) : 0));
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < C3S_XPT; ++j) {
            const int u = t + j * C3S_NT;
            if (u >= units) continue;
            const int r = u / (a.W * 8), rem = u - r * a.W * 8, c = rem >> 3, c4 = rem & 7;
            const bool ok = (unsigned)(pf_ho0 - 1 + r) < (unsigned)a.H;
            const f32x4 v = ok ? xr[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned a1, a2, a3, b1, b2, b3;
            bf3_split2(v[0], v[1], a1, a2, a3);
            bf3_split2(v[2], v[3], b1, b2, b3);
            unsigned short* dst = xs + (size_t)(r * C3S_PW + c + 1) * C3S_PIX + c4 * 4;
            *reinterpret_cast<u32x2*>(dst) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(dst + C3S_PLANE) = u32x2{a2, b2};
            *reinterpret_cast<u32x2*>(dst + 2 * C3S_PLANE) = u32x2{a3, b3};
        }
    };

    // weight fragments: k-step s = tap * 2 + j, one 16-byte load per plane and lane
    const unsigned short* wl = a.wq + lane * 8;
    constexpr size_t WPL = (size_t)18 * 512;
    auto wload = [&](u32x4 (&w)[3], int s) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const u32x4*>(wl + pl * WPL + (size_t)s * 512);
    };
    const int wo = li < a.W ? li : a.W - 1;  // lanes beyond the width compute a duplicate and store nothing

    fetch(first);
    for (int item = first; item < last; ++item) {
        stage();
        __syncthreads();
        if (item + 1 < last) fetch(item + 1);
        const int img = item / a.strips, ho0 = (item - img * a.strips) * C3S_TH;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
        u32x4 wcur[3], wnxt[3];
        wload(wcur, 0);
        // image element offset of this lane's fragment for tap (0, 0), channels 0 .. 7 (+ 8 hi), output rows 2 wave + i
        const int base = ((2 * wave) * C3S_PW + wo) * C3S_PIX + 8 * hi;
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + 1 < 18) wload(wnxt, s + 1);
            const int tap = s >> 1, j = s & 1, kh = tap / 3, kw = tap - kh * 3;
            c3s_bf16x8 bw[3], af[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bw[pl] = __builtin_bit_cast(c3s_bf16x8, u32x4{wcur[pl][0] ^ sgn2, wcur[pl][1] ^ sgn2, wcur[pl][2] ^ sgn2, wcur[pl][3] ^ sgn2});
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    af[i][pl] = *reinterpret_cast<const c3s_bf16x8*>(xs + pl * C3S_PLANE + base + ((i + kh) * C3S_PW + kw) * C3S_PIX + 16 * j);
            constexpr int PX[6] = {2, 0, 1, 1, 0, 0}, PW_[6] = {0, 2, 1, 0, 1, 0};  // smallest partial product first
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[PW_[q]], af[0][PX[q]], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[PW_[q]], af[1][PX[q]], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wcur[pl] = wnxt[pl];
        }
        // lane (pixel li of row 2 wave + i): channels 8 g + 4 hi .. +3 in acc[i][4 g .. 4 g + 3]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ho = ho0 + 2 * wave + i;
            if (ho < a.H && li < a.W) {
                const size_t o = (((size_t)img * a.H + ho) * a.W + li) * 32 + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 sv = f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                    if (flip) sv = -sv;
                    if (a.add) sv += *reinterpret_cast<const f32x4*>(a.add + o + 8 * g);
                    if (a.mask) {
                        const f32x4 m = *reinterpret_cast<const f32x4*>(a.mask + o + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sv[e] = m[e] > 0.f ? sv[e] : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(a.y + o + 8 * g) = sv;
                }
            }
        }
        __syncthreads();  // every wave is done with this strip's image
    }
}

// Weight planes in fragment order from a packed 3x3 filter w [32 rows][(tap, 32 columns)] (forward: Wf[co][(kh, kw, ci)]; data
// gradient: Wd[ci][(kh, kw, co)] with flip = 1, which visits the taps in reverse -- dX = conv(dY, taps flipped)):
//   out[(p * 18 + tap' * 2 + j) * 512 + lane * 8 + e] = plane_p(w[lane & 31][tap][16 j + 8 (lane >> 5) + e]),  tap = flip ? 8 - tap' : tap'
__global__ void c3s_split_weights_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int flip) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;  // pair index ((s * 64 + lane) * 4 + ep)
    constexpr int NP = 18 * 64 * 4;
    if (q >= NP) return;
    const int ep = q & 3, lane = (q >> 2) & 63, s = q >> 8;
    const int tapp = s >> 1, j = s & 1, tap = flip ? 8 - tapp : tapp;
    const int row = lane & 31, col = 16 * j + 8 * (lane >> 5) + 2 * ep;
    const float* src = w + ((size_t)row * 9 + tap) * 32 + col;
    unsigned p0, p1, p2;
    bf3_split2(src[0], src[1], p0, p1, p2);
    out[q] = p0; out[NP + q] = p1; out[2 * NP + q] = p2;
}
constexpr int C3S_W_FLOATS = 3 * 18 * 512 / 2;  // one set of planes, in floats of the packed arena

inline int conv3x3_strip_weights(const float* w_packed, int flip, unsigned short* planes, hipStream_t stream) {
    if (!w_packed || !planes) return HAB_ERR_ARG;
    c3s_split_weights_kernel<<<(18 * 64 * 4 + 255) / 256, 256, 0, stream>>>(w_packed, reinterpret_cast<unsigned*>(planes), flip);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

inline bool conv3x3_strip_covers(int H, int W, int C, int Cout, int KH, int KW, int stride, int pad) {
    return KH == 3 && KW == 3 && stride == 1 && pad == 1 && C == 32 && Cout == 32 && W > 16 && W <= 32 && H >= 1;
}

// 1: geometry not covered.
inline int conv3x3_strip(const float* x, const unsigned short* wq, const float* mask, const float* add, float* y, int B, int H, int W,
                         hipStream_t stream) {
    if (!x || !wq || !y || B <= 0) return HAB_ERR_ARG;
    if (!conv3x3_strip_covers(H, W, 32, 32, 3, 3, 1, 1)) return 1;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(mask) |
         reinterpret_cast<uintptr_t>(add)) & 15)
        return 1;
    C3sArgs a;
    a.x = x; a.wq = wq; a.mask = mask; a.add = add; a.y = y; a.B = B; a.H = H; a.W = W;
    a.strips = (H + C3S_TH - 1) / C3S_TH;
    if ((long long)B * a.strips > 0x7fffffffLL) return 1;
    a.items = B * a.strips;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_strip_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    int grid = 256;
    while (grid > 8 && grid > a.items) grid -= 8;
    conv3x3_strip_kernel<<<grid, C3S_NT, C3S_LDS_BYTES, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
