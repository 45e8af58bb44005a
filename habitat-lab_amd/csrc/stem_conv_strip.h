// stem_conv_strip.h -- forward of the GroupNorm-ResNet stem convolution (7x7 / stride 2 / padding 3, 4 input channels -> 32;
// habitat_baselines/rl/ddppo/policy/resnet.py:207-219 `conv1`, input = the 2x2-averaged RGB-D observation of
// resnet_policy.py:259-272) with the INPUT STRIP resident in LDS.
//
// Why: as an im2col contraction (igemm_bf3.h) the layer has K = 7 * 7 * 4 = 196, N = 32: every input element is gathered and split
// 49 / 4 = 12x, 4 channels (16 bytes) per gather; measured 3.0 ms per 4096 frames (70 TFLOP/s-eq, 0.17 of the split-bf16 ceiling) and
// 76 us per 64-frame rollout step -- the largest single kernel of the ResNet rollout and 6 % of a learner minibatch.  The roofline of
// the layer is HBM: 256 KB in + 512 KB out per frame = 0.1 us; the matrix pipe needs 0.12 us.
// Here:
//   * a workgroup owns a strip of 8 output rows x the full width of one frame: the 21 input rows it reads (+ 3 zero columns either
//     side) are loaded once with 16-byte loads (one pixel = 4 channels), split once (exact 3-term split, igemm_bf3.h) and kept as three
//     bf16 planes [row][column][4 channels], 8 bytes per pixel;
//   * the reduction is ordered (kh, kw, ci): the 7 taps x 4 channels of one filter row are 28 CONTIGUOUS elements of the image starting
//     at column 2 wo -- padded to 32 with zero weights they are exactly two k-steps of 16, so an MFMA fragment is one aligned 16-byte
//     LDS read (2 pixels) and 16 consecutive output pixels read 256 contiguous bytes: conflict-free, no index arithmetic per tap;
//   * the filter (three bf16 planes in fragment order, split once per optimiser step: stem_split_weights) sits in LDS beside the strip
//     (42 KB): one conflict-free 16-byte read per plane and k-step;
//   * wave w owns output row w of the strip: two 32-pixel tiles x 32 channels, 14 k-steps x 12 MFMAs, no reduction across waves; the
//     operands are swapped so that a lane ends with 4 consecutive channels of one pixel -> 16-byte stores.
//   * round 4, second half: the RunningMeanAndVar normalisation of the training forward is applied while the strip is staged
//     (StemArgs::norm: one fma per element on the pixels inside the image -- the 1 GB normalised copy of the observation tensor is never
//     written), and the GroupNorm that follows gets its per-strip partial statistics from the epilogue (StemArgs::part, template CPG).
// Sign schedule as everywhere on the split path: every second workgroup accumulates the negated sum.
#pragma once
#include "bf3_split.h"

namespace hab {

typedef __bf16 stem_bf16x8 __attribute__((ext_vector_type(8)));

struct StemArgs {
    const float* x;            // [B][H][W][4]
    const unsigned short* wq;  // [3 planes][14 k-steps][64 lanes][8] bf16 (stem_split_weights)
    float* y;                  // [B][Ho][Wo][32]
    int B, H, W, Ho, Wo, PW;   // PW: pixels per LDS image row (even, >= 2 Wo + 6)
    int strips;                // ceil(Ho / 8)
    int sign_schedule;
    float* part;               // CPG > 0: GroupNorm partial statistics [B][strips][32 / CPG][2] = (mean, M2) of the strip's outputs per group
    const float* norm;         // null, or the RunningMeanAndVar affine of the 4 input channels applied while staging: fma(x, norm[c],
                               // norm[8 + c]) on the pixels inside the image (rmv_normalize_kernel's arithmetic; the zero padding stays zero)
};

constexpr int STEM_TH = 8, STEM_ROWS = 2 * STEM_TH + 5, STEM_KS = 14;
constexpr size_t STEM_W_BYTES = (size_t)3 * STEM_KS * 1024;

// CPG: 0, or the channels per GroupNorm group (1, 2, 4) of the GroupNorm that follows -- the strip's per-group (mean, M2) then leave
// with the outputs (the chunk statistics gn_chunk_stats_kernel would compute in a separate pass over the 2 KB-per-pixel-row tensor:
// exact local two-pass on the accumulators, merged later by Chan's formula, resnet_ops.hip).
constexpr int STEM_STAT_FLOATS = 8 * 32 + 32;
template <int CPG>
__global__ void __launch_bounds__(512) stem_conv_strip_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short stem_sm[];
    unsigned short* ws = stem_sm;                                   // [3][14][64][8]
    unsigned short* xs = stem_sm + STEM_W_BYTES / 2;                // [3][STEM_ROWS][PW][4]   (+ STEM_STAT_FLOATS floats behind it)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int img = blockIdx.x / a.strips, strip = blockIdx.x - img * a.strips;
    const int ho0 = strip * STEM_TH;
    const bool flip = a.sign_schedule && (blockIdx.x & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;
    const int plane = STEM_ROWS * a.PW * 4;  // bf16 elements per plane of the strip image

    // ---- filter planes -> LDS (fragment order: a straight copy), sign applied here ----
    for (int u = t; u < (int)(STEM_W_BYTES / 16); u += 512) {
        u32x4 w = *reinterpret_cast<const u32x4*>(a.wq + (size_t)u * 8);
        *reinterpret_cast<u32x4*>(ws + (size_t)u * 8) = u32x4{w[0] ^ sgn2, w[1] ^ sgn2, w[2] ^ sgn2, w[3] ^ sgn2};
    }
    // ---- input strip -> three bf16 planes; image row r = input row 2 ho0 - 3 + r, image column c = input column c - 3 ----
    {
        const int units = STEM_ROWS * a.PW;
        const float* xb = a.x + (size_t)img * a.H * a.W * 4;
        f32x4 nb = {1.f, 1.f, 1.f, 1.f}, na = {0.f, 0.f, 0.f, 0.f};
        if (a.norm) { nb = *reinterpret_cast<const f32x4*>(a.norm); na = *reinterpret_cast<const f32x4*>(a.norm + 8); }
        constexpr int UB = 6;
        for (int u0 = t; u0 < units; u0 += 512 * UB) {
            f32x4 v[UB];
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                const int u = u0 + j * 512;
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (u < units) {
                    const int r = u / a.PW, c = u - r * a.PW;
                    const int hin = 2 * ho0 - 3 + r, win = c - 3;
                    if (((unsigned)hin < (unsigned)a.H) & ((unsigned)win < (unsigned)a.W))
                        v[j] = *reinterpret_cast<const f32x4*>(xb + ((size_t)hin * a.W + win) * 4);
                }
            }
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                const int u = u0 + j * 512;
                if (u < units) {
                    if (a.norm) {  // (after the batch of loads, so that they stay in flight together)
                        const int r = u / a.PW, c = u - r * a.PW;
                        const int hin = 2 * ho0 - 3 + r, win = c - 3;
                        if (((unsigned)hin < (unsigned)a.H) & ((unsigned)win < (unsigned)a.W)) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[j][q] = __builtin_fmaf(v[j][q], nb[q], na[q]);
                        }
                    }
                    unsigned a1, a2, a3, b1, b2, b3;
                    bf3_split2(v[j][0], v[j][1], a1, a2, a3);
                    bf3_split2(v[j][2], v[j][3], b1, b2, b3);
                    unsigned short* dst = xs + (size_t)u * 4;
                    *reinterpret_cast<u32x2*>(dst) = u32x2{a1, b1};
                    *reinterpret_cast<u32x2*>(dst + plane) = u32x2{a2, b2};
                    *reinterpret_cast<u32x2*>(dst + 2 * plane) = u32x2{a3, b3};
                }
            }
        }
    }
    __syncthreads();

    // ---- wave w: output row ho0 + w, pixel tiles wo = 32 i + li ----
    const int ho = ho0 + wave;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const int ntile = a.Wo > 32 ? 2 : 1;
    if (ho < a.Ho) {
        // element offset of this lane's fragment inside an image row: 2 pixels starting at column 2 wo + 4 j + 2 hi
        int coff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) coff[i] = (2 * min(32 * i + li, a.Wo - 1)) * 4 + 8 * hi;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const unsigned short* xrow = xs + (size_t)(2 * wave + kh) * a.PW * 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                stem_bf16x8 bw[3], af[2][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bw[pl] = *reinterpret_cast<const stem_bf16x8*>(ws + ((size_t)(pl * STEM_KS + kh * 2 + j) * 64 + lane) * 8);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        af[i][pl] = *reinterpret_cast<const stem_bf16x8*>(xrow + coff[i] + 16 * j + pl * plane);
                constexpr int PX[6] = {2, 0, 1, 1, 0, 0}, PW_[6] = {0, 2, 1, 0, 1, 0};  // smallest partial product first
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[PW_[q]], af[0][PX[q]], acc[0], 0, 0, 0);
                    // (also when Wo <= 32: the second tile then recomputes the clamped last pixel and is dropped -- a branch here
                    //  makes the compiler shuffle the accumulator registers around every MFMA)
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[PW_[q]], af[1][PX[q]], acc[1], 0, 0, 0);
                }
            }
        }
        // lane (pixel li of tile i): channels 8 g + 4 hi .. +3 in acc[i][4 g .. 4 g + 3]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int wo = 32 * i + li;
            if (i < ntile && wo < a.Wo) {
                float* o = a.y + (((size_t)img * a.Ho + ho) * a.Wo + wo) * 32 + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 s = f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                    if (flip) s = -s;
                    *reinterpret_cast<f32x4*>(o + 8 * g) = s;
                }
            }
        }
    }
    if constexpr (CPG > 0) {
        // lane value j: quad g = j / QG (channels 8 g + 4 hi .. + 3), sub-group j % QG inside the quad -> group (8 g + 4 hi) / CPG + j % QG
        constexpr int QG = 4 / CPG, NJ = 4 * QG, G = 32 / CPG;
        float* red = reinterpret_cast<float*>(xs + (size_t)3 * plane);  // [8 waves][32]
        float* mu = red + 8 * 32;
        const float sg = flip ? -1.f : 1.f;
        bool ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) ok[i] = (ho < a.Ho) & (i < ntile) & (32 * i + li < a.Wo);
        float sj[NJ];
        auto wave_total = [&](float* v) {  // over the 32 pixels of each half-wave (the halves hold different channels): fixed xor tree
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) v[j] += __shfl_xor(v[j], off);
            if (li == 0) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) red[wave * 32 + (8 * (j / QG) + 4 * hi) / CPG + j % QG] = v[j];
            }
        };
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            sj[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < CPG; ++k) sj[j] += ok[i] ? sg * acc[i][4 * (j / QG) + (j % QG) * CPG + k] : 0.f;
        }
        wave_total(sj);
        __syncthreads();
        const int rows = min(STEM_TH, a.Ho - ho0);
        if (t < G) {
            float tot = 0.f;
            for (int w = 0; w < STEM_TH; ++w) tot += red[w * 32 + t];
            mu[t] = tot / (float)(rows * a.Wo * CPG);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float m = mu[(8 * (j / QG) + 4 * hi) / CPG + j % QG];
            sj[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < CPG; ++k) {
                    const float d = sg * acc[i][4 * (j / QG) + (j % QG) * CPG + k] - m;
                    sj[j] += ok[i] ? d * d : 0.f;
                }
        }
        wave_total(sj);
        __syncthreads();
        if (t < G) {
            float m2 = 0.f;
            for (int w = 0; w < STEM_TH; ++w) m2 += red[w * 32 + t];
            float* o = a.part + ((size_t)blockIdx.x * G + t) * 2;
            o[0] = mu[t];
            o[1] = m2;
        }
    }
}

// Forward-packed stem weight wf [32][7][7][4] (cpad = 4) -> fragment-ordered planes [3][14][64][8]:
//   k-step s = 2 kh + j, lane = 32 hi + co, element e: reduction slot 16 j + 8 hi + e of filter row kh = (kw, ci) = (slot / 4, slot % 4), zero for kw = 7
__global__ void stem_split_weights_kernel(const float* __restrict__ wf, unsigned* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;  // pair index: ((s * 64 + lane) * 4 + ep)
    constexpr int NP = STEM_KS * 64 * 4;
    if (q >= NP) return;
    const int ep = q & 3, lane = (q >> 2) & 63, s = q >> 8;
    const int kh = s >> 1, j = s & 1, co = lane & 31, hi = lane >> 5;
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int slot = 16 * j + 8 * hi + 2 * ep + e, kw = slot >> 2, ci = slot & 3;
        v[e] = kw < 7 ? wf[((size_t)(co * 7 + kh) * 7 + kw) * 4 + ci] : 0.f;
    }
    unsigned p0, p1, p2;
    bf3_split2(v[0], v[1], p0, p1, p2);
    out[q] = p0; out[NP + q] = p1; out[2 * NP + q] = p2;
}

inline int stem_split_weights(const float* wf, unsigned short* planes, hipStream_t stream) {
    if (!wf || !planes) return HAB_ERR_ARG;
    stem_split_weights_kernel<<<(STEM_KS * 64 * 4 + 255) / 256, 256, 0, stream>>>(wf, reinterpret_cast<unsigned*>(planes));
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

inline bool stem_conv_strip_covers(int H, int W, int C, int Cout, int KH, int KW, int stride, int pad) {
    if (!(KH == 7 && KW == 7 && stride == 2 && pad == 3 && C == 4 && Cout == 32) || H < 1 || W < 1) return false;
    const int Wo = (W + 6 - 7) / 2 + 1;
    if (Wo > 64) return false;
    const int PW = (2 * Wo + 6 + 1) & ~1;
    return STEM_W_BYTES + (size_t)3 * STEM_ROWS * PW * 8 + STEM_STAT_FLOATS * sizeof(float) <= 160 * 1024;
}

// 1: geometry not covered.
// part / groups: GroupNorm partial statistics of the output per (frame, strip of 8 rows, group), see the kernel; groups in {8, 16, 32}
inline int stem_conv_strip(const float* x, const unsigned short* wq, float* y, int B, int H, int W, hipStream_t stream,
                           const float* norm = nullptr, float* part = nullptr, int groups = 0) {
    if (!x || !wq || !y || B <= 0 || (reinterpret_cast<uintptr_t>(norm) & 15)) return HAB_ERR_ARG;
    if (part && groups != 8 && groups != 16 && groups != 32) return HAB_ERR_ARG;
    if (!stem_conv_strip_covers(H, W, 4, 32, 7, 7, 2, 3)) return 1;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(y)) & 15) return 1;
    StemArgs a;
    a.x = x; a.wq = wq; a.y = y; a.B = B; a.H = H; a.W = W; a.norm = norm; a.part = part;
    a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1;
    a.PW = (2 * a.Wo + 6 + 1) & ~1;
    a.strips = (a.Ho + STEM_TH - 1) / STEM_TH;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    const size_t lds = STEM_W_BYTES + (size_t)3 * STEM_ROWS * a.PW * 8 + STEM_STAT_FLOATS * sizeof(float);
    static const hipError_t attr_err = [] {
        hipError_t e = hipSuccess;
        const void* ks[4] = {reinterpret_cast<const void*>(stem_conv_strip_kernel<0>), reinterpret_cast<const void*>(stem_conv_strip_kernel<1>),
                             reinterpret_cast<const void*>(stem_conv_strip_kernel<2>), reinterpret_cast<const void*>(stem_conv_strip_kernel<4>)};
        for (const void* k : ks) {
            const hipError_t r = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (r != hipSuccess) e = r;
        }
        return e;
    }();
    if (attr_err != hipSuccess) return (int)attr_err;
    if ((long long)B * a.strips > 0x7fffffffLL) return 1;
    const int cpg = part ? 32 / groups : 0;
    if (cpg == 0) stem_conv_strip_kernel<0><<<B * a.strips, 512, lds, stream>>>(a);
    else if (cpg == 1) stem_conv_strip_kernel<1><<<B * a.strips, 512, lds, stream>>>(a);
    else if (cpg == 2) stem_conv_strip_kernel<2><<<B * a.strips, 512, lds, stream>>>(a);
    else stem_conv_strip_kernel<4><<<B * a.strips, 512, lds, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
