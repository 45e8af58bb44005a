// obs_conv_bf3.h -- SimpleCNN's first convolution with the observation ingest fused (simple_cnn.py:139-156,68-74) on the bf16
// matrix pipe, fp32-equivalent arithmetic.  The dominant call site of the C2 step.
//
// Quad fast path of ObsConvFwdProb only (RGB-D, C == 4, KW % 4 == 0, stride % 4 == 0, pad == 0, W % 4 == 0) with K % 64 == 0.
//
// What makes this contraction special: the rgb operand is a uint8.  Every integer 0..255 is EXACTLY a bf16 (8 significant bits),
// so the A operand of the 48 rgb reduction elements of a 64-element tile needs ONE bf16 plane and no split at all -- the
// 1/255 of `x / 255.0` sits on the weight side as in the fp32 kernel (problems.h, obs_quad_cvt_raw).  Only the 16 depth elements
// (fp32 in [0,1]) take the exact three-term split of igemm_bf3.h.  The weights (N x K, a few thousand elements) are split once per
// call by a tiny kernel into three bf16 planes laid out in the tile order below; the contraction kernel copies them.
//
//      per 64-element tile and 32x32 output tile:   rgb  3 k-groups x 3 products (u8 x {w1,w2,w3})        9 MFMAs
//                                                   dep  1 k-group  x 6 products (igemm_bf3.h)             6 MFMAs
//      = 15 v_mfma_f32_32x32x16_bf16 (480 matrix-pipe cycles) against 32 v_mfma_f32_32x32x2_f32 (2048 issue cycles), and ~2.5 VALU
//      instructions per gathered element instead of ~7.
//
// Order of the reduction inside a tile (free, as long as A and B agree): gather unit u (4 taps x rgbd), tap q, channel c
//      rgb position 12 u + 3 q + c   (c < 3)            depth position 4 u + q
// LDS row pitches 112 B (rgb) / 48 B (depth): 16-byte fragment reads of 16 consecutive rows fall on 16 distinct 4-bank windows.
#pragma once
#include "igemm_bf3.h"

namespace hab {

constexpr int OBF_BK = 64, OBF_RGB = 48, OBF_DEP = 16;
constexpr int OBF_RGBP = 56, OBF_DEPP = 24;  // LDS row pitches in bf16 elements

// weight planes (bf16 bit patterns) in the workspace:  rgb [3][KT][NP][48], then dep [3][KT][NP][16]; a second set of the same size
// holds the planes of -w (sign schedule of igemm_bf3.h: odd output tiles accumulate the negated sum)
__global__ void obs_conv_bf3_split_weights(const float* __restrict__ w, int N, int K, int NP, unsigned short* __restrict__ planes_both) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * NP * K) return;
    const bool negset = e >= NP * K;
    if (negset) e -= NP * K;
    unsigned short* planes = planes_both + (negset ? (size_t)3 * NP * K : 0);
    const int n = e / K, k = e - n * K;
    const int kt = k >> 6, kk = k & 63, u = kk >> 4, q = (kk >> 2) & 3, c = kk & 3;
    const int KT = K >> 6;
    float v = n < N ? w[(size_t)n * K + k] : 0.f;
    if (c < 3) v *= HAB_RCP255;
    if (negset) v = -v;
    unsigned h1, h2, h3;
    bf3_split(v, h1, h2, h3);
    const size_t rgb_plane = (size_t)KT * NP * OBF_RGB, dep_plane = (size_t)KT * NP * OBF_DEP;
    if (c < 3) {
        const size_t o = ((size_t)kt * NP + n) * OBF_RGB + 12 * u + 3 * q + c;
        planes[o] = (unsigned short)h1;
        planes[rgb_plane + o] = (unsigned short)h2;
        planes[2 * rgb_plane + o] = (unsigned short)h3;
    } else {
        unsigned short* d = planes + 3 * rgb_plane;
        const size_t o = ((size_t)kt * NP + n) * OBF_DEP + 4 * u + q;
        d[o] = (unsigned short)h1;
        d[dep_plane + o] = (unsigned short)h2;
        d[2 * dep_plane + o] = (unsigned short)h3;
    }
}

template <int TM>
struct ObsBf3Cfg {
    static constexpr int NT = 256, BM = 4 * TM * 32, BN = 32;
    static constexpr int A_RGB = BM * OBF_RGBP, A_DEP = BM * OBF_DEPP, B_RGB = BN * OBF_RGBP, B_DEP = BN * OBF_DEPP;  // one plane
    static constexpr size_t LDS_BYTES = (size_t)(A_RGB + 3 * A_DEP + 3 * B_RGB + 3 * B_DEP) * 2;
    static constexpr int A_UNITS = BM * 4 / NT;
};

template <int TM>
__global__ void __launch_bounds__(256) obs_conv_bf3_kernel(const ObsConvFwdProb p, const unsigned short* __restrict__ planes, const int NP) {
    using P = ObsConvFwdProb;
    using Cfg = ObsBf3Cfg<TM>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, A_UNITS = Cfg::A_UNITS;
    static_assert(EpiV4<P>::value, "transposed-accumulator epilogue");
    static_assert(A_UNITS * 64 == BM, "unit j of a thread is row (t >> 2) + 64 j");

    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* Argb = smem16;
    unsigned short* Adep = Argb + Cfg::A_RGB;
    unsigned short* Brgb = Adep + 3 * Cfg::A_DEP;
    unsigned short* Bdep = Brgb + 3 * Cfg::B_RGB;

    const int t = threadIdx.x;
    const int lane = t & 63, wm = t >> 6;
    const int li = lane & 31, hi = lane >> 5;

    const int nt_m = cdiv(p.M, BM), nt_n = cdiv(p.N, 32);
    const int ntiles = nt_m * nt_n;
    const int KT = p.K >> 6;
    // Persistent workgroup: virtual block ids b, b + G, b + 2G, ... each mapped to a tile as a full-grid launch would (XCD-contiguous
    // runs of M-tiles).  K is short (4 tiles for the 8x8 RGB-D filter), so the loop over (tile, k-tile) is flattened and the gather of
    // the NEXT tile's first k-tile is in flight across the current tile's last MFMAs and its epilogue.
    auto tile_of = [&](int vb) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7, idx = vb >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    const int vgrid = ntiles;  // gridDim.x is a multiple of 8: a workgroup stays on its XCD's run

    // Gather addressing of the quad fast path: pad == 0, so every tap of a valid output pixel is inside the image; rows past M are
    // clamped to the last row (their accumulators are never stored).  Unit j of a thread = output pixel m0 + (t >> 2) + 64 j, the four
    // horizontally adjacent taps (t & 3) of the k-tile: 12 rgb bytes (4-byte aligned) + 4 depth floats (16-byte aligned).
    const uint8_t* rgb_ptr[A_UNITS];
    const float* dep_ptr[A_UNITS];
    // B copy: 16-byte chunk `t` of each of the three planes: rows of 6 rgb chunks (t < 192), rows of 2 depth chunks (t >= 192)
    const size_t rgb_plane = (size_t)KT * NP * OBF_RGB, dep_plane = (size_t)KT * NP * OBF_DEP;
    const bool b_is_rgb = t < 192;
    const int b_row = b_is_rgb ? t / 6 : (t - 192) >> 1, b_ch = b_is_rgb ? t % 6 : (t - 192) & 1;
    const unsigned short* b_src0 = b_is_rgb ? planes + (size_t)b_row * OBF_RGB + b_ch * 8 : planes + 3 * rgb_plane + (size_t)b_row * OBF_DEP + b_ch * 8;
    const int b_row_elems = b_is_rgb ? OBF_RGB : OBF_DEP;
    const size_t b_tile_stride = (size_t)NP * b_row_elems, b_plane_stride = b_is_rgb ? rgb_plane : dep_plane;
    unsigned short* b_dst = b_is_rgb ? Brgb + b_row * OBF_RGBP + b_ch * 8 : Bdep + b_row * OBF_DEPP + b_ch * 8;
    const int b_dst_plane = b_is_rgb ? Cfg::B_RGB : Cfg::B_DEP;

    struct Rgb12 { uint32_t d0, d1, d2; };
    Rgb12 a_rgb[A_UNITS];
    f32x4 a_dep[A_UNITS];
    u32x4 braw[3];
    auto setup = [&](int tile, int& m0, int& n0) {
        const int tile_n = tile % nt_n, tile_m = tile / nt_n;
        m0 = tile_m * BM; n0 = tile_n * 32;
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            int m = m0 + (t >> 2) + 64 * j;
            m = m < p.M ? m : p.M - 1;
            int img, rem, ho, wo;
            p.g.dHoWo.divmod(m, img, rem);
            p.g.dWo.divmod(rem, ho, wo);
            const size_t pix = ((size_t)p.obs.srow(img) * p.g.H + ho * p.g.stride) * p.g.W + wo * p.g.stride;
            rgb_ptr[j] = p.obs.rgb + pix * 3;
            dep_ptr[j] = p.obs.depth + pix;
        }
    };
    // sign schedule (igemm_bf3.h, short reduction): odd output tiles accumulate the negated sum (planes of -w), negated before the epilogue
    const size_t neg_set = (size_t)3 * NP * p.K;
    auto fetch = [&](int kt, int n0, bool neg) {
        int kh, kw;
        p.g.dKW.divmod(kt * 16 + (t & 3) * 4, kh, kw);  // first of the unit's four taps
        const int off = kh * p.g.W + kw;
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            a_rgb[j] = *reinterpret_cast<const Rgb12*>(rgb_ptr[j] + off * 3);
            a_dep[j] = ld4(dep_ptr[j] + off);
        }
        const unsigned short* src = b_src0 + (size_t)n0 * b_row_elems + kt * b_tile_stride + (neg ? neg_set : 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) braw[pl] = *reinterpret_cast<const u32x4*>(src + pl * b_plane_stride);
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < A_UNITS; ++j) {
            const int row = (t >> 2) + 64 * j, unit = t & 3;
            const unsigned d[3] = {a_rgb[j].d0, a_rgb[j].d1, a_rgb[j].d2};
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
            bf3_store4(a_dep[j], dd, dd + Cfg::A_DEP, dd + 2 * Cfg::A_DEP);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(b_dst + pl * b_dst_plane) = braw[pl];
    };

    int vb = blockIdx.x;
    if (vb >= vgrid) return;
    int m0, n0;
    setup(tile_of(vb), m0, n0);
    bool neg = ((tile_of(vb) / nt_n + tile_of(vb) % nt_n) & 1) != 0;
    fetch(0, n0, neg);
    for (;;) {
        f32x16 acc[TM][1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][0][v] = 0.0f;
        const int m0_cur = m0, n0_cur = n0;
        const bool neg_cur = neg;
        bool more = false;
        for (int kt = 0; kt < KT; ++kt) {
            stage();
            __syncthreads();
            if (kt + 1 < KT) {
                fetch(kt + 1, n0_cur, neg_cur);
            } else {
                vb += gridDim.x;
                more = vb < vgrid;
                if (more) {
                    setup(tile_of(vb), m0, n0);
                    neg = ((tile_of(vb) / nt_n + tile_of(vb) % nt_n) & 1) != 0;
                    fetch(0, n0, neg);
                }
            }
            // rgb: three k-groups, A exact in one plane
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                bf16x8 a[TM], b[3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const bf16x8*>(Argb + ((wm * TM + i) * 32 + li) * OBF_RGBP + g * 16 + hi * 8);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(Brgb + pl * Cfg::B_RGB + li * OBF_RGBP + g * 16 + hi * 8);
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[pl], a[i], acc[i][0], 0, 0, 0);
            }
            // depth: one k-group, both operands split
            {
                bf16x8 a[TM][3], b[3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[i][pl] = *reinterpret_cast<const bf16x8*>(Adep + pl * Cfg::A_DEP + ((wm * TM + i) * 32 + li) * OBF_DEPP + hi * 8);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(Bdep + pl * Cfg::B_DEP + li * OBF_DEPP + hi * 8);
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB[q]], a[i][PA[q]], acc[i][0], 0, 0, 0);
            }
            __syncthreads();
        }
        if (neg_cur) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][0][v] = -acc[i][0][v];
        }
        igemm_epilogue_v4<P, TM, 1>(p, acc, m0_cur + wm * TM * 32, n0_cur, li, hi);
        if (!more) break;
    }
}

// returns HAB_OK, an error, or 1 when the problem / workspace does not fit this path (caller falls back to the generic kernels)
template <int TM>
inline int obs_conv_bf3_launch(const ObsConvFwdProb& p, float* ws, size_t ws_floats, hipStream_t stream) {
    using Cfg = ObsBf3Cfg<TM>;
    if (!p.quad || (p.K & 63) || p.M <= 0 || p.N <= 0) return 1;
    const int KT = p.K >> 6, NP = cdiv(p.N, 32) * 32;
    const size_t plane_bytes = (size_t)2 * 3 * KT * NP * OBF_BK * 2;  // +w and -w sets
    if (!ws || ws_floats * 4 < plane_bytes || (reinterpret_cast<uintptr_t>(ws) & 15)) return 1;
    unsigned short* planes = reinterpret_cast<unsigned short*>(ws);
    obs_conv_bf3_split_weights<<<cdiv(2 * NP * p.K, 256), 256, 0, stream>>>(p.w, p.N, p.K, NP, planes);
    HAB_LAUNCH_CHECK();
    auto kern = obs_conv_bf3_kernel<TM>;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (attr_err != hipSuccess) return (int)attr_err;
    constexpr int wg_per_cu = 2;
    const int ntiles = cdiv(p.M, Cfg::BM) * cdiv(p.N, 32);
    const int grid = ntiles < 256 * wg_per_cu ? ntiles : 256 * wg_per_cu;
    kern<<<(grid + 7) / 8 * 8, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, planes, NP);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
