// obs_wgrad_bf3.h -- weight gradient of SimpleCNN's first convolution (observation ingest fused, simple_cnn.py:139-156,68-74) on the
// bf16 matrix pipe, fp32-equivalent arithmetic.  Counterpart of obs_conv_bf3.h; second-largest call site of the C2 step.
//
//      dW[co][ci][kh][kw] = sum over output pixels r = (img, ho, wo) of  X[r][(kh, kw, ci)] * dY[r][co]
//
// Quad fast path of ObsConvWgradProb with an 8x8 filter (KH*KW == 64), Cout <= 32.  As in the forward kernel the uint8 rgb operand
// is EXACT in one bf16 plane (its 1/255 is applied to the rgb rows of dW by the epilogue, as the fp32 kernel does); depth and dY take
// the exact three-term split.  Per 16 reduction elements and 32x32 output tile: 3 MFMAs for an rgb tile, 6 for a depth tile.
//
// The reduction runs over output pixels, but both operands are stored pixel-major (X rows are 12 B rgb + 16 B depth per 4 taps, dY
// rows are Cout floats), whereas an MFMA lane needs 8 CONSECUTIVE reduction elements of one output row.  The transpose happens in
// registers: a thread gathers the same gather unit (4 taps x rgbd) at 4 consecutive pixels and then owns, for each of its 16 (tap,
// channel) rows, 4 consecutive-k values = one 8-byte LDS store per plane (dY: 2 pixels x 4 channels -> 4-byte stores).
//
// Reduction tile = 64 consecutive wo of ONE output row (img, ho) -- block-uniform frame / row decode on the scalar unit, no per-lane
// divisions; rows with Wo % 64 != 0 are padded: the A gather is clamped to the last pixel, dY is zero there.
// Output tile = all of dW (256 x 32): rows re-ordered rgb-first (6 tiles) then depth (2 tiles).  The four waves of a workgroup split
// the REDUCTION (wave w takes k-group w of every tile: 18 + 12 MFMAs, 15 fragment reads) and are summed through LDS at the end;
// workgroups split the pixel range, one slab each, summed by the split-K second pass (igemm.h) which applies the epilogue.
#pragma once
#include "igemm_bf3.h"

namespace hab {

constexpr int OWG_BK = 64;             // pixels per reduction tile
constexpr int OWG_P = OWG_BK + 8;      // LDS row pitch (bf16 elements): 144 B
constexpr int OWG_RGB_ROWS = 192, OWG_DEP_ROWS = 64;
constexpr size_t OWG_LDS_BYTES = (size_t)(OWG_RGB_ROWS + 3 * OWG_DEP_ROWS + 3 * 32) * OWG_P * 2;

struct ObsWgradGeom {
    FastDiv dWB, dHo;  // tile -> (row, wo block) -> (img, ho)
    int WB, tiles, tiles_per_wg;
};

__global__ void __launch_bounds__(256, 2) obs_wgrad_bf3_kernel(const ObsConvWgradProb p, const ObsWgradGeom gg, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* Argb = smem16;                            // [192][OWG_P]          row = tap * 3 + c
    unsigned short* Adep = Argb + OWG_RGB_ROWS * OWG_P;       // [3][64][OWG_P]        row = tap
    unsigned short* Bs = Adep + 3 * OWG_DEP_ROWS * OWG_P;     // [3][32][OWG_P]        row = co
    constexpr int ADP = OWG_DEP_ROWS * OWG_P, BP = 32 * OWG_P;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const ConvGeom& g = p.g;
    const int upr = g.KW >> 2;  // gather units per filter row

    // A staging role: gather unit u (4 taps kw0..kw0+3 of filter row kh), pixels 4 kr .. 4 kr + 3 of the tile
    const int a_u = t >> 4, a_kr = t & 15;
    const int a_kh = a_u / upr, a_kw0 = (a_u - a_kh * upr) * 4;
    // B staging role: channels 4 coq .. +3, pixels 2 kr2, 2 kr2 + 1
    const int b_coq = t >> 5, b_kr = t & 31;
    const bool b_col_ok = b_coq * 4 < p.N;  // N % 4 == 0 (checked by the launcher)

    const int tile_begin = blockIdx.x * gg.tiles_per_wg;
    const int tile_end = min(gg.tiles, tile_begin + gg.tiles_per_wg);

    struct Rgb12 { uint32_t d0, d1, d2; };
    Rgb12 a_rgb[4];
    f32x4 a_dep[4], b_raw[2];
    f32x4 cs = zero4();
    const bool has_cs = p.colsum != nullptr;

    auto fetch = [&](int tile) {
        int row, wb, img, ho;
        gg.dWB.divmod(tile, row, wb);
        gg.dHo.divmod(row, img, ho);
        const int srow = p.obs.srow(img);
        const size_t fpix = ((size_t)srow * g.H + ho * g.stride + a_kh) * g.W + a_kw0;  // pad == 0
        const uint8_t* rgb = p.obs.rgb + fpix * 3;
        const float* dep = p.obs.depth + fpix;
        const int wo0 = wb * OWG_BK + a_kr * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int wo = wo0 + r;
            wo = wo < g.Wo ? wo : g.Wo - 1;
            const int off = wo * g.stride;
            a_rgb[r] = *reinterpret_cast<const Rgb12*>(rgb + off * 3);
            a_dep[r] = ld4(dep + off);
        }
        const float* dyrow = p.dy + ((size_t)row * g.Wo) * p.N + b_coq * 4;
        const int wq = wb * OWG_BK + b_kr * 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bool ok = b_col_ok & (wq + r < g.Wo);
            b_raw[r] = ok ? ld4(dyrow + (size_t)(wq + r) * p.N) : zero4();
        }
    };
    auto stage = [&]() {
        // rgb: row (a_u*4 + q)*3 + c, 4 k-values = byte (3q + c) of the four pixels
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int bi = 3 * q + c;
                unsigned f[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned d = (bi >> 2) == 0 ? a_rgb[r].d0 : ((bi >> 2) == 1 ? a_rgb[r].d1 : a_rgb[r].d2);
                    f[r] = __float_as_uint((float)((d >> (8 * (bi & 3))) & 0xffu));  // exact in bf16
                }
                u32x2 wv;
                wv[0] = bf3_pack(f[0], f[1]);
                wv[1] = bf3_pack(f[2], f[3]);
                *reinterpret_cast<u32x2*>(Argb + ((a_u * 4 + q) * 3 + c) * OWG_P + a_kr * 4) = wv;
            }
        // depth: row a_u*4 + q
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = a_dep[r][q];
            unsigned short* d = Adep + (a_u * 4 + q) * OWG_P + a_kr * 4;
            bf3_store_run<4>(x, d, d + ADP, d + 2 * ADP);
        }
        // dY: row co = b_coq*4 + e, 2 k-values
        if (has_cs) cs += b_raw[0] + b_raw[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x[2] = {b_raw[0][e], b_raw[1][e]};
            unsigned short* d = Bs + (b_coq * 4 + e) * OWG_P + b_kr * 2;
            bf3_store_run<2>(x, d, d + BP, d + 2 * BP);
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;

    // Sign schedule (igemm_bf3.h): the bf16 MFMA truncates towards -infinity below its alignment window, a coherent ~0.0025 ulp per
    // instruction.  Waves 0, 1 accumulate +sum, waves 2, 3 the NEGATED sum (dY fragments sign-flipped as they are read); the final
    // cross-wave reduction subtracts, and the truncation biases of the two pairs cancel.
    const unsigned sf = wave >= 2 ? 0x80008000u : 0u;
    u32x4 sign_flip;
    sign_flip[0] = sf; sign_flip[1] = sf; sign_flip[2] = sf; sign_flip[3] = sf;
    if (tile_begin < tile_end) fetch(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        stage();
        __syncthreads();
        if (tile + 1 < tile_end) fetch(tile + 1);
        const int koff = wave * 16 + hi * 8;
        bf16x8 b[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(Bs + pl * BP + li * OWG_P + koff);
            raw ^= sign_flip;  // waves 2, 3: -dY (sign schedule, see below)
            b[pl] = __builtin_bit_cast(bf16x8, raw);
        }
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Argb + (mt * 32 + li) * OWG_P + koff);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[pl], acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(Adep + pl * ADP + (dt * 32 + li) * OWG_P + koff);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q) acc[6 + dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b[PB[q]], acc[6 + dt], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- sum the four waves (k-groups) through LDS, then write this workgroup's slab in the problem's row order i = tap*4 + ci ----
    float* red = reinterpret_cast<float*>(smem16);  // [256][32] fp32 = 32 KB (the tile buffers are free: last barrier passed)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int rrow = mt * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
                    float* dst = red + rrow * 32 + li;
                    *dst = (w == 0 ? 0.f : *dst) + (w >= 2 ? -acc[mt][v] : acc[mt][v]);
                }
        }
        __syncthreads();
    }
    const int MP = p.M + (has_cs ? 1 : 0);
    float* slab = partial + (size_t)blockIdx.x * MP * p.N;
    for (int e = t; e < 256 * 32; e += 256) {
        const int rrow = e >> 5, co = e & 31;
        const int i = rrow < OWG_RGB_ROWS ? (rrow / 3) * 4 + (rrow % 3) : (rrow - OWG_RGB_ROWS) * 4 + 3;
        if (co < p.N) slab[(size_t)i * p.N + co] = red[e];
    }
    if (has_cs) {  // bias gradient: column sums of dY; thread t holds channels b_coq*4.. of its pixels
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + t * 4) = cs;
        __syncthreads();
        if (t < p.N) {
            float s = 0.f;
            for (int q = 0; q < 32; ++q) s += red[((t >> 2) * 32 + q) * 4 + (t & 3)];
            slab[(size_t)p.M * p.N + t] = s;
        }
    }
}

// returns HAB_OK, an error, or 1 when the problem / workspace does not fit this path (caller falls back to the generic kernels)
inline int obs_wgrad_bf3_launch(const ObsConvWgradProb& p, float* ws, size_t ws_floats, hipStream_t stream) {
    const ConvGeom& g = p.g;
    if (!p.quad || g.KH * g.KW != 64 || p.M != 256 || p.N > 32 || (p.N & 3) || g.pad != 0 || !ws) return 1;
    ObsWgradGeom gg;
    gg.WB = cdiv(g.Wo, OWG_BK);
    gg.tiles = g.B * g.Ho * gg.WB;
    gg.dWB = FastDiv(gg.WB);
    gg.dHo = FastDiv(g.Ho);
    constexpr int wgs_env = 512;
    int wgs = gg.tiles < wgs_env ? gg.tiles : wgs_env;
    const int MP = p.M + (p.colsum ? 1 : 0);
    while (wgs > 1 && (size_t)wgs * MP * p.N > ws_floats) wgs >>= 1;
    if ((size_t)wgs * MP * p.N > ws_floats) return 1;
    gg.tiles_per_wg = cdiv(gg.tiles, wgs);
    wgs = cdiv(gg.tiles, gg.tiles_per_wg);
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(obs_wgrad_bf3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)OWG_LDS_BYTES);
    if (attr_err != hipSuccess) return (int)attr_err;
    obs_wgrad_bf3_kernel<<<wgs, 256, OWG_LDS_BYTES, stream>>>(p, gg, ws);
    HAB_LAUNCH_CHECK();
    if (wgs > 1) {
        igemm_splitk_reduce<ObsConvWgradProb>(p, ws, wgs, stream);
        HAB_LAUNCH_CHECK();
    } else {
        igemm_splitk_reduce<ObsConvWgradProb>(p, ws, 1, stream);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

}  // namespace hab
