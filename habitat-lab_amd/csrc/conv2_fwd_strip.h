// conv2_fwd_strip.h -- forward of SimpleCNN's second convolution (4x4 / stride 2, 32 -> 64 channels; simple_cnn.py:70-83: 63 x 63 -> 30 x 30
// at 256 x 256 observations, any input up to 63 pixels wide -- 20 x 20 at 84^2, 31 x 31 at 128^2, 55 x 55 at 224^2 -- through the
// runtime-geometry instantiation) with the INPUT STRIP resident in LDS and the WEIGHTS resident in registers.
//
// Why: the implicit-GEMM form (igemm_bf3.h, 128 x 64 tiles) re-reads every input element 4x (16 taps / stride^2) and the whole 128 KB
// filter once per 128 output pixels through L2 -- 5.5 GB of L2 -> LDS traffic per 2048 frames, the bound of that kernel (NOTEBOOK.md 5c:
// removing its split VALU changes nothing, its DMA-only skeleton takes 0.39 of 0.55 ms).  The filter as three bf16 planes is 196 KB, more
// than LDS, so the patch-resident scheme of conv_patch_bf3.h does not apply as it stands.  Here the reduction (16 taps x 32 channels = 512)
// is cut ACROSS THE WAVES of a workgroup instead:
//   * a workgroup owns a strip of R output rows of one frame: the 2 R + 2 input rows are read from HBM once (whole rows, 16-byte loads),
//     split once into three bf16 planes, [pixel][32 channels] with 64-byte rows, columns in even / odd order (stride 2: the 32 pixels of
//     an MFMA tile read consecutive rows), 16-byte chunks XOR-swizzled by (row >> 2) & 3 (the 16 lanes of a read phase hit 64 banks);
//   * wave w of 8 owns the two taps (kh = w >> 1, kw = 2 (w & 1) .. +1): its 64 x 64 slice of the filter -- 4 k-steps x 2 output-channel
//     tiles x 3 planes = 96 VGPRs -- is loaded and split ONCE per workgroup and stays in registers for every strip the (persistent)
//     workgroup processes; per 32-pixel tile the wave issues 4 x 3 fragment reads and 48 MFMAs (operands swapped: a lane ends up with
//     4 consecutive output channels of one pixel);
//   * the eight partial sums of a tile meet in LDS ([wave][pixel][64 + 4 pad] fp32, conflict-free 16-byte accesses) and are summed in
//     wave order (fixed: deterministic), + bias, ReLU, one coalesced 16-byte store per thread;
//   * the next strip's global loads are in flight under the MFMAs (registers), as in wgrad3x3_bf3.h.
// Sign schedule as everywhere on the split path: every second workgroup accumulates the negated sum (weights negated when split).
#pragma once
#include "igemm_bf3.h"

namespace hab {

struct C2fArgs {
    const float* x;     // [B][63][63][32]
    const float* wf;    // packed filter [64 co][4][4][32 ci] (the engine's forward layout)
    const float* bias;  // [64] or null
    float* y;           // [B][30][30][64]
    int B;
    int strips, items;
    int relu, sign_schedule;
    int H, W, Ho, Wo;   // runtime-geometry instantiation (RT): input H x W (W <= 63), output Ho x Wo
};

template <int R>
struct C2fCfg {
    static constexpr int W = 63, C = 32, N = 64, Wo = 30, Ho = 30, NT = 512;
    static constexpr int XRS = 2 * (R - 1) + 4, XROWS = XRS * W, NPIX = R * Wo, NPT = (NPIX + 31) / 32;
    static constexpr int XU = XRS * W * C / 4, XPT = (XU + NT - 1) / NT;
    static constexpr int WH = (W + 1) / 2;          // even columns first, then the odd ones
    static constexpr int X_PLANE = XROWS * 32;      // bf16 elements
    static constexpr int RED_LD = N + 4;            // floats per pixel row of a partial tile (pad: 16-byte accesses of 16 lanes spread over all banks)
    static constexpr size_t X_BYTES = (size_t)3 * X_PLANE * 2, RED_BYTES = (size_t)8 * 32 * RED_LD * 4;
    static constexpr size_t LDS_BYTES = X_BYTES + RED_BYTES;
    static_assert(X_BYTES % 16 == 0 && Ho % R == 0, "");
    __host__ __device__ static constexpr int xcol(int w) { return (w & 1) * WH + (w >> 1); }
};

// RT = false: the benchmark geometry (63 x 63 -> 30 x 30), every index a compile-time constant.  RT = true: H, W, Ho, Wo from the
// arguments (W <= 63: the register prefetch and the LDS image are sized for 63); the last strip may hold one output row, input rows
// below the image are staged as zeros and never fetched.
template <int R, bool RT>
__global__ void __launch_bounds__(512) conv2_fwd_strip_kernel(const C2fArgs a) {
    using Cfg = C2fCfg<R>;
    constexpr int NT = Cfg::NT;
    const int W = RT ? a.W : Cfg::W, H = RT ? a.H : Cfg::W, Wo = RT ? a.Wo : Cfg::Wo, Ho = RT ? a.Ho : Cfg::Ho;
    const int WH = (W + 1) / 2, X_PLANE = Cfg::XRS * W * 32, XU = Cfg::XRS * W * 8, NPIX = R * Wo, NPT = (NPIX + 31) / 32;
    auto xcol = [&](int w) { return (w & 1) * WH + (w >> 1); };
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* xs = smem16;                                                        // [plane][XROWS][32], swizzled
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem16) + (size_t)3 * X_PLANE * 2);  // [8][32][RED_LD]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = wave >> 1, kwp = wave & 1;
    const int li = lane & 31, hi = lane >> 5;

    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int per_xcd = (a.items + 7) >> 3, per_wg = (per_xcd + wg_per_xcd - 1) / wg_per_xcd;
    const int xcd_end = min(a.items, (xcd + 1) * per_xcd);
    const int first = min(xcd_end, xcd * per_xcd + jw * per_wg), last = min(xcd_end, first + per_wg);
    if (first >= last) return;

    const bool flip = a.sign_schedule && (blockIdx.x & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;

    // ---- this wave's slice of the filter: k-step j = (tap kw = 2 kwp + (j >> 1), channels 16 (j & 1) .. +15), lane = (co, 8 channels) ----
    bf16x8 bw[4][2][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int co = ct * 32 + li, kw = 2 * kwp + (j >> 1), ci = (j & 1) * 16 + hi * 8;
            const float* src = a.wf + ((size_t)(co * 4 + kh) * 4 + kw) * 32 + ci;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
            unsigned p[3][4];
            bf3_split2(v0[0], v0[1], p[0][0], p[1][0], p[2][0]);
            bf3_split2(v0[2], v0[3], p[0][1], p[1][1], p[2][1]);
            bf3_split2(v1[0], v1[1], p[0][2], p[1][2], p[2][2]);
            bf3_split2(v1[2], v1[3], p[0][3], p[1][3], p[2][3]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bw[j][ct][pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0] ^ sgn2, p[pl][1] ^ sgn2, p[pl][2] ^ sgn2, p[pl][3] ^ sgn2});
        }

    f32x4 xr[Cfg::XPT];
    int pf_units = XU;  // RT: units of the prefetched strip that lie inside the image (whole input rows)
    auto fetch = [&](int item) {  // issues the loads only (see wgrad3x3_bf3.h)
        const int img = item / a.strips, ho0 = (item - img * a.strips) * R;
        const float* xb = a.x + ((size_t)img * H + ho0 * 2) * (size_t)(W * 32);
        if (RT) pf_units = min(Cfg::XRS, H - 2 * ho0) * W * 8;
#pragma unroll
        for (int j = 0; j < Cfg::XPT; ++j) {
            const int u = t + j * NT;
            const bool in = RT ? u < pf_units : (Cfg::XU % NT == 0 || u < Cfg::XU);
            xr[j] = *reinterpret_cast<const f32x4*>(xb + (in ? (size_t)u * 4 : 0));
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < Cfg::XPT; ++j) {
            const int u = t + j * NT;
            if (RT ? u >= XU : (Cfg::XU % NT != 0 && u >= Cfg::XU)) continue;
            const int c4 = u & 7, pix = u >> 3, w = pix % W, hh = pix / W;
            const int row = hh * W + xcol(w);
            unsigned short* dst = xs + row * 32 + (((c4 >> 1) ^ ((row >> 2) & 3)) << 3) + (c4 & 1) * 4;
            bf3_store4((!RT || u < pf_units) ? xr[j] : f32x4{0.f, 0.f, 0.f, 0.f}, dst, dst + X_PLANE, dst + 2 * X_PLANE);
        }
    };

    // reduction / epilogue role of a thread: pixel t >> 4 of the tile, output channels 4 (t & 15) .. +3
    const int rpix = t >> 4, rco = (t & 15) * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + rco);

    fetch(first);
    for (int item = first; item < last; ++item) {
        stage();
        __syncthreads();
        if (item + 1 < last) fetch(item + 1);
        const int img = item / a.strips, ho0 = (item - img * a.strips) * R;
        float* yb = a.y + ((size_t)img * Ho + ho0) * (size_t)(Wo * 64);
        const int npix_here = RT ? min(R, Ho - ho0) * Wo : NPIX;  // (RT: the last strip of an odd Ho holds one row)
#pragma unroll 1
        for (int pt = 0; pt < NPT; ++pt) {
            // ---- this wave's share of the tile: pixels pt*32 .. +31, its two taps ----
            const int p = pt * 32 + li;
            const int pc = (!RT && Cfg::NPIX % 32 == 0) || p < NPIX ? p : 0;
            const int hol = pc / Wo, wo = pc - hol * Wo;
            const int row0 = (hol * 2 + kh) * W + wo;  // + the tap's column class offset
            f32x16 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kw = 2 * kwp + (j >> 1);
                const int row = row0 + (kw & 1) * WH + (kw >> 1);
                const int chunk = (j & 1) * 2 + hi;
                const unsigned short* src = xs + row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3);
                bf16x8 af[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(src + pl * X_PLANE);
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest weight first (A: input, B: filter)
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)  // operands swapped: D[m = output channel][n = pixel]
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][ct][PB[q]], af[PA[q]], acc[ct], 0, 0, 0);
            }
            if (pt > 0) __syncthreads();  // the previous tile's reduction has read `red`
            // lane (pixel li): channels ct*32 + 8 g + 4 hi .. +3 in acc[ct][4 g .. 4 g + 3]
            float* mine = red + (size_t)(wave * 32 + li) * Cfg::RED_LD;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(mine + ct * 32 + 8 * g + 4 * hi) =
                        f32x4{acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]};
            __syncthreads();
            // ---- the eight partial sums, in wave order ----
            f32x4 s = *reinterpret_cast<const f32x4*>(red + (size_t)rpix * Cfg::RED_LD + rco);
#pragma unroll
            for (int w8 = 1; w8 < 8; ++w8) s += *reinterpret_cast<const f32x4*>(red + (size_t)(w8 * 32 + rpix) * Cfg::RED_LD + rco);
            if (flip) s = -s;
            s += bias4;
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : 0.f;
            }
            const int op = pt * 32 + rpix;
            if (RT ? op < npix_here : (Cfg::NPIX % 32 == 0 || op < Cfg::NPIX)) *reinterpret_cast<f32x4*>(yb + (size_t)op * 64 + rco) = s;
        }
        __syncthreads();  // every wave is done with this strip's image (and with `red`)
    }
}

inline bool conv2_fwd_strip_covers(const ConvGeom& g) {
    return g.KH == 4 && g.KW == 4 && g.stride == 2 && g.pad == 0 && g.C == 32 && g.Cout == 64 && g.H >= 4 && g.W >= 4 && g.W <= 63;
}

// 1: shape not covered.
inline int conv2_fwd_strip(const ConvFwdProb& p, float* /*ws*/, size_t /*ws_floats*/, hipStream_t stream) {
    const ConvGeom& g = p.g;
    if (!conv2_fwd_strip_covers(g)) return 1;
    if ((p.ldy != 0 && p.ldy != 64) || g.B < 16) return 1;
    if ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.w) | reinterpret_cast<uintptr_t>(p.y) | reinterpret_cast<uintptr_t>(p.bias)) & 15) return 1;
    constexpr int R = 2;
    using Cfg = C2fCfg<R>;
    C2fArgs a;
    a.x = p.x; a.wf = p.w; a.bias = p.bias; a.y = p.y; a.B = g.B;
    a.H = g.H; a.W = g.W; a.Ho = (g.H - 4) / 2 + 1; a.Wo = (g.W - 4) / 2 + 1;
    a.strips = (a.Ho + R - 1) / R;
    if ((long long)g.B * a.strips > 0x7fffffffLL) return 1;
    a.items = g.B * a.strips;
    a.relu = p.relu;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    const bool bench_geom = g.H == 63 && g.W == 63;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = [] {
        const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_fwd_strip_kernel<R, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_fwd_strip_kernel<R, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        return e0 != hipSuccess ? e0 : e1;
    }();
    if (attr_err != hipSuccess) return (int)attr_err;
    int grid = 256;
    while (grid > 8 && grid > a.items) grid -= 8;
    if (bench_geom) conv2_fwd_strip_kernel<R, false><<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(a);
    else conv2_fwd_strip_kernel<R, true><<<grid, Cfg::NT, (size_t)3 * Cfg::XRS * g.W * 32 * 2 + Cfg::RED_BYTES, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
