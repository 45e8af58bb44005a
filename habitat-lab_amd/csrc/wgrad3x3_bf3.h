// wgrad3x3_bf3.h -- weight gradient of the small-channel convolutions (3x3 / stride 1 with 32 or 64 input and output channels; since
// the k-slot groups may cross output rows also SimpleCNN conv2: 4x4 / stride 2, 32 -> 64 at 63 x 63) on the bf16 matrix pipe (exact
// 3-term operand split, six partial products, igemm_bf3.h), operands resident in LDS in their NATURAL layout.
//
//      dW[co][kh][kw][ci] = sum over (img, ho, wo) of  dY[img][ho][wo][co] * X[img][ho + kh - p][wo + kw - p][ci]
//
// The contraction index is the pixel; both operands are stored pixel-major (NHWC), i.e. k-strided for the MFMA, whose lane wants 8
// consecutive k of one channel.  The implicit-GEMM weight gradient (igemm_bf3.h with A_RC = B_RC = false) transposes in registers
// and re-gathers every x element once per tap through L2: 60 TFLOP/s-equivalent on SimpleCNN conv3 and ResNet layer1, a sixth of
// either roofline.  Here
//   * a workgroup owns a strip of R output rows of one image: the x rows it needs (R + 2, zero-padded columns included) and the dY
//     rows are read from HBM ONCE, with whole-row coalesced 16-byte loads, split once, and stored as [pixel][32 channels] bf16
//     planes (64 bytes per pixel);
//   * MFMA fragments are built with the LDS TRANSPOSE read (ds_read_b64_tr_b16, semantics pinned on the hardware by
//     tools/ubench/tr_read_probe.hip): every 16-lane group reads a [4 pixels][16 channels] block and lane i receives channel i of
//     the 4 pixels -- exactly the k-contiguous fragment, from the pixel-major image.  A filter tap is a ROW offset in that image
//     (64-byte pitch), so all nine taps read the same resident strip with aligned reads; nothing is re-gathered;
//   * k-slots are groups of 4 consecutive pixels of one output row (Wo % 4 == 0), two groups per lane and K = 16 step; 4 pixel rows of
//     two 16-lane groups are 256 contiguous bytes = all 64 banks once;
//   * wave (kh, ks) of a workgroup accumulates the three taps of filter row kh over the k-steps s = ks mod KS: 3 x C/32 x N/32
//     accumulator tiles; the dY fragments are read once per k-step and reused by the 3 x C/32 tap tiles;
//   * workgroups are persistent over an XCD-contiguous range of strips (consecutive strips of an image meet in the same L2 for
//     their shared halo rows) with the next strip's global loads in flight during the current strip's MFMAs;
//   * every (workgroup, ks) writes one slab [9 C][N]; the implicit-GEMM path's split-K reduction sums the slabs in slab order and
//     scatters to OIHW (deterministic, as everywhere).  Sign schedule as in igemm_bf3.h: every second workgroup accumulates the
//     negated sum.
#pragma once
#include "igemm_bf3.h"

namespace hab {

struct W3bArgs {
    const float* x;
    const float* dy;
    float* partial;
    int B, H, Ho;
    int strips;  // Ho / R
    int items;   // B * strips
    int sign_schedule;
    int colsum;  // slabs carry one more row: the column sums of dY (bias gradient)
};

template <int C32, int N32, int W, int PAD, int KHW, int S, int R, int KS, int NS>
struct W3bCfg {
    static constexpr int C = 32 * C32, N = 32 * N32, WP = W + 2 * PAD, Wo = (WP - KHW) / S + 1, NT = 64 * KHW * KS * NS, NW = N32 / NS;
    static constexpr int XRS = (R - 1) * S + KHW;  // x rows under a strip of R output rows
    static constexpr int XROWS = XRS * WP, NPIX = R * Wo, NQ = (NPIX + 3) / 4, NSTEPS = (NQ + 3) / 4, YROWS = NSTEPS * 16;
    static constexpr int XU = XRS * W * C / 4, YU = R * Wo * N / 4, XPT = (XU + NT - 1) / NT, YPT = (YU + NT - 1) / NT;
    static constexpr int X_HALF = XROWS * 32, Y_HALF = YROWS * 32;  // bf16 elements of one [rows][32] image
    static constexpr size_t LDS_BYTES = (size_t)(3 * C32 * X_HALF + 3 * N32 * Y_HALF) * 2;
    // Column order of an x row in LDS.  Stride 2: even columns first, then the odd ones -- the four pixels wo .. wo + 3 of a k-slot
    // group read columns 2 wo + kw, i.e. CONSECUTIVE rows of one parity class (256 contiguous bytes, every bank once) instead of rows
    // 128 bytes apart (two-way conflicts)
    static constexpr int WH = (WP + 1) / 2;
    __host__ __device__ static constexpr int xcol(int w) { return S == 2 ? (w & 1) * WH + (w >> 1) : w; }
    __host__ __device__ static constexpr int tap_col(int kw) { return S == 2 ? (kw & 1) * WH + (kw >> 1) : kw; }
    static_assert(S == 1 || S == 2, "");
    static_assert(LDS_BYTES % 16 == 0 && N32 % NS == 0 && NT % (N / 4) == 0 && LDS_BYTES >= (size_t)NT * 16, "");
};

typedef short w3b_v4s __attribute__((ext_vector_type(4)));
typedef short w3b_v8s __attribute__((ext_vector_type(8)));

// [4 pixels][16 channels] block at `p` (row pitch 64 bytes): lane i of a 16-lane group passes the address of 8-byte chunk i
// (row i / 4, chunk i % 4) and receives channel i of the four pixels.
__device__ __forceinline__ w3b_v4s w3b_tr_read(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3b_v4s*)p);
}
__device__ __forceinline__ bf16x8 w3b_join(const w3b_v4s lo, const w3b_v4s hi) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int C32, int N32, int W, int PAD, int KHW, int S, int R, int KS, int NS>
__global__ void __launch_bounds__(64 * KHW * KS * NS) wgrad3x3_bf3_kernel(const W3bArgs a) {
    using Cfg = W3bCfg<C32, N32, W, PAD, KHW, S, R, KS, NS>;
    constexpr int C = Cfg::C, N = Cfg::N, WP = Cfg::WP, Wo = Cfg::Wo, NT = Cfg::NT, NW = Cfg::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* xs = smem16;                          // [plane][channel half][XROWS][32]
    unsigned short* ys = smem16 + 3 * C32 * Cfg::X_HALF;  // [plane][channel half][YROWS][32]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = wave % KHW, ns = (wave / KHW) % NS, ks = wave / (KHW * NS);  // filter row, output-channel part, k-step class

    // zero padding columns / out-of-image rows of x and the tail rows of dY (k-slots beyond the strip): written once
    for (int i = t; i < (int)(Cfg::LDS_BYTES / 16); i += NT) reinterpret_cast<u32x4*>(smem16)[i] = u32x4{0u, 0u, 0u, 0u};

    // this workgroup's strips: XCD x takes the x-th eighth of the strips, its workgroups contiguous pieces of it
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int per_xcd = (a.items + 7) >> 3, per_wg = (per_xcd + wg_per_xcd - 1) / wg_per_xcd;
    const int xcd_end = min(a.items, (xcd + 1) * per_xcd);
    const int first = min(xcd_end, xcd * per_xcd + jw * per_wg), last = min(xcd_end, first + per_wg);

    const bool flip = a.sign_schedule && (blockIdx.x & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;

    f32x4 xr[Cfg::XPT], yr[Cfg::YPT];
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};  // this thread's share of the dY column sums: its units all hold channels 4 (t % (N/4)) .. +3
    int pf_ho0 = 0;  // first output row of the strip whose loads are in xr / yr
    // fetch only ISSUES the loads (rows outside the image load the tensor's first bytes instead): anything that consumed a loaded
    // value here would put the wait for HBM in front of the MFMAs the loads are meant to hide behind -- the zeros go in at stage()
    auto fetch = [&](int item) {
        const int img = item / a.strips, ho0 = (item - img * a.strips) * R;
        pf_ho0 = ho0;
        const long long xrow0 = (long long)img * a.H + ho0 * S - PAD;  // first x row of the strip (may lie outside the image)
#pragma unroll
        for (int j = 0; j < Cfg::XPT; ++j) {
            const int u = t + j * NT;
            const int hh = u / (W * C / 4);
            const bool ok = (Cfg::XU % NT == 0 || u < Cfg::XU) && (unsigned)(ho0 * S - PAD + hh) < (unsigned)a.H;
            const long long off = ok ? xrow0 * (W * C) + (long long)u * 4 : 0;
            xr[j] = *reinterpret_cast<const f32x4*>(a.x + off);
        }
        const float* yb = a.dy + ((size_t)img * a.Ho + ho0) * (size_t)(Wo * N);
#pragma unroll
        for (int j = 0; j < Cfg::YPT; ++j) {
            const int u = t + j * NT;
            const bool ok = Cfg::YU % NT == 0 || u < Cfg::YU;
            yr[j] = *reinterpret_cast<const f32x4*>(yb + (ok ? (size_t)u * 4 : 0));
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < Cfg::XPT; ++j) {
            const int u = t + j * NT;
            if (Cfg::XU % NT != 0 && u >= Cfg::XU) continue;
            const int c4 = u % (C / 4), pix = u / (C / 4), w = pix % W, hh = pix / W;
            unsigned short* dst = xs + (c4 >> 3) * Cfg::X_HALF + (hh * WP + Cfg::xcol(w + PAD)) * 32 + (c4 & 7) * 4;
            const bool in_image = (unsigned)(pf_ho0 * S - PAD + hh) < (unsigned)a.H;
            bf3_store4(in_image ? xr[j] : f32x4{0.f, 0.f, 0.f, 0.f}, dst, dst + C32 * Cfg::X_HALF, dst + 2 * C32 * Cfg::X_HALF);
        }
#pragma unroll
        for (int j = 0; j < Cfg::YPT; ++j) {
            const int u = t + j * NT;
            if (Cfg::YU % NT != 0 && u >= Cfg::YU) continue;
            const int c4 = u % (N / 4), pix = u / (N / 4);
            unsigned short* dst = ys + (c4 >> 3) * Cfg::Y_HALF + pix * 32 + (c4 & 7) * 4;
            cs += yr[j];
            unsigned a1, a2, a3, b1, b2, b3;
            bf3_split2(yr[j][0], yr[j][1], a1, a2, a3);
            bf3_split2(yr[j][2], yr[j][3], b1, b2, b3);
            *reinterpret_cast<u32x2*>(dst) = u32x2{a1 ^ sgn2, b1 ^ sgn2};
            *reinterpret_cast<u32x2*>(dst + N32 * Cfg::Y_HALF) = u32x2{a2 ^ sgn2, b2 ^ sgn2};
            *reinterpret_cast<u32x2*>(dst + 2 * N32 * Cfg::Y_HALF) = u32x2{a3 ^ sgn2, b3 ^ sgn2};
        }
    };

    f32x16 acc[KHW][C32][NW];
#pragma unroll
    for (int kw = 0; kw < KHW; ++kw)
#pragma unroll
        for (int hc = 0; hc < C32; ++hc)
#pragma unroll
            for (int hn = 0; hn < NW; ++hn)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[kw][hc][hn][v] = 0.f;

    // lane constants of the transpose reads: 16-lane group g = lane >> 4 reads channels 16 (g & 1) .. +15 of the k-block lane >> 5
    // and passes the address of ITS 8-byte chunk: pixel (i >> 2) of the group's four, chunk i & 3 -- the four pixels need not be
    // equidistant in LDS, so a k-slot group may cross an output row and the taps of a strided convolution are legal
    const int i16 = lane & 15;
    const int chunk_off = ((lane >> 4) & 1) * 16 + (i16 & 3) * 4;  // elements inside the pixel's 32-channel row
    const int kblk = lane >> 5;

    auto compute = [&]() {
        for (int s = ks; s < Cfg::NSTEPS; s += KS) {
            int yoff[2], xoff[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int pix = 4 * (4 * s + 2 * kblk + r) + (i16 >> 2);  // this lane's pixel of the strip (row-major over R x Wo)
                yoff[r] = pix * 32 + chunk_off;                            // beyond the strip: the zero tail rows of the dY image
                const int pc = Cfg::NPIX % 16 == 0 ? pix : (pix < Cfg::NPIX ? pix : 0);
                const int hol = pc / Wo, wo = pc - hol * Wo;
                xoff[r] = ((hol * S + kh) * WP + (S == 2 ? wo : wo * S)) * 32 + chunk_off;
            }
            bf16x8 bfr[NW][3];
#pragma unroll
            for (int hn = 0; hn < NW; ++hn)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned short* src = ys + (pl * N32 + ns * NW + hn) * Cfg::Y_HALF;
                    bfr[hn][pl] = w3b_join(w3b_tr_read(src + yoff[0]), w3b_tr_read(src + yoff[1]));
                }
#pragma unroll
            for (int kw = 0; kw < KHW; ++kw)
#pragma unroll
                for (int hc = 0; hc < C32; ++hc) {
                    bf16x8 afr[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned short* src = xs + (pl * C32 + hc) * Cfg::X_HALF + Cfg::tap_col(kw) * 32;
                        afr[pl] = w3b_join(w3b_tr_read(src + xoff[0]), w3b_tr_read(src + xoff[1]));
                    }
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest weight first
#pragma unroll
                    for (int q6 = 0; q6 < 6; ++q6)
#pragma unroll
                        for (int hn = 0; hn < NW; ++hn)
                            acc[kw][hc][hn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[PA[q6]], bfr[hn][PB[q6]], acc[kw][hc][hn], 0, 0, 0);
                }
        }
    };

    __syncthreads();  // the zero fill is complete
    if (first < last) fetch(first);
    for (int item = first; item < last; ++item) {
        stage();
        __syncthreads();
        if (item + 1 < last) fetch(item + 1);  // in flight during this strip's MFMAs
        compute();
        __syncthreads();
    }

    // slab[blockIdx.x * KS + ks][i = (kh*KW + kw)*C + hc*32 + ci][co = (ns*NW + hn)*32 + li]  (+ row KH KW C: column sums of dY)
    const int li = lane & 31, hi = lane >> 5;
    constexpr int MROWS = KHW * KHW * C;
    const size_t slab = (size_t)(MROWS + (a.colsum ? 1 : 0)) * N;
    float* out = a.partial + (size_t)(blockIdx.x * KS + ks) * slab;
#pragma unroll
    for (int kw = 0; kw < KHW; ++kw)
#pragma unroll
        for (int hc = 0; hc < C32; ++hc)
#pragma unroll
            for (int hn = 0; hn < NW; ++hn)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int ci = (v & 3) + 8 * (v >> 2) + 4 * hi;
                    const float val = acc[kw][hc][hn][v];
                    out[(size_t)((kh * KHW + kw) * C + hc * 32 + ci) * N + (ns * NW + hn) * 32 + li] = flip ? -val : val;
                }
    if (a.colsum) {  // the strips' dY passed through this workgroup's registers once: fixed-order sum of the threads' shares
        float* red = reinterpret_cast<float*>(smem16);  // the images are dead behind the loop's last barrier
        *reinterpret_cast<f32x4*>(red + t * 4) = cs;
        __syncthreads();
        if (t < N) {
            float sum = 0.f;
            for (int q = (t >> 2); q < NT; q += N / 4) sum += red[q * 4 + (t & 3)];
            a.partial[(size_t)(blockIdx.x * KS) * slab + (size_t)MROWS * N + t] = sum;
#pragma unroll
            for (int z = 1; z < KS; ++z) a.partial[(size_t)(blockIdx.x * KS + z) * slab + (size_t)MROWS * N + t] = 0.f;
        }
    }
}

template <int C32, int N32, int W, int PAD, int KHW, int S, int R, int KS, int NS>
inline int wgrad3x3_bf3_run(const ConvWgradProb& p, float* ws, size_t ws_floats, hipStream_t stream) {
    using Cfg = W3bCfg<C32, N32, W, PAD, KHW, S, R, KS, NS>;
    const ConvGeom& g = p.g;
    W3bArgs a;
    a.x = p.x; a.dy = p.dy; a.partial = ws; a.B = g.B; a.H = g.H; a.Ho = g.Ho;
    a.strips = g.Ho / R;
    a.items = g.B * a.strips;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    a.colsum = p.colsum != nullptr;
    auto kern = wgrad3x3_bf3_kernel<C32, N32, W, PAD, KHW, S, R, KS, NS>;
    // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
    static const hipError_t attr_err = (Cfg::LDS_BYTES > 64 * 1024) ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) : hipSuccess;
    if (attr_err != hipSuccess) return (int)attr_err;
    const size_t MN = (size_t)(KHW * KHW * Cfg::C + (a.colsum ? 1 : 0)) * Cfg::N;
    const int per_cu = (int)std::min<size_t>(Cfg::NT <= 192 ? 4 : 2, (160 * 1024) / Cfg::LDS_BYTES);
    int grid = 256 * std::max(per_cu, 1);
    while (grid > 8 && (grid > a.items || (size_t)grid * KS * MN > ws_floats)) grid -= 8;
    if ((size_t)grid * KS * MN > ws_floats || (grid < 128 && a.items >= 1024)) return 1;  // workspace too small for a chip-filling launch
    kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(a);
    HAB_LAUNCH_CHECK();
    igemm_splitk_reduce<ConvWgradProb>(p, ws, grid * KS, stream);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Shapes with an instance: 1 SimpleCNN conv3 (64 -> 32, 30 x 30, no padding), 2 ResNet layer1 at 128^2 input (32 -> 32, 32 x 32),
// 3 ResNet layer2 (64 -> 64, 16 x 16), 4 SimpleCNN conv2 (4x4 / 2, 32 -> 64, 63 x 63); 0: not covered (the caller keeps the implicit-GEMM weight gradient).
inline int wgrad3x3_bf3_shape(const ConvWgradProb& p) {
    const ConvGeom& g = p.g;
    if (p.Creal != g.C || (size_t)g.B * g.H * g.W * std::max(g.C, g.Cout) >= 0x7fffffffull / 2) return 0;
    if (g.KH == 4 && g.KW == 4 && g.stride == 2 && g.C == 32 && g.Cout == 64 && g.W == 63 && g.pad == 0 && g.Ho % 2 == 0) return 4;
    if (!(g.KH == 3 && g.KW == 3 && g.stride == 1)) return 0;
    if (g.C == 64 && g.Cout == 32 && g.W == 30 && g.pad == 0 && g.Ho % 2 == 0) return 1;
    if (g.C == 32 && g.Cout == 32 && g.W == 32 && g.pad == 1 && g.Ho % 4 == 0) return 2;
    if (g.C == 64 && g.Cout == 64 && g.W == 16 && g.pad == 1 && g.Ho % 4 == 0) return 3;
    return 0;
}
// p.colsum (bias gradient) is produced by the same pass: dY goes through the workgroups' registers anyway.
// Returns 1 when the workspace cannot hold the slabs of a chip-filling launch (the caller falls back to the implicit-GEMM form).
inline int wgrad3x3_bf3(const ConvWgradProb& p, float* ws, size_t ws_floats, hipStream_t stream) {
    if (!ws) return HAB_ERR_ARG;
    // Instances (measured at 2048 frames, kernel + slab reduction; implicit-GEMM weight gradient before):
    //   conv3   R = 2, 6 waves (kh x 2 k-step classes), 58 KB LDS, 2 workgroups / CU: 0.48 ms (1.04; its bias gradient included)
    //   layer1  R = 2, 3 waves, 38 KB LDS, 4 workgroups / CU:                          0.24 ms (0.66); R = 4 with 6 waves: 0.275
    //   layer2  R = 4, 6 waves (kh x 2 output-channel halves), 66 KB LDS:              0.225 ms (0.50); 3 waves x both halves: 0.227
    switch (wgrad3x3_bf3_shape(p)) {
        case 1: return wgrad3x3_bf3_run<2, 1, 30, 0, 3, 1, 2, 2, 1>(p, ws, ws_floats, stream);
        case 2: return wgrad3x3_bf3_run<1, 1, 32, 1, 3, 1, 2, 1, 1>(p, ws, ws_floats, stream);
        case 3: return wgrad3x3_bf3_run<2, 2, 16, 1, 3, 1, 4, 1, 2>(p, ws, ws_floats, stream);
        // SimpleCNN conv2 (4x4 / 2, 32 -> 64 at 63 x 63): R = 2, 8 waves = 4 filter rows x 2 k-step classes, both output-channel
        // halves per wave (an x fragment feeds two tiles), 97 KB LDS: 0.71 ms (im2col form 1.07-1.15); wave = (kh, channel half): 0.75;
        // x columns in row order instead of even / odd order (two-way conflicts of the transpose reads): 0.80
        case 4: return wgrad3x3_bf3_run<1, 2, 63, 0, 4, 2, 2, 2, 1>(p, ws, ws_floats, stream);
    }
    return HAB_ERR_UNSUPPORTED;
}

}  // namespace hab
