// pl_ops.hip -- the pl32 operand-plane path (bf3_planes.h, igemm_pl.h): fp32 <-> planes conversion and the DMA-staged
// split-bf16 contractions that consume planes: convolution forward, convolution data gradient (per stride class and merged), Linear
// forward / data gradient as 1x1 convolutions of 1x1 images.
#include "igemm_pl.h"
#include "ops.h"
#include "prob_build.h"
#include "../../include/habitat_amd.h"

namespace hab {

// ------------------------------------------------------------------------------------------------------
// fp32 [rows][cols] (row stride ld floats) -> compact pl32 planes of the logical array [rows * cols]; cols % 32 == 0.
// HBM-bound: 4 bytes read, 6 written per element.  Used for tensors whose producer is not a contraction epilogue (packed weights once
// per optimiser step, the recurrent encoder's gradient wrt the visual feature) -- activations between contractions never pass here.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ x, long long rows, int cols, int ld, pl16* __restrict__ out) {
    const long long quads = rows * (cols >> 2);
    const int qpr = cols >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
        const long long r = q / qpr;
        const int c = (int)(q - r * qpr) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + c);
        pl_store4(out, (size_t)(r * cols + c), v);
    }
}
__global__ void __launch_bounds__(256) merge_planes_kernel(const pl16* __restrict__ in, long long n, float* __restrict__ out) {
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < (n >> 2); q += (long long)gridDim.x * 256)
        *reinterpret_cast<f32x4*>(out + 4 * q) = pl_load4(in, (size_t)(4 * q));
}

int split_planes(const float* x, long long rows, int cols, int ld, pl16* out, hipStream_t s) {
    if (!x || !out || rows <= 0 || cols <= 0 || (cols % 32) || ld < cols || (ld & 3) || (((uintptr_t)x | (uintptr_t)out) & 15)) return HAB_ERR_ARG;
    const long long quads = rows * (cols >> 2);
    split_planes_kernel<<<(int)fmin(16384.0, (double)cdivl(quads, 256)), 256, 0, s>>>(x, rows, cols, ld, out);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
int merge_planes(const pl16* in, long long n, float* out, hipStream_t s) {
    if (!in || !out || n <= 0 || (n % 32) || (((uintptr_t)in | (uintptr_t)out) & 15)) return HAB_ERR_ARG;
    merge_planes_kernel<<<(int)fmin(16384.0, (double)cdivl(n >> 2, 256)), 256, 0, s>>>(in, n, out);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// Tile choice.  Tall-skinny problems (N = 32 .. 512, M = 10^4 .. 10^7).  LDS per buffer = (BM + BN) * 192 bytes:
//   256 x 32: 54 KB   256 x 64: 60 KB (2 workgroups / CU)   128 x 64: 36 KB (4 / CU)   128 x 32: 30 KB   128 x 128: 48 KB (3 / CU)
// HAB_PL_TALL=0 selects the 128-row tiles for N <= 64 at every M, HAB_PL_DB=1 the double-buffered k-loop (development: A/B runs).
// ------------------------------------------------------------------------------------------------------
static int pl_tall() { static const int v = hab_env_int("HAB_PL_TALL", 1); return v; }
static int pl_db() { static const int v = hab_env_int("HAB_PL_DB", 0); return v; }

template <class P>
static int run_pl(const P& p, const pl16* apl, const pl16* bpl, float* ws, size_t ws_floats, hipStream_t stream) {
    const int target = 256;  // forward-form problems split K only until every CU has a workgroup (as run_igemm)
    const bool tall = pl_tall() && cdiv(p.M, 256) >= 512;  // 256-row tiles only when they still fill the chip twice over
    if (p.N <= 32) {
        if (tall) return pl_db() ? igemm_pl_launch<P, 2, 1, 4, 1, true>(p, apl, bpl, ws, ws_floats, target, stream)
                                 : igemm_pl_launch<P, 2, 1, 4, 1, false>(p, apl, bpl, ws, ws_floats, target, stream);
        return pl_db() ? igemm_pl_launch<P, 1, 1, 4, 1, true>(p, apl, bpl, ws, ws_floats, target, stream)
                       : igemm_pl_launch<P, 1, 1, 4, 1, false>(p, apl, bpl, ws, ws_floats, target, stream);
    }
    if (p.N <= 64) {
        if (tall) return pl_db() ? igemm_pl_launch<P, 2, 2, 4, 1, true>(p, apl, bpl, ws, ws_floats, target, stream)
                                 : igemm_pl_launch<P, 2, 2, 4, 1, false>(p, apl, bpl, ws, ws_floats, target, stream);
        return pl_db() ? igemm_pl_launch<P, 1, 2, 4, 1, true>(p, apl, bpl, ws, ws_floats, target, stream)
                       : igemm_pl_launch<P, 1, 2, 4, 1, false>(p, apl, bpl, ws, ws_floats, target, stream);
    }
    return pl_db() ? igemm_pl_launch<P, 2, 2, 2, 2, true>(p, apl, bpl, ws, ws_floats, target, stream)
                   : igemm_pl_launch<P, 2, 2, 2, 2, false>(p, apl, bpl, ws, ws_floats, target, stream);
}

// Convolution forward on planes: x and the packed weights Wf[co][(kh,kw,ci)] as pl32; output fp32 (y, row stride ldy) and / or planes.
int conv_fwd_pl(const ConvDesc& d, const pl16* xpl, const pl16* wfpl, const float* bias, float* y, int ldy, pl16* ypl, int relu,
                float* ws, size_t ws_floats, hipStream_t stream) {
    if (!xpl || !wfpl || (!y && !ypl)) return HAB_ERR_ARG;
    ConvFwdProb p;
    HAB_TRY(build(p, d, nullptr, nullptr, bias, y, relu));
    if (!p.dma_ok() || (p.K % 32)) return HAB_ERR_UNSUPPORTED;
    if (ypl && ((p.N % 32) || (ldy > 0 && ldy != p.N))) return HAB_ERR_ARG;
    if (y && ldy > 0 && (ldy < p.N || (ldy & 3))) return HAB_ERR_ARG;
    p.ypl = ypl;
    p.ldy = ldy;
    return run_pl(p, xpl, wfpl, ws, ws_floats, stream);
}

// Convolution data gradient on planes: dy and the packed weights Wd[ci][(kh,kw,co)] as pl32; ReLU mask from fp32 (`mask`) or from the
// planes of the layer's input (`maskpl`); output fp32 and / or planes.
int conv_dgrad_pl(const ConvDesc& d, const pl16* dypl, const pl16* wdpl, const float* mask, const pl16* maskpl, float* dx, pl16* dxpl,
                  float* ws, size_t ws_floats, hipStream_t stream) {
    if (!dypl || !wdpl || (!dx && !dxpl)) return HAB_ERR_ARG;
    HAB_TRY(check_conv(d));
    if (dxpl && (d.C % 32)) return HAB_ERR_ARG;
    if (d.stride > 1) {
        ConvDgradMergedProb q;
        q.g = make_geom(d);
        if (ConvDgradMergedProb::applicable(q.g)) {
            q.dy = nullptr; q.w = nullptr; q.mask = mask; q.add = nullptr; q.dx = dx; q.maskpl = maskpl; q.dxpl = dxpl;
            q.finish();
            if (q.dma_ok() && q.K % 32 == 0) return run_pl(q, dypl, wdpl, ws, ws_floats, stream);
        }
    }
    for (int ph = 0; ph < d.stride; ++ph)
        for (int pw = 0; pw < d.stride; ++pw) {
            ConvDgradProb p;
            HAB_TRY(build(p, d, nullptr, nullptr, mask, nullptr, dx, ph, pw));
            p.maskpl = maskpl; p.dxpl = dxpl;
            if (p.Hc <= 0 || p.Wc <= 0) continue;
            if (p.K <= 0 || !p.dma_ok()) return HAB_ERR_UNSUPPORTED;
            HAB_TRY(run_pl(p, dypl, wdpl, ws, ws_floats, stream));
        }
    return HAB_OK;
}

}  // namespace hab

using namespace hab;

// ---- C ABI (include/habitat_amd.h) -----------------------------------------------------------------------------------------------
extern "C" int hab_pl_split(const float* x, int64_t rows, int cols, int ld, uint16_t* planes, hipStream_t stream) {
    return split_planes(x, rows, cols, ld, planes, stream);
}
extern "C" int hab_pl_merge(const uint16_t* planes, int64_t n, float* out, hipStream_t stream) { return merge_planes(planes, n, out, stream); }
extern "C" int hab_conv2d_fwd_pl(const uint16_t* xpl, const uint16_t* wfpl, const float* bias, float* y, int ldy, uint16_t* ypl, int B, int H, int W,
                                 int C, int Cout, int KH, int KW, int stride, int pad, int relu, float* ws, size_t ws_floats, hipStream_t stream) {
    ConvDesc d{B, H, W, C, Cout, KH, KW, stride, pad};
    return conv_fwd_pl(d, xpl, wfpl, bias, y, ldy, ypl, relu, ws, ws_floats, stream);
}
extern "C" int hab_conv2d_dgrad_pl(const uint16_t* dypl, const uint16_t* wdpl, const float* mask, const uint16_t* maskpl, float* dx, uint16_t* dxpl,
                                   int B, int H, int W, int C, int Cout, int KH, int KW, int stride, int pad, float* ws, size_t ws_floats,
                                   hipStream_t stream) {
    ConvDesc d{B, H, W, C, Cout, KH, KW, stride, pad};
    return conv_dgrad_pl(d, dypl, wdpl, mask, maskpl, dx, dxpl, ws, ws_floats, stream);
}
