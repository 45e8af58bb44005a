// problems.h -- operand gathers + epilogues that turn the layers of the policy into instances of
// the igemm kernel (igemm.h).  Activations are NHWC fp32 (channels % 4 == 0 so a 16-byte gather never
// straddles a filter tap); convolution weights are consumed from packed copies
//   Wf[co][(kh,kw,ci)]  (forward)      Wd[ci][(kh,kw,co)]  (data gradient)
// produced by repack_conv (gemm_ops.hip) from the reference-layout (OIHW) master parameters; weight
// gradients are scattered straight back into OIHW.  All functors are __host__ __device__ so that
// tests/hostcheck can execute the very same index math on the CPU against the oracle.
//
// Interface of a problem type P (see igemm.h for how the kernel drives it):
//   M, N, K; A_RC, B_RC (operand forms); optional A_KV (gather-unit width of A, default 4)
//   KCtx k_ctx(k0, k_end)                       per K-tile, block-uniform
//   ACtx a_ctx(i) / BCtx b_ctx(j)               per gather unit, once per tile
//   AKey a_key(KCtx, k, k_end)                  decode of the reduction coordinate (filter tap / pixel), shared by all
//                                               units of a thread in the r-contiguous form
//   ARaw a_fetch(ACtx, KCtx, AKey)              BRANCH-FREE global loads from a masked offset + validity
//   void a_cvt(ACtx, ARaw, k, k_end, f32x4*)    zero fill / conversion, applied when staging to LDS
//   BKey b_key(KCtx, k, k_end); BRaw b_fetch(BCtx, KCtx, BKey); f32x4 b_cvt(BRaw)
//   EpiCol epi_col(n); EpiRow epi_row(m); EpiAux epi_fetch(row, col); epi_store(row, col, aux, v)
//   store(m, n, v)                              = the four epilogue pieces in sequence
#pragma once
#include <math.h>

#include "hab_common.h"
#include "bf3_split.h"

namespace hab {

#define HAB_HD __host__ __device__ inline

HAB_HD f32x4 zero4() {
    f32x4 z;
    z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
    return z;
}
HAB_HD f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

struct Raw4 { f32x4 v; int ok; };
struct KKey { int k, ok; };  // plain reduction coordinate + in-range flag
HAB_HD KKey kkey(int k, int k_end) { KKey q; q.k = k; q.ok = k < k_end; return q; }
HAB_HD f32x4 sel4(const Raw4& r) { return r.ok ? r.v : zero4(); }
struct NoCtx {};
struct NoAux {};

constexpr int HAB_FAR = -(1 << 28);  // coordinate that fails every bounds test

#define HAB_GENERIC_STORE                                    \
    HAB_HD void store(int m, int n, float v) const {         \
        const EpiRow r__ = epi_row(m);                       \
        const EpiCol c__ = epi_col(n);                       \
        epi_store(r__, c__, epi_fetch(r__, c__), v);         \
    }
#define HAB_NO_KCTX                                          \
    using KCtx = NoCtx;                                      \
    HAB_HD KCtx k_ctx(int, int) const { return KCtx(); }
#define HAB_PLAIN_CVT                                                                                        \
    HAB_HD void a_cvt(const ACtx&, const ARaw& r, int, int, f32x4* out) const { out[0] = sel4(r); }          \
    HAB_HD f32x4 b_cvt(const BRaw& r) const { return sel4(r); }

struct ConvGeom {
    int B, H, W, C;       // input  NHWC
    int Ho, Wo, Cout;     // output NHWC
    int KH, KW, stride, pad;
    FastDiv dHoWo, dWo, dC, dKW, dCout, dHW, dW;
    void finish() {
        Ho = (H + 2 * pad - KH) / stride + 1;
        Wo = (W + 2 * pad - KW) / stride + 1;
        dHoWo = FastDiv(Ho * Wo); dWo = FastDiv(Wo); dC = FastDiv(C); dKW = FastDiv(KW); dCout = FastDiv(Cout);
        dHW = FastDiv(H * W); dW = FastDiv(W);
    }
};

// LDS-DMA staging (igemm_dma.h): block-uniform buffer window of an operand + the index it is relative to.
struct DmaTile { const float* base; uint32_t records; int origin; };
HAB_HD uint32_t dma_records(size_t bytes) { return bytes > 0x7fffffffull ? 0x7fffffffu : (uint32_t)bytes; }

// Validity mask of a KH x KW tap window (bit kh*KW + kw) when the valid taps form the rectangle [r_lo, r_hi) x [c_lo, c_hi):
// (bits of one tap row) * (one bit per valid tap row) -- the product has no carries because the row pattern is < 2^KW.
// Replaces a KH*KW loop of bounds tests in the prologue of every DMA-staged tile (fp32 MFMA shares its issue port with the VALU).
HAB_HD uint32_t tap_rect_mask(int r_lo, int r_hi, int c_lo, int c_hi, int KW) {
    if (r_lo >= r_hi || c_lo >= c_hi) return 0u;
    const uint32_t cols = ((1u << (c_hi - c_lo)) - 1u) << c_lo;   // KW <= 31
    uint32_t rows = 0;
    for (int r = r_lo; r < r_hi; ++r) rows |= 1u << (r * KW);      // <= KH iterations of one OR
    return cols * rows;
}

// A weight matrix row-major [rows][K] consumed as the r-contiguous B operand.
struct WRowCtx { const float* row; int ok; };

// Epilogue pieces shared by the NHWC-output problems.
struct ColN { int n, ok; float bias; };
struct RowBase { size_t base; int ok; };
// Vector epilogue (igemm.h EpiV4): four consecutive columns n .. n+3, n % 4 == 0.  `ok` = all four exist and the output row
// stride keeps them 16-byte aligned; `tail` = a partial quad (N % 4 != 0), handled element-wise.
struct ColN4 { int n, ok, tail; f32x4 bias; };
struct AddMask4 { f32x4 add, mask; };
HAB_HD f32x4 splat4(float x) { f32x4 z; z[0] = x; z[1] = x; z[2] = x; z[3] = x; return z; }
#define HAB_BIAS_RELU_VEC4                                                                                        \
    static constexpr bool EPI_VEC4 = true;                                                                       \
    using EpiCol4 = ColN4;                                                                                       \
    using EpiAux4 = NoAux;                                                                                       \
    HAB_HD EpiCol4 epi_col4(int n) const {                                                                       \
        EpiCol4 c;                                                                                               \
        c.n = n; c.ok = ((N & 3) == 0) & (n + 3 < N); c.tail = (n < N) & !c.ok;                                  \
        c.bias = zero4();                                                                                        \
        if (bias && c.ok) { c.bias[0] = bias[n]; c.bias[1] = bias[n + 1]; c.bias[2] = bias[n + 2]; c.bias[3] = bias[n + 3]; } \
        return c;                                                                                                \
    }                                                                                                            \
    HAB_HD EpiAux4 epi_fetch4(const EpiRow&, const EpiCol4&) const { return EpiAux4(); }                         \
    HAB_HD void epi_store4(const EpiRow& r, const EpiCol4& c, const EpiAux4&, f32x4 v) const {                   \
        if (!r.ok) return;                                                                                       \
        if (c.ok) {                                                                                              \
            v += c.bias;                                                                                         \
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); } \
            if (y) *reinterpret_cast<f32x4*>(y + r.base + c.n) = v;                                              \
        } else if (c.tail) {                                                                                     \
            for (int e = 0; e < 4; ++e) {                                                                        \
                const EpiCol ce = epi_col(c.n + e);                                                              \
                epi_store(r, ce, epi_fetch(r, ce), v[e]);                                                        \
            }                                                                                                    \
        }                                                                                                        \
    }

// ----------------------------------------------------------------------------------------------
// Convolution forward: Y[(img,ho,wo)][co] = sum_{kh,kw,ci} X[img, ho*s-p+kh, wo*s-p+kw, ci] * Wf[co][(kh,kw,ci)]
// Epilogue: + bias[co], optional ReLU.   (simple_cnn.py:68-93, resnet.py:19-34,207-219)
// ----------------------------------------------------------------------------------------------
struct ConvFwdProb {
    static constexpr bool A_RC = true, B_RC = true;
    int M, N, K;
    ConvGeom g;
    const float* x;
    const float* w;
    const float* bias;
    float* y;           // fp32 NHWC output
    int relu;
    int ldy = 0;          // row stride of y in floats (0: N)
    HAB_HD const float* dma_a_origin() const { return x; }
    HAB_HD const float* dma_b_origin() const { return w; }
    HAB_NO_KCTX
    struct ACtx { const float* base; int h0, w0, off0; };  // off0 = (h0*W + w0)*C
    struct AKey { int kh, kw, koff, ok; };                  // koff = (kh*W + kw)*C + ci
    using BCtx = WRowCtx;
    using BKey = KKey;
    using ARaw = Raw4;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int m) const {
        ACtx c;
        if (m >= M) { c.base = x; c.h0 = HAB_FAR; c.w0 = 0; c.off0 = 0; return c; }
        int img, rem, ho, wo;
        g.dHoWo.divmod(m, img, rem);
        g.dWo.divmod(rem, ho, wo);
        c.base = x + (size_t)img * g.H * g.W * g.C;
        c.h0 = ho * g.stride - g.pad;
        c.w0 = wo * g.stride - g.pad;
        c.off0 = (c.h0 * g.W + c.w0) * g.C;
        return c;
    }
    HAB_HD AKey a_key(const KCtx&, int k, int k_end) const {
        AKey q;
        int tap, ci;
        g.dC.divmod(k, tap, ci);
        g.dKW.divmod(tap, q.kh, q.kw);
        q.koff = (q.kh * g.W + q.kw) * g.C + ci;
        q.ok = k < k_end;
        return q;
    }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const {
        const int h = c.h0 + q.kh, w_ = c.w0 + q.kw;
        ARaw r;
        r.ok = q.ok & ((unsigned)h < (unsigned)g.H) & ((unsigned)w_ < (unsigned)g.W);
        r.v = ld4(c.base + ((c.off0 + q.koff) & -r.ok));
        return r;
    }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.ok = n < N; c.row = c.ok ? w + (size_t)n * K : w; return c; }
    HAB_HD BKey b_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const {
        BRaw r;
        r.ok = c.ok & q.ok;
        r.v = ld4(c.row + (q.k & -r.ok));
        return r;
    }
    // ---- LDS-DMA interface (igemm_dma.h): C % 32 == 0 so a K-tile is 32 channels of ONE filter tap ----
    HAB_HD bool dma_ok() const { return (g.C % 32 == 0) && (g.KH * g.KW <= 32) && (g.KW < 32) && ((size_t)g.H * g.W * g.C * 4 * 64 < 0x7fffffffull); }
    HAB_HD DmaTile dma_a_tile(int m0) const {
        DmaTile t;
        t.origin = g.dHoWo.div(m0 < M ? m0 : M - 1);  // first image of the tile
        const size_t shift = (size_t)g.pad * (g.W + 1) * g.C;  // so that padded coordinates are >= 0
        t.base = x + (size_t)t.origin * g.H * g.W * g.C - shift;
        t.records = dma_records(((size_t)(g.B - t.origin) * g.H * g.W * g.C + shift) * 4);
        return t;
    }
    HAB_HD uint32_t dma_a_row(const DmaTile& t, int m, uint32_t& mask) const {
        mask = 0;
        if (m >= M) return 0;
        int img, rem, ho, wo;
        g.dHoWo.divmod(m, img, rem);
        g.dWo.divmod(rem, ho, wo);
        const int h0 = ho * g.stride - g.pad, w0 = wo * g.stride - g.pad;
        // tap (kh, kw) is inside the image for kh in [-h0, H - h0) and kw in [-w0, W - w0), clipped to the kernel
        mask = tap_rect_mask(h0 < 0 ? -h0 : 0, g.H - h0 < g.KH ? g.H - h0 : g.KH, w0 < 0 ? -w0 : 0, g.W - w0 < g.KW ? g.W - w0 : g.KW, g.KW);
        return (uint32_t)((((img - t.origin) * g.H + h0 + g.pad) * g.W + w0 + g.pad) * g.C) * 4u;
    }
    HAB_HD void dma_tap(int k0, int& tap, uint32_t& sa, uint32_t& sb) const {
        int c0, kh, kw;
        g.dC.divmod(k0, tap, c0);
        g.dKW.divmod(tap, kh, kw);
        sa = (uint32_t)((kh * g.W + kw) * g.C + c0) * 4u;
        sb = (uint32_t)k0 * 4u;
    }
    HAB_HD DmaTile dma_b_tile(int n0) const {
        DmaTile t;
        t.origin = n0;
        t.base = w + (size_t)n0 * K;
        t.records = dma_records((size_t)(N - n0) * K * 4);
        return t;
    }
    HAB_HD uint32_t dma_b_row(const DmaTile& t, int n, uint32_t& ok) const { ok = n < N; return (uint32_t)(n - t.origin) * (uint32_t)K * 4u; }
    HAB_PLAIN_CVT
    using EpiCol = ColN;
    using EpiRow = RowBase;
    using EpiAux = NoAux;
    HAB_HD EpiCol epi_col(int n) const { EpiCol c; c.n = n; c.ok = n < N; c.bias = (bias && c.ok) ? bias[n] : 0.f; return c; }
    HAB_HD EpiRow epi_row(int m) const { EpiRow r; r.ok = m < M; r.base = (size_t)m * (ldy > 0 ? ldy : N); return r; }
    HAB_HD EpiAux epi_fetch(const EpiRow&, const EpiCol&) const { return EpiAux(); }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux&, float v) const {
        if (!(r.ok & c.ok)) return;
        v += c.bias;
        if (relu) v = v > 0.f ? v : 0.f;
        if (y) y[r.base + c.n] = v;
    }
    HAB_BIAS_RELU_VEC4
    HAB_GENERIC_STORE
};

// ----------------------------------------------------------------------------------------------
// Observation ingest fused into SimpleCNN's first conv (K1+K4): rgb uint8 NHWC /255.0 and depth
// fp32 NHWC are read in place from the rollout arena (frame -> arena row through `rows`), no
// NCHW / float copy of the observation is ever materialised.  (simple_cnn.py:139-156,68-74)
// Channel order: rgb (3) then depth (1).  CIN = n_rgb + n_depth in {1,3,4}.
// Gather unit = 16 k-elements.  Fast path (`quad`: C == 4, KW % 4 == 0, stride % 4 == 0, pad == 0,
// W % 4 == 0): the unit is four horizontally adjacent taps = 12 aligned bytes of rgb (3 dwords) +
// 16 aligned bytes of depth.  Otherwise every element is gathered on its own (small configs only).
// ----------------------------------------------------------------------------------------------
// x / 255.0f for integer x in [0, 255]: multiply by the reciprocal + one Newton step; equal to the IEEE
// division for all 256 inputs (checked exhaustively in tests/test_hostcheck.py).
HAB_HD float div255(float x) {
    const float rcp = 1.0f / 255.0f;
    const float q = x * rcp;
    const float r = fmaf(-q, 255.0f, x);
    return fmaf(r, rcp, q);
}

struct ObsView {
    const uint8_t* rgb;   // [rows][H][W][3] or null
    const float* depth;   // [rows][H][W][1] or null
    const int* rows;      // frame -> arena row, or null (identity)
    int H, W, C;          // C = 3*(rgb!=0) + (depth!=0)
    HAB_HD float get(int srow, int h, int w, int c) const {
        const size_t pix = ((size_t)srow * H + h) * W + w;
        if (rgb && c < 3) return div255((float)rgb[pix * 3 + c]);
        return depth[pix];
    }
    HAB_HD int srow(int img) const { return rows ? rows[img] : img; }
};

struct ObsRaw { uint32_t d0, d1, d2; f32x4 dep; int ok; };  // ok: 1 valid quad, 0 zero, -1 slow path

HAB_HD void obs_quad_cvt(const ObsRaw& r, f32x4* out) {
    if (!r.ok) { out[0] = zero4(); out[1] = zero4(); out[2] = zero4(); out[3] = zero4(); return; }
    const uint32_t d[3] = {r.d0, r.d1, r.d2};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int i = 3 * q + c;
            out[q][c] = div255((float)((d[i >> 2] >> (8 * (i & 3))) & 0xffu));
        }
        out[q][3] = r.dep[q];
    }
}
// Quad fast path of the fused convolutions: the uint8 channels enter the contraction as the INTEGER value (one
// v_cvt_f32_ubyteN per element) and the 1/255 of `x / 255.0` (simple_cnn.py:143-146) is applied on the other, 8-64x smaller side:
// to the rgb columns of the weight tile when it is staged (forward) or to the rgb rows of dW when they are stored (weight
// gradient).  fp32 MFMA shares the SIMD issue port with the VALU, so the 3 extra VALU ops per gathered element of the exact
// division cost ~15 % of the kernel; the products differ from (x / 255) * w by <= 1.5 ulp each, below the accumulation noise.
constexpr float HAB_RCP255 = 1.0f / 255.0f;
HAB_HD void obs_quad_cvt_raw(const ObsRaw& r, f32x4* out) {
    if (!r.ok) { out[0] = zero4(); out[1] = zero4(); out[2] = zero4(); out[3] = zero4(); return; }
    const uint32_t d[3] = {r.d0, r.d1, r.d2};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int i = 3 * q + c;
            out[q][c] = (float)((d[i >> 2] >> (8 * (i & 3))) & 0xffu);
        }
        out[q][3] = r.dep[q];
    }
}

HAB_HD ObsRaw obs_quad_fetch(const ObsView& obs, int srow, int h, int w, int ok) {
    ObsRaw r;
    r.ok = ok;
    const size_t pix = ok ? ((size_t)srow * obs.H + h) * obs.W + w : 0;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(obs.rgb + pix * 3);
    r.d0 = p[0]; r.d1 = p[1]; r.d2 = p[2];
    r.dep = ld4(obs.depth + pix);
    return r;
}

struct ObsConvFwdProb {
    static constexpr bool A_RC = true, B_RC = true;
    static constexpr int A_KV = 16;
    int M, N, K;
    ConvGeom g;  // g.C = obs.C
    ObsView obs;
    const float* w;
    const float* bias;
    float* y;             // fp32 NHWC output
    int relu;
    int quad;
    HAB_NO_KCTX
    struct ACtx { int srow, h0, w0; };
    struct AKey { int kh, kw, ok; };
    using BCtx = WRowCtx;
    using BKey = KKey;
    using ARaw = ObsRaw;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int m) const {
        ACtx c;
        if (m >= M) { c.srow = 0; c.h0 = HAB_FAR; c.w0 = 0; return c; }
        int img, rem, ho, wo;
        g.dHoWo.divmod(m, img, rem);
        g.dWo.divmod(rem, ho, wo);
        c.srow = obs.srow(img);
        c.h0 = ho * g.stride - g.pad;
        c.w0 = wo * g.stride - g.pad;
        return c;
    }
    HAB_HD AKey a_key(const KCtx&, int k, int k_end) const {
        AKey q;
        g.dKW.divmod(k >> 2, q.kh, q.kw);  // used by the quad path only (C == 4)
        q.ok = k < k_end;
        return q;
    }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const {
        if (quad) {
            const int h = c.h0 + q.kh, w_ = c.w0 + q.kw;
            const int ok = q.ok & ((unsigned)h < (unsigned)g.H) & ((unsigned)w_ < (unsigned)g.W);
            return obs_quad_fetch(obs, c.srow, h, w_, ok);
        }
        ARaw r;
        r.d0 = r.d1 = r.d2 = 0; r.dep = zero4(); r.ok = -1;
        return r;
    }
    HAB_HD void a_cvt(const ACtx& c, const ARaw& r, int k, int k_end, f32x4* out) const {
        if (r.ok >= 0) { obs_quad_cvt_raw(r, out); return; }
        for (int e = 0; e < 16; ++e) {  // slow path: element-wise gather
            const int kk = k + e;
            float v = 0.f;
            if (kk < k_end && c.h0 != HAB_FAR) {
                int tap, ci, kh, kw;
                g.dC.divmod(kk, tap, ci);
                g.dKW.divmod(tap, kh, kw);
                const int h = c.h0 + kh, w_ = c.w0 + kw;
                if ((unsigned)h < (unsigned)g.H && (unsigned)w_ < (unsigned)g.W) v = obs.get(c.srow, h, w_, ci);
            }
            out[e >> 2][e & 3] = v;
        }
    }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.ok = n < N; c.row = c.ok ? w + (size_t)n * K : w; return c; }
    HAB_HD BKey b_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const {
        BRaw r;
        if ((K & 3) == 0) {
            r.ok = c.ok & q.ok;
            r.v = ld4(c.row + (q.k & -r.ok));
        } else {  // K = KH*KW*C not a multiple of 4 (C in {1,3} with odd taps): scalar gathers
            r.ok = 1;
            r.v = zero4();
            for (int e = 0; e < 4; ++e)
                if (c.ok && q.ok && q.k + e < K) r.v[e] = c.row[q.k + e];
        }
        return r;
    }
    HAB_HD f32x4 b_cvt(const BRaw& r) const {
        f32x4 v = sel4(r);
        if (quad) { v[0] *= HAB_RCP255; v[1] *= HAB_RCP255; v[2] *= HAB_RCP255; }  // a unit = channels (r, g, b, depth) of one tap
        return v;
    }
    using EpiCol = ColN;
    using EpiRow = RowBase;
    using EpiAux = NoAux;
    HAB_HD EpiCol epi_col(int n) const { EpiCol c; c.n = n; c.ok = n < N; c.bias = (bias && c.ok) ? bias[n] : 0.f; return c; }
    HAB_HD EpiRow epi_row(int m) const { EpiRow r; r.ok = m < M; r.base = (size_t)m * N; return r; }
    HAB_HD EpiAux epi_fetch(const EpiRow&, const EpiCol&) const { return EpiAux(); }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux&, float v) const {
        if (!(r.ok & c.ok)) return;
        v += c.bias;
        if (relu) v = v > 0.f ? v : 0.f;
        if (y) y[r.base + c.n] = v;
    }
    HAB_BIAS_RELU_VEC4
    HAB_GENERIC_STORE
};

// ----------------------------------------------------------------------------------------------
// Convolution data gradient: dX[(img,h,w)][ci] = sum_{kh,kw,co} dY[img,(h+p-kh)/s,(w+p-kw)/s,co] * Wd[ci][(kh,kw,co)]
// For stride s only taps with kh == (h+p) mod s (and likewise kw) contribute, so the problem is
// split into s*s "stride classes" (ph, pw): class (ph,pw) covers the input pixels with
// (h+p) % s == ph, (w+p) % s == pw and contracts over the taps kh = ph + s*a, kw = pw + s*b only --
// no multiply-by-zero work (4x4/s2: 4 classes x 2x2 taps instead of 16 taps).  One launch per class.
// Epilogue: optional add (residual gradient), optional ReLU mask of the producer's output.
// ----------------------------------------------------------------------------------------------
struct ConvDgradProb {
    static constexpr bool A_RC = true, B_RC = true;
    int M, N, K;  // M = B*Hc*Wc (pixels of this class), N = Cin, K = KHs*KWs*Cout
    ConvGeom g;
    int ph, pw;        // stride class
    int h_first, w_first;  // first h / w of the class
    int Hc, Wc;        // pixels of the class per image along h / w
    int KHs, KWs;      // taps of the class
    int Kfull;         // KH*KW*Cout (row length of Wd)
    FastDiv dHcWc, dWc, dKWs;
    const float* dy;
    const float* w;     // Wd packed [Cin][Kfull]
    const float* mask;  // same shape as dx, or null: dx *= (mask > 0)
    const float* add;   // same shape as dx, or null: dx += add   (applied before the mask)
    float* dx;          // fp32 output
    HAB_HD const float* dma_a_origin() const { return dy; }
    HAB_HD const float* dma_b_origin() const { return w; }
    void set_class(int ph_, int pw_) {
        const int s = g.stride;
        ph = ph_; pw = pw_;
        h_first = ((ph - g.pad) % s + s) % s;
        w_first = ((pw - g.pad) % s + s) % s;
        Hc = h_first < g.H ? (g.H - h_first + s - 1) / s : 0;
        Wc = w_first < g.W ? (g.W - w_first + s - 1) / s : 0;
        KHs = ph < g.KH ? (g.KH - ph + s - 1) / s : 0;
        KWs = pw < g.KW ? (g.KW - pw + s - 1) / s : 0;
        Kfull = g.KH * g.KW * g.Cout;
        M = g.B * Hc * Wc; N = g.C; K = KHs * KWs * g.Cout;
        dHcWc = FastDiv(Hc * Wc > 0 ? Hc * Wc : 1); dWc = FastDiv(Wc > 0 ? Wc : 1); dKWs = FastDiv(KWs > 0 ? KWs : 1);
    }
    HAB_NO_KCTX
    struct ACtx { const float* base; int hq, wq, off0; };  // hq = (h + pad - ph) / s, off0 = (hq*Wo + wq)*Cout
    struct AKey { int a, b, koff, ok; };                    // koff = co - (a*Wo + b)*Cout
    using BCtx = WRowCtx;
    struct BKey { int off, ok; };                           // offset of (tap, co) inside a row of Wd
    using ARaw = Raw4;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int m) const {
        ACtx c;
        if (m >= M) { c.base = dy; c.hq = HAB_FAR; c.wq = 0; c.off0 = 0; return c; }
        int img, rem, hc, wc;
        dHcWc.divmod(m, img, rem);
        dWc.divmod(rem, hc, wc);
        c.base = dy + (size_t)img * g.Ho * g.Wo * g.Cout;
        c.hq = (h_first + hc * g.stride + g.pad - ph) / g.stride;
        c.wq = (w_first + wc * g.stride + g.pad - pw) / g.stride;
        c.off0 = (c.hq * g.Wo + c.wq) * g.Cout;
        return c;
    }
    HAB_HD AKey a_key(const KCtx&, int k, int k_end) const {
        AKey q;
        int tap, co;
        g.dCout.divmod(k, tap, co);
        dKWs.divmod(tap, q.a, q.b);
        q.koff = co - (q.a * g.Wo + q.b) * g.Cout;
        q.ok = k < k_end;
        return q;
    }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const {
        const int hs = c.hq - q.a, ws = c.wq - q.b;  // output row / col that tap (ph + s*a, pw + s*b) reads
        ARaw r;
        r.ok = q.ok & ((unsigned)hs < (unsigned)g.Ho) & ((unsigned)ws < (unsigned)g.Wo);
        r.v = ld4(c.base + ((c.off0 + q.koff) & -r.ok));
        return r;
    }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.ok = n < N; c.row = c.ok ? w + (size_t)n * Kfull : w; return c; }
    HAB_HD BKey b_key(const KCtx&, int k, int k_end) const {
        BKey q;
        int tap, co, a, b;
        g.dCout.divmod(k, tap, co);
        dKWs.divmod(tap, a, b);
        q.off = ((ph + g.stride * a) * g.KW + (pw + g.stride * b)) * g.Cout + co;
        q.ok = k < k_end;
        return q;
    }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const {
        BRaw r;
        r.ok = c.ok & q.ok;
        r.v = ld4(c.row + (q.off & -r.ok));
        return r;
    }
    // ---- LDS-DMA interface (igemm_dma.h): Cout % 32 == 0 so a K-tile is 32 output channels of ONE tap of the class ----
    HAB_HD bool dma_ok() const {
        return (g.Cout % 32 == 0) && (KHs * KWs <= 32) && (KWs < 32) && (K % 32 == 0) && ((size_t)g.Ho * g.Wo * g.Cout * 4 * 64 < 0x7fffffffull);
    }
    HAB_HD DmaTile dma_a_tile(int m0) const {
        DmaTile t;
        t.origin = dHcWc.div(m0 < M ? m0 : M - 1);
        const size_t shift = (size_t)((KHs - 1) * g.Wo + (KWs - 1)) * g.Cout;  // taps read at (hq - a, wq - b): shift so offsets are >= 0
        t.base = dy + (size_t)t.origin * g.Ho * g.Wo * g.Cout - shift;
        t.records = dma_records(((size_t)(g.B - t.origin) * g.Ho * g.Wo * g.Cout + shift) * 4);
        return t;
    }
    HAB_HD uint32_t dma_a_row(const DmaTile& t, int m, uint32_t& mask) const {
        mask = 0;
        if (m >= M) return 0;
        int img, rem, hc, wc;
        dHcWc.divmod(m, img, rem);
        dWc.divmod(rem, hc, wc);
        const int hq = (h_first + hc * g.stride + g.pad - ph) / g.stride, wq = (w_first + wc * g.stride + g.pad - pw) / g.stride;
        // tap (a, b) reads dY row hq - a in [0, Ho): a in (hq - Ho, hq], clipped to the class's taps; likewise b
        mask = tap_rect_mask(hq - g.Ho + 1 > 0 ? hq - g.Ho + 1 : 0, hq + 1 < KHs ? hq + 1 : KHs,
                             wq - g.Wo + 1 > 0 ? wq - g.Wo + 1 : 0, wq + 1 < KWs ? wq + 1 : KWs, KWs);
        return (uint32_t)((((img - t.origin) * g.Ho + hq) * g.Wo + wq) * g.Cout) * 4u;
    }
    HAB_HD void dma_tap(int k0, int& tap, uint32_t& sa, uint32_t& sb) const {
        int co0, a, b;
        g.dCout.divmod(k0, tap, co0);
        dKWs.divmod(tap, a, b);
        sa = (uint32_t)(((KHs - 1 - a) * g.Wo + (KWs - 1 - b)) * g.Cout + co0) * 4u;
        sb = (uint32_t)(((ph + g.stride * a) * g.KW + (pw + g.stride * b)) * g.Cout + co0) * 4u;
    }
    HAB_HD DmaTile dma_b_tile(int n0) const {
        DmaTile t;
        t.origin = n0;
        t.base = w + (size_t)n0 * Kfull;
        t.records = dma_records((size_t)(N - n0) * Kfull * 4);
        return t;
    }
    HAB_HD uint32_t dma_b_row(const DmaTile& t, int n, uint32_t& ok) const { ok = n < N; return (uint32_t)(n - t.origin) * (uint32_t)Kfull * 4u; }
    HAB_PLAIN_CVT
    struct EpiCol { int n, ok; };
    using EpiRow = RowBase;
    struct EpiAux { float add, mask; };
    HAB_HD EpiCol epi_col(int n) const { EpiCol c; c.n = n; c.ok = n < N; return c; }
    HAB_HD EpiRow epi_row(int m) const {
        EpiRow r;
        r.ok = m < M;
        int img, rem, hc, wc;
        dHcWc.divmod(r.ok ? m : 0, img, rem);
        dWc.divmod(rem, hc, wc);
        r.base = ((((size_t)img * g.H) + h_first + hc * g.stride) * g.W + w_first + wc * g.stride) * N;
        return r;
    }
    HAB_HD EpiAux epi_fetch(const EpiRow& r, const EpiCol& c) const {
        EpiAux a;
        const size_t i = (r.ok & c.ok) ? r.base + c.n : 0;
        a.add = add ? add[i] : 0.f;
        a.mask = mask ? mask[i] : 1.f;
        return a;
    }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux& a, float v) const {
        if (!(r.ok & c.ok)) return;
        v += a.add;
        if (!(a.mask > 0.f)) v = 0.f;
        if (dx) dx[r.base + c.n] = v;
    }
    static constexpr bool EPI_VEC4 = true;
    struct EpiCol4 { int n, ok, tail; };
    using EpiAux4 = AddMask4;
    HAB_HD EpiCol4 epi_col4(int n) const { EpiCol4 c; c.n = n; c.ok = ((N & 3) == 0) & (n + 3 < N); c.tail = (n < N) & !c.ok; return c; }
    HAB_HD EpiAux4 epi_fetch4(const EpiRow& r, const EpiCol4& c) const {
        EpiAux4 a;
        const size_t i = (r.ok & c.ok) ? r.base + c.n : 0;
        a.add = add ? ld4(add + i) : zero4();
        a.mask = mask ? ld4(mask + i) : splat4(1.f);
        return a;
    }
    HAB_HD void epi_store4(const EpiRow& r, const EpiCol4& c, const EpiAux4& a, f32x4 v) const {
        if (!r.ok) return;
        if (c.ok) {
            v += a.add;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(a.mask[e] > 0.f)) v[e] = 0.f;
            if (dx) *reinterpret_cast<f32x4*>(dx + r.base + c.n) = v;
        } else if (c.tail) {
            for (int e = 0; e < 4; ++e) {
                const EpiCol ce = epi_col(c.n + e);
                epi_store(r, ce, epi_fetch(r, ce), v[e]);
            }
        }
    }
    HAB_GENERIC_STORE
};

// ----------------------------------------------------------------------------------------------
// Data gradient with the stride classes MERGED into one contraction, for kernels whose size is a multiple of the stride
// (SimpleCNN conv2: 4x4 / 2).  Then every class (ph, pw) has the same KH/s x KW/s taps and the s*s pixels
// h = hq*s - pad + ph, w = wq*s - pad + pw of one quotient position (hq, wq) read the SAME dY neighbourhood
// (hq - a, wq - b): the s*s class problems are one GEMM with
//     rows    m = (img, hq, wq)            M = B * Hq * Wq,  Hq = (H - 1 + pad) / s + 1
//     columns n = (ph, pw, ci)             N = s * s * Cin
//     red.    k = (a, b, co)               K = (KH/s) * (KW/s) * Cout
// i.e. a stride-1 convolution of dY followed by a depth-to-space scatter in the epilogue.  The dY gather is shared by the
// s*s classes (4x less A traffic and LDS fill than the per-class launches) and the tile is 128 columns wide instead of 32.
// LDS-DMA interface only (igemm_dma.h); conv_dgrad falls back to the per-class problems when dma_ok() is false.
// ----------------------------------------------------------------------------------------------
struct ConvDgradMergedProb {
    int M, N, K;
    ConvGeom g;
    int Hq, Wq, KHs, KWs, Kfull;
    FastDiv dHqWq, dWq, dKWs;
    const float* dy;
    const float* w;     // Wd packed [Cin][Kfull = KH*KW*Cout]
    const float* mask;  // same shape as dx, or null: dx *= (mask > 0)
    const float* add;   // same shape as dx, or null: dx += add   (applied before the mask)
    float* dx;          // fp32 output
    HAB_HD const float* dma_a_origin() const { return dy; }
    HAB_HD const float* dma_b_origin() const { return w; }
    static bool applicable(const ConvGeom& g) {
        return g.stride > 1 && g.KH % g.stride == 0 && g.KW % g.stride == 0 && g.Cout % 32 == 0 && g.C % 8 == 0 &&
               g.stride * g.stride * g.C <= 128;
    }
    void finish() {
        const int s = g.stride;
        Hq = (g.H - 1 + g.pad) / s + 1; Wq = (g.W - 1 + g.pad) / s + 1;
        KHs = g.KH / s; KWs = g.KW / s;
        Kfull = g.KH * g.KW * g.Cout;
        M = g.B * Hq * Wq; N = s * s * g.C; K = KHs * KWs * g.Cout;
        dHqWq = FastDiv(Hq * Wq); dWq = FastDiv(Wq); dKWs = FastDiv(KWs);
    }
    // ---- register-staged interface (igemm_bf3.h): both operands r-contiguous (Cout % 32 == 0: a k-quad is 4 channels of one tap) ----
    static constexpr bool A_RC = true, B_RC = true;
    HAB_NO_KCTX
    struct ACtx { const float* base; int hq, wq; };   // base = dY of the row's image
    struct AKey { int a, b, co, ok; };
    using BCtx = WRowCtx;
    struct BKey { int koff, ok; };
    using ARaw = Raw4;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int m) const {
        ACtx c;
        if (m >= M) { c.base = dy; c.hq = HAB_FAR; c.wq = 0; return c; }
        int img, rem;
        dHqWq.divmod(m, img, rem);
        dWq.divmod(rem, c.hq, c.wq);
        c.base = dy + (size_t)img * g.Ho * g.Wo * g.Cout;
        return c;
    }
    HAB_HD AKey a_key(const KCtx&, int k, int k_end) const {
        AKey q;
        int tap;
        g.dCout.divmod(k, tap, q.co);
        dKWs.divmod(tap, q.a, q.b);
        q.ok = k < k_end;
        return q;
    }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const {
        const int h = c.hq - q.a, w_ = c.wq - q.b;
        ARaw r;
        r.ok = q.ok & ((unsigned)h < (unsigned)g.Ho) & ((unsigned)w_ < (unsigned)g.Wo);
        r.v = ld4(c.base + (((h * g.Wo + w_) * g.Cout + q.co) & -r.ok));
        return r;
    }
    HAB_HD BCtx b_ctx(int n) const {
        BCtx c;
        c.ok = n < N;
        int pp, ci;
        g.dC.divmod(c.ok ? n : 0, pp, ci);
        const int ph = pp / g.stride, pw = pp - ph * g.stride;
        c.row = w + (size_t)ci * Kfull + (ph * g.KW + pw) * g.Cout;
        return c;
    }
    HAB_HD BKey b_key(const KCtx&, int k, int k_end) const {
        BKey q;
        int tap, co, a, b;
        g.dCout.divmod(k, tap, co);
        dKWs.divmod(tap, a, b);
        q.koff = (g.stride * a * g.KW + g.stride * b) * g.Cout + co;
        q.ok = k < k_end;
        return q;
    }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const {
        BRaw r;
        r.ok = c.ok & q.ok;
        r.v = ld4(c.row + (q.koff & -r.ok));
        return r;
    }
    HAB_PLAIN_CVT
    HAB_HD bool dma_ok() const { return (KHs * KWs <= 32) && (KWs < 32) && ((size_t)g.Ho * g.Wo * g.Cout * 4 * 64 < 0x7fffffffull); }
    HAB_HD DmaTile dma_a_tile(int m0) const {
        DmaTile t;
        t.origin = dHqWq.div(m0 < M ? m0 : M - 1);
        const size_t shift = (size_t)((KHs - 1) * g.Wo + (KWs - 1)) * g.Cout;  // taps read at (hq - a, wq - b): shift so offsets are >= 0
        t.base = dy + (size_t)t.origin * g.Ho * g.Wo * g.Cout - shift;
        t.records = dma_records(((size_t)(g.B - t.origin) * g.Ho * g.Wo * g.Cout + shift) * 4);
        return t;
    }
    HAB_HD uint32_t dma_a_row(const DmaTile& t, int m, uint32_t& mask_) const {
        mask_ = 0;
        if (m >= M) return 0;
        int img, rem, hq, wq;
        dHqWq.divmod(m, img, rem);
        dWq.divmod(rem, hq, wq);
        mask_ = tap_rect_mask(hq - g.Ho + 1 > 0 ? hq - g.Ho + 1 : 0, hq + 1 < KHs ? hq + 1 : KHs,
                              wq - g.Wo + 1 > 0 ? wq - g.Wo + 1 : 0, wq + 1 < KWs ? wq + 1 : KWs, KWs);
        return (uint32_t)((((img - t.origin) * g.Ho + hq) * g.Wo + wq) * g.Cout) * 4u;
    }
    HAB_HD void dma_tap(int k0, int& tap, uint32_t& sa, uint32_t& sb) const {
        int co0, a, b;
        g.dCout.divmod(k0, tap, co0);
        dKWs.divmod(tap, a, b);
        sa = (uint32_t)(((KHs - 1 - a) * g.Wo + (KWs - 1 - b)) * g.Cout + co0) * 4u;
        sb = (uint32_t)((g.stride * a * g.KW + g.stride * b) * g.Cout + co0) * 4u;
    }
    HAB_HD DmaTile dma_b_tile(int) const {
        DmaTile t;
        t.origin = 0;
        t.base = w;
        t.records = dma_records((size_t)g.C * Kfull * 4);
        return t;
    }
    HAB_HD uint32_t dma_b_row(const DmaTile&, int n, uint32_t& ok) const {
        ok = n < N;
        int pp, ci;
        g.dC.divmod(ok ? n : 0, pp, ci);
        const int ph = pp / g.stride, pw = pp - ph * g.stride;
        return (uint32_t)(ci * Kfull + (ph * g.KW + pw) * g.Cout) * 4u;
    }
    struct EpiCol { int off, ph, pw, ok; };        // off = (ph*W + pw)*Cin + ci
    struct EpiRow { long long base; int hb, wb, ok; };  // hb = hq*s - pad, base = ((img*H + hb)*W + wb)*Cin
    struct EpiAux { float add, mask; };
    HAB_HD EpiCol epi_col(int n) const {
        EpiCol c;
        c.ok = n < N;
        int pp, ci;
        g.dC.divmod(c.ok ? n : 0, pp, ci);
        c.ph = pp / g.stride; c.pw = pp - c.ph * g.stride;
        c.off = (c.ph * g.W + c.pw) * g.C + ci;
        return c;
    }
    HAB_HD EpiRow epi_row(int m) const {
        EpiRow r;
        r.ok = m < M;
        int img, rem, hq, wq;
        dHqWq.divmod(r.ok ? m : 0, img, rem);
        dWq.divmod(rem, hq, wq);
        r.hb = hq * g.stride - g.pad; r.wb = wq * g.stride - g.pad;
        r.base = (((long long)img * g.H + r.hb) * g.W + r.wb) * g.C;
        return r;
    }
    HAB_HD bool epi_ok(const EpiRow& r, const EpiCol& c) const {
        return r.ok & c.ok & ((unsigned)(r.hb + c.ph) < (unsigned)g.H) & ((unsigned)(r.wb + c.pw) < (unsigned)g.W);
    }
    HAB_HD EpiAux epi_fetch(const EpiRow& r, const EpiCol& c) const {
        EpiAux a;
        const long long i = epi_ok(r, c) ? r.base + c.off : 0;
        a.add = add ? add[i] : 0.f;
        a.mask = mask ? mask[i] : 1.f;
        return a;
    }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux& a, float v) const {
        if (!epi_ok(r, c)) return;
        v += a.add;
        if (!(a.mask > 0.f)) v = 0.f;
        if (dx) dx[r.base + c.off] = v;
    }
    static constexpr bool EPI_VEC4 = true;  // Cin % 8 == 0: a quad of columns is 4 channels of one (ph, pw) class
    using EpiCol4 = EpiCol;
    using EpiAux4 = AddMask4;
    HAB_HD EpiCol4 epi_col4(int n) const { return epi_col(n); }
    HAB_HD EpiAux4 epi_fetch4(const EpiRow& r, const EpiCol4& c) const {
        EpiAux4 a;
        const long long i = epi_ok(r, c) ? r.base + c.off : 0;
        a.add = add ? ld4(add + i) : zero4();
        a.mask = mask ? ld4(mask + i) : splat4(1.f);
        return a;
    }
    HAB_HD void epi_store4(const EpiRow& r, const EpiCol4& c, const EpiAux4& a, f32x4 v) const {
        if (!epi_ok(r, c)) return;
        v += a.add;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!(a.mask[e] > 0.f)) v[e] = 0.f;
        if (dx) *reinterpret_cast<f32x4*>(dx + r.base + c.off) = v;
    }
    HAB_GENERIC_STORE
};

// ----------------------------------------------------------------------------------------------
// Convolution weight gradient: dW[co][ci][kh][kw] = sum_{img,ho,wo} X[img,ho*s-p+kh,wo*s-p+kw,ci] * dY[img,ho,wo,co]
// GEMM view: i = (kh,kw,ci) (M), j = co (N), reduction r = (img,ho,wo) (K).  Both operands are
// i/j-contiguous.  The result is scattered into the reference OIHW layout.
// ----------------------------------------------------------------------------------------------
struct OihwRow { int off, ok; };  // offset of (ci,kh,kw) inside one output channel's OIHW block
struct ColOnly { int n, ok; };

struct ConvWgradProb {
    static constexpr bool A_RC = false, B_RC = false;
    static constexpr bool COLSUM_B = true;
    int M, N, K;
    ConvGeom g;
    const float* x;
    const float* dy;
    float* dw;      // OIHW with Creal input channels (Creal <= g.C: the stem conv reads a zero-padded input)
    float* colsum;  // bias gradient [Cout] (sum of dY over all pixels), or null
    int Creal;
    HAB_HD void store_colsum(int j, float v) const { colsum[j] = v; }
    HAB_NO_KCTX
    struct ACtx { int kh, kw, ioff; };                 // ioff = (kh*W + kw)*C + ci
    struct AKey { size_t roff; int hb, wb, ok; };      // pixel r: hb = ho*s - p, wb = wo*s - p, roff = ((img*H + hb)*W + wb)*C
    struct BCtx { int co; };
    struct BKey { size_t roff; int ok; };              // r * N
    using ARaw = Raw4;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int i) const {
        ACtx c;
        if (i >= M) { c.kh = HAB_FAR; c.kw = 0; c.ioff = 0; return c; }
        int tap, ci;
        g.dC.divmod(i, tap, ci);
        g.dKW.divmod(tap, c.kh, c.kw);
        c.ioff = (c.kh * g.W + c.kw) * g.C + ci;
        return c;
    }
    HAB_HD AKey a_key(const KCtx&, int r, int k_end) const {
        AKey q;
        int img, rem, ho, wo;
        g.dHoWo.divmod(r, img, rem);
        g.dWo.divmod(rem, ho, wo);
        q.hb = ho * g.stride - g.pad;
        q.wb = wo * g.stride - g.pad;
        q.roff = (size_t)(((long long)img * g.H + q.hb) * g.W + q.wb) * (size_t)g.C;
        q.ok = r < k_end;
        return q;
    }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const {
        const int h = q.hb + c.kh, w_ = q.wb + c.kw;
        ARaw v;
        v.ok = q.ok & ((unsigned)h < (unsigned)g.H) & ((unsigned)w_ < (unsigned)g.W);
        const size_t off = (q.roff + (size_t)(long long)c.ioff) & (size_t)(-(long long)v.ok);
        v.v = ld4(x + off);
        return v;
    }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.co = (j < N) ? j : -1; return c; }
    HAB_HD BKey b_key(const KCtx&, int r, int k_end) const { BKey q; q.roff = (size_t)r * N; q.ok = r < k_end; return q; }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const {
        BRaw v;
        v.ok = (c.co >= 0) & q.ok;
        v.v = ld4(dy + ((q.roff + (size_t)c.co) & (size_t)(-(long long)v.ok)));
        return v;
    }
    HAB_PLAIN_CVT
    using EpiCol = ColOnly;
    using EpiRow = OihwRow;
    using EpiAux = NoAux;
    HAB_HD EpiCol epi_col(int n) const { EpiCol c; c.n = n; c.ok = n < N; return c; }
    HAB_HD EpiRow epi_row(int i) const {
        EpiRow r;
        r.ok = i < M;
        int tap, ci, kh, kw;
        g.dC.divmod(r.ok ? i : 0, tap, ci);
        g.dKW.divmod(tap, kh, kw);
        r.ok &= ci < Creal;
        r.off = (ci * g.KH + kh) * g.KW + kw;
        return r;
    }
    HAB_HD EpiAux epi_fetch(const EpiRow&, const EpiCol&) const { return EpiAux(); }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux&, float v) const {
        if (!(r.ok & c.ok)) return;
        dw[(size_t)c.n * Creal * g.KH * g.KW + r.off] = v;
    }
    HAB_GENERIC_STORE
};

// Weight gradient of the observation-ingest conv.  Gather unit of A = 16 consecutive i = (kh, kw..kw+3, ci)
// at one output pixel r.  One K-tile (32 pixels) spans at most two frames when Ho*Wo >= 32, so the
// frame -> arena row lookups are block-uniform (KCtx).
struct ObsConvWgradProb {
    static constexpr bool A_RC = false, B_RC = false;
    static constexpr bool COLSUM_B = true;
    static constexpr int A_KV = 16;
    int M, N, K;
    ConvGeom g;
    ObsView obs;
    const float* dy;
    float* dw;
    float* colsum;  // bias gradient [Cout], or null
    int quad;
    HAB_HD void store_colsum(int j, float v) const { colsum[j] = v; }
    struct KCtx { int img0, srow0, srow1; };
    HAB_HD KCtx k_ctx(int k0, int) const {
        KCtx c;
        c.img0 = g.dHoWo.div(k0 < K ? k0 : K - 1);
        c.srow0 = obs.srow(c.img0);
        c.srow1 = obs.srow(c.img0 + 1 < g.B ? c.img0 + 1 : c.img0);
        return c;
    }
    struct ACtx { int i, kh, kw; };
    struct AKey { int srow, hb, wb, ok; };
    struct BCtx { int co; };
    struct BKey { size_t roff; int ok; };
    using ARaw = ObsRaw;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int i) const {
        ACtx c;
        c.i = i;
        if (i >= M) { c.kh = HAB_FAR; c.kw = 0; return c; }
        g.dKW.divmod(i >> 2, c.kh, c.kw);  // used by the quad path only (C == 4)
        return c;
    }
    HAB_HD int srow_of(const KCtx& kc, int img) const {
        if (g.Ho * g.Wo >= IGEMM_BK_) return img == kc.img0 ? kc.srow0 : kc.srow1;
        return obs.srow(img < g.B ? img : g.B - 1);
    }
    static constexpr int IGEMM_BK_ = 32;
    HAB_HD AKey a_key(const KCtx& kc, int r, int k_end) const {
        AKey q;
        int img, rem, ho, wo;
        g.dHoWo.divmod(r, img, rem);
        g.dWo.divmod(rem, ho, wo);
        q.hb = ho * g.stride - g.pad;
        q.wb = wo * g.stride - g.pad;
        q.ok = r < k_end;
        q.srow = quad ? srow_of(kc, img) : 0;
        return q;
    }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const {
        if (quad) {
            const int h = q.hb + c.kh, w_ = q.wb + c.kw;
            const int ok = q.ok & ((unsigned)h < (unsigned)g.H) & ((unsigned)w_ < (unsigned)g.W);
            return obs_quad_fetch(obs, q.srow, h, w_, ok);
        }
        ARaw v;
        v.d0 = v.d1 = v.d2 = 0; v.dep = zero4(); v.ok = -1;
        return v;
    }
    HAB_HD void a_cvt(const ACtx& c, const ARaw& q, int r, int k_end, f32x4* out) const {
        if (q.ok >= 0) { obs_quad_cvt_raw(q, out); return; }
        int img = 0, rem = 0, ho = 0, wo = 0, srow = 0;
        const bool rv = r < k_end;
        if (rv) {
            g.dHoWo.divmod(r, img, rem);
            g.dWo.divmod(rem, ho, wo);
            srow = obs.srow(img);
        }
        for (int e = 0; e < 16; ++e) {
            const int ii = c.i + e;
            float v = 0.f;
            if (rv && ii < M) {
                int tap, ci, kh, kw;
                g.dC.divmod(ii, tap, ci);
                g.dKW.divmod(tap, kh, kw);
                const int h = ho * g.stride - g.pad + kh, w_ = wo * g.stride - g.pad + kw;
                if ((unsigned)h < (unsigned)g.H && (unsigned)w_ < (unsigned)g.W) v = obs.get(srow, h, w_, ci);
            }
            out[e >> 2][e & 3] = v;
        }
    }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.co = (j < N) ? j : -1; return c; }
    HAB_HD BKey b_key(const KCtx&, int r, int k_end) const { BKey q; q.roff = (size_t)r * N; q.ok = r < k_end; return q; }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const {
        BRaw v;
        v.ok = (c.co >= 0) & q.ok;
        v.v = ld4(dy + ((q.roff + (size_t)c.co) & (size_t)(-(long long)v.ok)));
        return v;
    }
    HAB_HD f32x4 b_cvt(const BRaw& r) const { return sel4(r); }
    using EpiCol = ColOnly;
    struct EpiRow { int off, ok; float scale; };  // scale: 1/255 for the rgb rows of the quad path (integer-valued gather)
    using EpiAux = NoAux;
    HAB_HD EpiCol epi_col(int n) const { EpiCol c; c.n = n; c.ok = n < N; return c; }
    HAB_HD EpiRow epi_row(int i) const {
        EpiRow r;
        r.ok = i < M;
        int tap, ci, kh, kw;
        g.dC.divmod(r.ok ? i : 0, tap, ci);
        g.dKW.divmod(tap, kh, kw);
        r.off = (ci * g.KH + kh) * g.KW + kw;
        r.scale = (quad && ci < 3) ? HAB_RCP255 : 1.0f;
        return r;
    }
    HAB_HD EpiAux epi_fetch(const EpiRow&, const EpiCol&) const { return EpiAux(); }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux&, float v) const {
        if (!(r.ok & c.ok)) return;
        dw[(size_t)c.n * g.C * g.KH * g.KW + r.off] = v * r.scale;
    }
    HAB_GENERIC_STORE
};

// ----------------------------------------------------------------------------------------------
// Linear layers.  x: [M][K] (ldx), w: [N][K] (ldw) as torch.nn.Linear stores it.
// vec = 1 requires ldx, ldw, K multiples of 4 and 16-byte aligned bases; vec = 0 gathers scalars.
// ----------------------------------------------------------------------------------------------
HAB_HD Raw4 row_fetch4(const float* row, int row_ok, const KKey& q, int K, int vec) {
    Raw4 r;
    if (vec) {
        r.ok = row_ok & q.ok;
        r.v = ld4(row + (q.k & -r.ok));
    } else {  // scalar gathers (K or the leading dimension not a multiple of 4)
        r.ok = 1;
        r.v = zero4();
        for (int e = 0; e < 4; ++e)
            if (row_ok && q.ok && q.k + e < K) r.v[e] = row[q.k + e];
    }
    return r;
}
// four consecutive columns j..j+3 of row-major `base` (ld) at row r; n = number of columns
HAB_HD Raw4 col_fetch4(const float* base, int ld, int r, int r_ok, int j, int n) {
    Raw4 q;
    if (((ld & 3) == 0) && (j + 3 < n || j >= n)) {
        q.ok = r_ok & (j < n);
        q.v = ld4(base + (((size_t)r * ld + j) & (size_t)(-(long long)q.ok)));
    } else {
        q.ok = 1;
        q.v = zero4();
        for (int e = 0; e < 4; ++e)
            if (r_ok && j + e < n) q.v[e] = base[(size_t)r * ld + j + e];
    }
    return q;
}

// Y = X W^T + b (+ReLU), written with row stride ldy (lets a layer write into a concat buffer).
struct LinearFwdProb {
    static constexpr bool A_RC = true, B_RC = true;
    int M, N, K;
    const float* x; int ldx;
    const float* w; int ldw;
    const float* bias;
    float* y; int ldy;
    int relu, vec;
    int accumulate;  // y += (used for the second operand of a fused two-input projection)
    HAB_NO_KCTX
    using ACtx = WRowCtx;
    using BCtx = WRowCtx;
    using ARaw = Raw4;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int m) const { ACtx c; c.ok = m < M; c.row = c.ok ? x + (size_t)m * ldx : x; return c; }
    using AKey = KKey;
    using BKey = KKey;
    HAB_HD AKey a_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD BKey b_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const { return row_fetch4(c.row, c.ok, q, K, vec); }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.ok = n < N; c.row = c.ok ? w + (size_t)n * ldw : w; return c; }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const { return row_fetch4(c.row, c.ok, q, K, vec); }
    HAB_PLAIN_CVT
    using EpiCol = ColN;
    using EpiRow = RowBase;
    struct EpiAux { float old; };
    HAB_HD EpiCol epi_col(int n) const { EpiCol c; c.n = n; c.ok = n < N; c.bias = (bias && c.ok) ? bias[n] : 0.f; return c; }
    HAB_HD EpiRow epi_row(int m) const { EpiRow r; r.ok = m < M; r.base = (size_t)m * ldy; return r; }
    HAB_HD EpiAux epi_fetch(const EpiRow& r, const EpiCol& c) const {
        EpiAux a;
        a.old = accumulate ? y[(r.ok & c.ok) ? r.base + c.n : 0] : 0.f;
        return a;
    }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux& a, float v) const {
        if (!(r.ok & c.ok)) return;
        v += c.bias;
        v += a.old;
        if (relu) v = v > 0.f ? v : 0.f;
        y[r.base + c.n] = v;
    }
    HAB_GENERIC_STORE
};

// dX[m][k] = sum_n dY[m][n] W[n][k]   (M = rows, N = in-features, reduction over out-features)
struct LinearDgradProb {
    static constexpr bool A_RC = true, B_RC = false;
    int M, N, K;
    const float* dy; int lddy;
    const float* w; int ldw;
    const float* mask; int ldmask;  // dx *= (mask > 0) for columns < mask_cols, or null
    int mask_cols;
    float* dx; int lddx;
    int vec_a;
    int accumulate;
    HAB_NO_KCTX
    using ACtx = WRowCtx;
    struct BCtx { int j; };
    using ARaw = Raw4;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int m) const { ACtx c; c.ok = m < M; c.row = c.ok ? dy + (size_t)m * lddy : dy; return c; }
    using AKey = KKey;
    using BKey = KKey;
    HAB_HD AKey a_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD BKey b_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const { return row_fetch4(c.row, c.ok, q, K, vec_a); }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.j = j; return c; }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const { return col_fetch4(w, ldw, q.k, q.ok, c.j, N); }
    HAB_PLAIN_CVT
    using EpiCol = ColOnly;
    struct EpiRow { size_t base, mbase; int ok; };
    struct EpiAux { float old, mask; };
    HAB_HD EpiCol epi_col(int n) const { EpiCol c; c.n = n; c.ok = n < N; return c; }
    HAB_HD EpiRow epi_row(int m) const { EpiRow r; r.ok = m < M; r.base = (size_t)m * lddx; r.mbase = (size_t)m * ldmask; return r; }
    HAB_HD EpiAux epi_fetch(const EpiRow& r, const EpiCol& c) const {
        EpiAux a;
        const int ok = r.ok & c.ok;
        a.old = accumulate ? dx[ok ? r.base + c.n : 0] : 0.f;
        a.mask = (mask && c.n < mask_cols) ? mask[ok ? r.mbase + c.n : 0] : 1.f;
        return a;
    }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux& a, float v) const {
        if (!(r.ok & c.ok)) return;
        v += a.old;
        if (!(a.mask > 0.f)) v = 0.f;
        dx[r.base + c.n] = v;
    }
    HAB_GENERIC_STORE
};

// dW[n][k] = sum_m dY[m][n] X[m][k]   (M = out-features, N = in-features, reduction over rows).
// perm_c > 0: X columns are in NHWC-flatten order (hw*C + c) while the reference weight expects
// NCHW-flatten (c*HW + hw) -- nn.Flatten after an NCHW conv (simple_cnn.py:91, resnet_policy.py:588).
struct LinearWgradProb {
    static constexpr bool A_RC = false, B_RC = false;
    int M, N, K;
    const float* dy; int lddy;
    const float* x; int ldx;
    float* dw; int lddw;
    int perm_c, perm_hw;
    FastDiv dPermC;
    int accumulate;
    HAB_NO_KCTX
    struct ACtx { int i; };
    struct BCtx { int j; };
    using ARaw = Raw4;
    using BRaw = Raw4;
    HAB_HD ACtx a_ctx(int i) const { ACtx c; c.i = i; return c; }
    using AKey = KKey;
    using BKey = KKey;
    HAB_HD AKey a_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD BKey b_key(const KCtx&, int k, int k_end) const { return kkey(k, k_end); }
    HAB_HD ARaw a_fetch(const ACtx& c, const KCtx&, const AKey& q) const { return col_fetch4(dy, lddy, q.k, q.ok, c.i, M); }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.j = j; return c; }
    HAB_HD BRaw b_fetch(const BCtx& c, const KCtx&, const BKey& q) const { return col_fetch4(x, ldx, q.k, q.ok, c.j, N); }
    HAB_PLAIN_CVT
    struct EpiCol { int col, ok; };
    using EpiRow = RowBase;
    struct EpiAux { float old; };
    HAB_HD EpiCol epi_col(int j) const {
        EpiCol c;
        c.ok = j < N;
        c.col = j;
        if (perm_c > 0) {
            int hw, cc;
            dPermC.divmod(c.ok ? j : 0, hw, cc);
            c.col = cc * perm_hw + hw;
        }
        return c;
    }
    HAB_HD EpiRow epi_row(int i) const { EpiRow r; r.ok = i < M; r.base = (size_t)i * lddw; return r; }
    HAB_HD EpiAux epi_fetch(const EpiRow& r, const EpiCol& c) const {
        EpiAux a;
        a.old = accumulate ? dw[(r.ok & c.ok) ? r.base + c.col : 0] : 0.f;
        return a;
    }
    HAB_HD void epi_store(const EpiRow& r, const EpiCol& c, const EpiAux& a, float v) const {
        if (!(r.ok & c.ok)) return;
        dw[r.base + c.col] = v + a.old;
    }
    HAB_GENERIC_STORE
};

}  // namespace hab
