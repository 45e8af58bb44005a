// problems.h -- operand gathers + epilogues that turn the layers of the policy into instances of
// the igemm kernel (igemm.h).  Activations are NHWC fp32 (channels % 4 == 0 so a 16-byte gather never
// straddles a filter tap); convolution weights are consumed from packed copies
//   Wf[co][(kh,kw,ci)]  (forward)      Wd[ci][(kh,kw,co)]  (data gradient)
// produced by repack.hip from the reference-layout (OIHW) master parameters; weight gradients are
// scattered straight back into OIHW.  All functors are __host__ __device__ so that
// tests/hostcheck can execute the very same index math on the CPU against the oracle.
#pragma once
#include "hab_common.h"

namespace hab {

#define HAB_HD __host__ __device__ inline

HAB_HD f32x4 zero4() {
    f32x4 z;
    z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
    return z;
}
HAB_HD f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

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
    float* y;
    int relu;
    struct ACtx { const float* base; int h0, w0; };
    struct BCtx { const float* row; };
    HAB_HD ACtx a_ctx(int m) const {
        ACtx c;
        if (m >= M) { c.base = nullptr; c.h0 = 0; c.w0 = 0; return c; }
        int img, rem, ho, wo;
        g.dHoWo.divmod(m, img, rem);
        g.dWo.divmod(rem, ho, wo);
        c.base = x + (size_t)img * g.H * g.W * g.C;
        c.h0 = ho * g.stride - g.pad;
        c.w0 = wo * g.stride - g.pad;
        return c;
    }
    HAB_HD f32x4 a_load(const ACtx& c, int k, int k_end) const {
        if (!c.base || k >= k_end) return zero4();
        int tap, ci, kh, kw;
        g.dC.divmod(k, tap, ci);
        g.dKW.divmod(tap, kh, kw);
        const int h = c.h0 + kh, w_ = c.w0 + kw;
        if ((unsigned)h >= (unsigned)g.H || (unsigned)w_ >= (unsigned)g.W) return zero4();
        return ld4(c.base + ((size_t)h * g.W + w_) * g.C + ci);
    }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.row = (n < N) ? w + (size_t)n * K : nullptr; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int k, int k_end) const {
        if (!c.row || k >= k_end) return zero4();
        return ld4(c.row + k);
    }
    HAB_HD void store(int m, int n, float v) const {
        if (bias) v += bias[n];
        if (relu) v = v > 0.f ? v : 0.f;
        y[(size_t)m * N + n] = v;
    }
};

// ----------------------------------------------------------------------------------------------
// Observation ingest fused into SimpleCNN's first conv (K1+K4): rgb uint8 NHWC /255.0 and depth
// fp32 NHWC are read in place from the rollout arena (frame -> arena row through `rows`), no
// NCHW / float copy of the observation is ever materialised.  (simple_cnn.py:139-156,68-74)
// Channel order: rgb (3) then depth (1).  CIN = n_rgb + n_depth in {1,3,4}.
// ----------------------------------------------------------------------------------------------
struct ObsView {
    const uint8_t* rgb;   // [rows][H][W][3] or null
    const float* depth;   // [rows][H][W][1] or null
    const int* rows;      // frame -> arena row, or null (identity)
    int H, W, C;          // C = 3*(rgb!=0) + (depth!=0)
    HAB_HD float get(int srow, int h, int w, int c) const {
        const size_t pix = ((size_t)srow * H + h) * W + w;
        if (rgb && c < 3) return (float)rgb[pix * 3 + c] / 255.0f;
        return depth[pix];
    }
    HAB_HD f32x4 get4_rgbd(int srow, int h, int w) const {  // C == 4 fast path
        const size_t pix = ((size_t)srow * H + h) * W + w;
        const uint8_t* p = rgb + pix * 3;
        f32x4 r;
        r[0] = (float)p[0] / 255.0f; r[1] = (float)p[1] / 255.0f; r[2] = (float)p[2] / 255.0f; r[3] = depth[pix];
        return r;
    }
};

struct ObsConvFwdProb {
    static constexpr bool A_RC = true, B_RC = true;
    int M, N, K;
    ConvGeom g;  // g.C = obs.C
    ObsView obs;
    const float* w;
    const float* bias;
    float* y;
    int relu;
    struct ACtx { int srow, h0, w0; };
    struct BCtx { const float* row; };
    HAB_HD ACtx a_ctx(int m) const {
        ACtx c;
        if (m >= M) { c.srow = -1; c.h0 = 0; c.w0 = 0; return c; }
        int img, rem, ho, wo;
        g.dHoWo.divmod(m, img, rem);
        g.dWo.divmod(rem, ho, wo);
        c.srow = obs.rows ? obs.rows[img] : img;
        c.h0 = ho * g.stride - g.pad;
        c.w0 = wo * g.stride - g.pad;
        return c;
    }
    HAB_HD f32x4 a_load(const ACtx& c, int k, int k_end) const {
        if (c.srow < 0 || k >= k_end) return zero4();
        if (g.C == 4) {
            int kh, kw;
            g.dKW.divmod(k >> 2, kh, kw);
            const int h = c.h0 + kh, w_ = c.w0 + kw;
            if ((unsigned)h >= (unsigned)g.H || (unsigned)w_ >= (unsigned)g.W) return zero4();
            return obs.get4_rgbd(c.srow, h, w_);
        }
        f32x4 r = zero4();
        for (int e = 0; e < 4; ++e) {
            const int kk = k + e;
            if (kk >= k_end) break;
            int tap, ci, kh, kw;
            g.dC.divmod(kk, tap, ci);
            g.dKW.divmod(tap, kh, kw);
            const int h = c.h0 + kh, w_ = c.w0 + kw;
            if ((unsigned)h < (unsigned)g.H && (unsigned)w_ < (unsigned)g.W) r[e] = obs.get(c.srow, h, w_, ci);
        }
        return r;
    }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.row = (n < N) ? w + (size_t)n * K : nullptr; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int k, int k_end) const {
        if (!c.row || k >= k_end) return zero4();
        return ld4(c.row + k);
    }
    HAB_HD void store(int m, int n, float v) const {
        if (bias) v += bias[n];
        if (relu) v = v > 0.f ? v : 0.f;
        y[(size_t)m * N + n] = v;
    }
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
    float* dx;
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
    struct ACtx { const float* base; int hq, wq; };  // hq = (h + pad - ph) / s
    struct BCtx { const float* row; };
    HAB_HD ACtx a_ctx(int m) const {
        ACtx c;
        if (m >= M) { c.base = nullptr; c.hq = 0; c.wq = 0; return c; }
        int img, rem, hc, wc;
        dHcWc.divmod(m, img, rem);
        dWc.divmod(rem, hc, wc);
        c.base = dy + (size_t)img * g.Ho * g.Wo * g.Cout;
        c.hq = (h_first + hc * g.stride + g.pad - ph) / g.stride;
        c.wq = (w_first + wc * g.stride + g.pad - pw) / g.stride;
        return c;
    }
    HAB_HD f32x4 a_load(const ACtx& c, int k, int k_end) const {
        if (!c.base || k >= k_end) return zero4();
        int tap, co, a, b;
        g.dCout.divmod(k, tap, co);
        dKWs.divmod(tap, a, b);
        const int hs = c.hq - a, ws = c.wq - b;  // output row / col that tap (ph + s*a, pw + s*b) reads
        if ((unsigned)hs >= (unsigned)g.Ho || (unsigned)ws >= (unsigned)g.Wo) return zero4();
        return ld4(c.base + ((size_t)hs * g.Wo + ws) * g.Cout + co);
    }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.row = (n < N) ? w + (size_t)n * Kfull : nullptr; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int k, int k_end) const {
        if (!c.row || k >= k_end) return zero4();
        int tap, co, a, b;
        g.dCout.divmod(k, tap, co);
        dKWs.divmod(tap, a, b);
        return ld4(c.row + ((size_t)(ph + g.stride * a) * g.KW + (pw + g.stride * b)) * g.Cout + co);
    }
    HAB_HD void store(int m, int n, float v) const {
        int img, rem, hc, wc;
        dHcWc.divmod(m, img, rem);
        dWc.divmod(rem, hc, wc);
        const size_t i = ((((size_t)img * g.H) + h_first + hc * g.stride) * g.W + w_first + wc * g.stride) * N + n;
        if (add) v += add[i];
        if (mask && !(mask[i] > 0.f)) v = 0.f;
        dx[i] = v;
    }
};

// ----------------------------------------------------------------------------------------------
// Convolution weight gradient: dW[co][ci][kh][kw] = sum_{img,ho,wo} X[img,ho*s-p+kh,wo*s-p+kw,ci] * dY[img,ho,wo,co]
// GEMM view: i = (kh,kw,ci) (M), j = co (N), reduction r = (img,ho,wo) (K).  Both operands are
// i/j-contiguous.  The result is scattered into the reference OIHW layout.
// ----------------------------------------------------------------------------------------------
struct ConvWgradProb {
    static constexpr bool A_RC = false, B_RC = false;
    static constexpr bool COLSUM_B = true;
    int M, N, K;
    ConvGeom g;
    const float* x;
    const float* dy;
    float* dw;      // OIHW
    float* colsum;  // bias gradient [Cout] (sum of dY over all pixels), or null
    HAB_HD void store_colsum(int j, float v) const { colsum[j] = v; }
    struct ACtx { int kh, kw, ci; };
    struct BCtx { int co; };
    HAB_HD ACtx a_ctx(int i) const {
        ACtx c;
        if (i >= M) { c.kh = -1; c.kw = 0; c.ci = 0; return c; }
        int tap;
        g.dC.divmod(i, tap, c.ci);
        g.dKW.divmod(tap, c.kh, c.kw);
        return c;
    }
    HAB_HD f32x4 a_load(const ACtx& c, int r, int k_end) const {
        if (c.kh < 0 || r >= k_end) return zero4();
        int img, rem, ho, wo;
        g.dHoWo.divmod(r, img, rem);
        g.dWo.divmod(rem, ho, wo);
        const int h = ho * g.stride - g.pad + c.kh, w_ = wo * g.stride - g.pad + c.kw;
        if ((unsigned)h >= (unsigned)g.H || (unsigned)w_ >= (unsigned)g.W) return zero4();
        return ld4(x + (((size_t)img * g.H + h) * g.W + w_) * g.C + c.ci);
    }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.co = (j < N) ? j : -1; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int r, int k_end) const {
        if (c.co < 0 || r >= k_end) return zero4();
        return ld4(dy + (size_t)r * N + c.co);
    }
    HAB_HD void store(int i, int j, float v) const {
        int tap, ci, kh, kw;
        g.dC.divmod(i, tap, ci);
        g.dKW.divmod(tap, kh, kw);
        dw[(((size_t)j * g.C + ci) * g.KH + kh) * g.KW + kw] = v;
    }
};

struct ObsConvWgradProb {
    static constexpr bool A_RC = false, B_RC = false;
    static constexpr bool COLSUM_B = true;
    int M, N, K;
    ConvGeom g;
    ObsView obs;
    const float* dy;
    float* dw;
    float* colsum;  // bias gradient [Cout], or null
    HAB_HD void store_colsum(int j, float v) const { colsum[j] = v; }
    struct ACtx { int i; };
    struct BCtx { int co; };
    HAB_HD ACtx a_ctx(int i) const { ACtx c; c.i = (i < M) ? i : -1; return c; }
    HAB_HD f32x4 a_load(const ACtx& c, int r, int k_end) const {
        if (c.i < 0 || r >= k_end) return zero4();
        int img, rem, ho, wo;
        g.dHoWo.divmod(r, img, rem);
        g.dWo.divmod(rem, ho, wo);
        const int srow = obs.rows ? obs.rows[img] : img;
        const int h0 = ho * g.stride - g.pad, w0 = wo * g.stride - g.pad;
        if (g.C == 4) {
            int kh, kw;
            g.dKW.divmod(c.i >> 2, kh, kw);
            const int h = h0 + kh, w_ = w0 + kw;
            if ((unsigned)h >= (unsigned)g.H || (unsigned)w_ >= (unsigned)g.W) return zero4();
            return obs.get4_rgbd(srow, h, w_);
        }
        f32x4 q = zero4();
        for (int e = 0; e < 4; ++e) {
            const int ii = c.i + e;
            if (ii >= M) break;
            int tap, ci, kh, kw;
            g.dC.divmod(ii, tap, ci);
            g.dKW.divmod(tap, kh, kw);
            const int h = h0 + kh, w_ = w0 + kw;
            if ((unsigned)h < (unsigned)g.H && (unsigned)w_ < (unsigned)g.W) q[e] = obs.get(srow, h, w_, ci);
        }
        return q;
    }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.co = (j < N) ? j : -1; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int r, int k_end) const {
        if (c.co < 0 || r >= k_end) return zero4();
        return ld4(dy + (size_t)r * N + c.co);
    }
    HAB_HD void store(int i, int j, float v) const {
        int tap, ci, kh, kw;
        g.dC.divmod(i, tap, ci);
        g.dKW.divmod(tap, kh, kw);
        dw[(((size_t)j * g.C + ci) * g.KH + kh) * g.KW + kw] = v;
    }
};

// ----------------------------------------------------------------------------------------------
// Linear layers.  x: [M][K] (ldx), w: [N][K] (ldw) as torch.nn.Linear stores it.
// vec = 1 requires ldx, ldw, K multiples of 4 and 16-byte aligned bases; vec = 0 gathers scalars.
// ----------------------------------------------------------------------------------------------
HAB_HD f32x4 row_load4(const float* row, int k, int k_end, int vec) {
    if (vec) return (k < k_end) ? ld4(row + k) : zero4();
    f32x4 r = zero4();
    for (int e = 0; e < 4; ++e)
        if (k + e < k_end) r[e] = row[k + e];
    return r;
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
    struct ACtx { const float* row; };
    struct BCtx { const float* row; };
    HAB_HD ACtx a_ctx(int m) const { ACtx c; c.row = (m < M) ? x + (size_t)m * ldx : nullptr; return c; }
    HAB_HD f32x4 a_load(const ACtx& c, int k, int k_end) const { return c.row ? row_load4(c.row, k, k_end, vec) : zero4(); }
    HAB_HD BCtx b_ctx(int n) const { BCtx c; c.row = (n < N) ? w + (size_t)n * ldw : nullptr; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int k, int k_end) const { return c.row ? row_load4(c.row, k, k_end, vec) : zero4(); }
    HAB_HD void store(int m, int n, float v) const {
        float* o = y + (size_t)m * ldy + n;
        if (bias) v += bias[n];
        if (accumulate) v += *o;
        if (relu) v = v > 0.f ? v : 0.f;
        *o = v;
    }
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
    struct ACtx { const float* row; };
    struct BCtx { int j; };
    HAB_HD ACtx a_ctx(int m) const { ACtx c; c.row = (m < M) ? dy + (size_t)m * lddy : nullptr; return c; }
    HAB_HD f32x4 a_load(const ACtx& c, int k, int k_end) const { return c.row ? row_load4(c.row, k, k_end, vec_a) : zero4(); }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.j = j; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int r, int k_end) const {
        if (r >= k_end || c.j >= N) return zero4();
        const float* p = w + (size_t)r * ldw + c.j;
        if (c.j + 3 < N && ((ldw & 3) == 0)) return ld4(p);
        f32x4 q = zero4();
        for (int e = 0; e < 4; ++e)
            if (c.j + e < N) q[e] = p[e];
        return q;
    }
    HAB_HD void store(int m, int n, float v) const {
        float* o = dx + (size_t)m * lddx + n;
        if (accumulate) v += *o;
        if (mask && n < mask_cols && !(mask[(size_t)m * ldmask + n] > 0.f)) v = 0.f;
        *o = v;
    }
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
    struct ACtx { int i; };
    struct BCtx { int j; };
    HAB_HD ACtx a_ctx(int i) const { ACtx c; c.i = i; return c; }
    HAB_HD f32x4 a_load(const ACtx& c, int r, int k_end) const {
        if (r >= k_end || c.i >= M) return zero4();
        const float* p = dy + (size_t)r * lddy + c.i;
        if (c.i + 3 < M && ((lddy & 3) == 0)) return ld4(p);
        f32x4 q = zero4();
        for (int e = 0; e < 4; ++e)
            if (c.i + e < M) q[e] = p[e];
        return q;
    }
    HAB_HD BCtx b_ctx(int j) const { BCtx c; c.j = j; return c; }
    HAB_HD f32x4 b_load(const BCtx& c, int r, int k_end) const {
        if (r >= k_end || c.j >= N) return zero4();
        const float* p = x + (size_t)r * ldx + c.j;
        if (c.j + 3 < N && ((ldx & 3) == 0)) return ld4(p);
        f32x4 q = zero4();
        for (int e = 0; e < 4; ++e)
            if (c.j + e < N) q[e] = p[e];
        return q;
    }
    HAB_HD void store(int i, int j, float v) const {
        int col = j;
        if (perm_c > 0) {
            int hw, c;
            dPermC.divmod(j, hw, c);
            col = c * perm_hw + hw;
        }
        float* o = dw + (size_t)i * lddw + col;
        if (accumulate) v += *o;
        *o = v;
    }
};

}  // namespace hab
