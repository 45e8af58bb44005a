// TEST INFRASTRUCTURE ONLY -- CPU executor for the igemm problem functors.
//
// The product kernels split every contraction into (a) gather/epilogue functors (problems.h,
// prob_build.h: im2col index math, packed-weight addressing, OIHW scatter, NHWC<->NCHW flatten
// permutation, masks) and (b) the shared MFMA tile machinery (igemm.h).  (a) is `__host__
// __device__`, so this file runs the *same functors* in a plain triple loop on host memory; the
// CPU test-suite compares the result with the oracle (tests/test_hostcheck.py).  That pins all the
// index math without a GPU; the GPU tests then only have to establish the tile machinery.
// Built into tests/hostcheck/libhab_hostcheck.so; never loaded by the product package.
#include <algorithm>
#include <vector>

#include "../../habitat-lab_amd/csrc/igemm.h"
#include "../../habitat-lab_amd/csrc/prob_build.h"

using namespace hab;

// Element (m, k) of the A operand / (k, n) of the B operand, through the same fetch + cvt functors the kernel uses.
template <class P>
static float host_a(const P& p, int m, int k) {
    constexpr int KV = AKv<P>::value;
    f32x4 out[KV / 4];
    const typename P::KCtx kc = p.k_ctx(k - k % IGEMM_BK, p.K);
    if constexpr (P::A_RC) {
        const int kb = k - k % KV;
        const typename P::ACtx c = p.a_ctx(m);
        p.a_cvt(c, p.a_fetch(c, kc, p.a_key(kc, kb, p.K)), kb, p.K, out);
        return out[(k - kb) >> 2][(k - kb) & 3];
    } else {
        const int mb = m - m % KV;
        const typename P::ACtx c = p.a_ctx(mb);
        p.a_cvt(c, p.a_fetch(c, kc, p.a_key(kc, k, p.K)), k, p.K, out);
        return out[(m - mb) >> 2][(m - mb) & 3];
    }
}
template <class P>
static float host_b(const P& p, int k, int n) {
    const typename P::KCtx kc = p.k_ctx(k - k % IGEMM_BK, p.K);
    if constexpr (P::B_RC) {
        const int kb = k & ~3;
        const f32x4 v = p.b_cvt(p.b_fetch(p.b_ctx(n), kc, p.b_key(kc, kb, p.K)));
        return v[k - kb];
    } else {
        const int nb = n & ~3;
        const f32x4 v = p.b_cvt(p.b_fetch(p.b_ctx(nb), kc, p.b_key(kc, k, p.K)));
        return v[n - nb];
    }
}

template <class P>
static void host_igemm(const P& p) {
    std::vector<float> bm((size_t)p.K * p.N);
    for (int k = 0; k < p.K; ++k)
        for (int n = 0; n < p.N; ++n) bm[(size_t)k * p.N + n] = host_b(p, k, n);
    std::vector<double> acc((size_t)p.N);
    for (int m = 0; m < p.M; ++m) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int k = 0; k < p.K; ++k) {
            const double a = (double)host_a(p, m, k);
            if (a == 0.0) continue;
            const float* br = &bm[(size_t)k * p.N];
            for (int n = 0; n < p.N; ++n) acc[n] += a * (double)br[n];
        }
        for (int n = 0; n < p.N; ++n) p.store(m, n, (float)acc[n]);
    }
}

// exhaustive check of the uint8 scaling used by the observation gathers: returns the number of mismatches
extern "C" int hc_div255_mismatches() {
    int bad = 0;
    for (int x = 0; x < 256; ++x) {
        volatile float ref = (float)x / 255.0f;
        if (div255((float)x) != ref) ++bad;
    }
    return bad;
}

static ConvDesc mk(int B, int H, int W, int C, int Cout, int KH, int KW, int stride, int pad) {
    ConvDesc d;
    d.B = B; d.H = H; d.W = W; d.C = C; d.Cout = Cout; d.KH = KH; d.KW = KW; d.stride = stride; d.pad = pad;
    return d;
}
static ObsView mkobs(const uint8_t* rgb, const float* depth, const int* rows, int H, int W) {
    ObsView o;
    o.rgb = rgb; o.depth = depth; o.rows = rows; o.H = H; o.W = W; o.C = (rgb ? 3 : 0) + (depth ? 1 : 0);
    return o;
}

extern "C" int hc_conv2d_fwd(const float* x, const float* wf, const float* bias, float* y, int B, int H, int W, int C, int Cout,
                             int KH, int KW, int stride, int pad, int relu) {
    ConvFwdProb p;
    HAB_TRY(build(p, mk(B, H, W, C, Cout, KH, KW, stride, pad), x, wf, bias, y, relu));
    host_igemm(p);
    return 0;
}
extern "C" int hc_obs_conv2d_fwd(const uint8_t* rgb, const float* depth, const int* rows, const float* wf, const float* bias,
                                 float* y, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu) {
    ObsView o = mkobs(rgb, depth, rows, H, W);
    ObsConvFwdProb p;
    HAB_TRY(build(p, mk(B, H, W, o.C, Cout, KH, KW, stride, pad), o, wf, bias, y, relu));
    host_igemm(p);
    return 0;
}
extern "C" int hc_conv2d_dgrad(const float* dy, const float* wd, const float* mask, const float* add, float* dx, int B, int H,
                               int W, int C, int Cout, int KH, int KW, int stride, int pad) {
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            ConvDgradProb p;
            HAB_TRY(build(p, mk(B, H, W, C, Cout, KH, KW, stride, pad), dy, wd, mask, add, dx, ph, pw));
            if (p.Hc <= 0 || p.Wc <= 0) continue;
            if (p.K <= 0) {
                for (int m = 0; m < p.M; ++m)
                    for (int n = 0; n < p.N; ++n) p.store(m, n, 0.f);
                continue;
            }
            host_igemm(p);
        }
    return 0;
}
extern "C" int hc_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int C, int Cout, int KH, int KW,
                               int stride, int pad) {
    ConvWgradProb p;
    HAB_TRY(build(p, mk(B, H, W, C, Cout, KH, KW, stride, pad), x, dy, dw, nullptr));
    host_igemm(p);
    return 0;
}
extern "C" int hc_obs_conv2d_wgrad(const uint8_t* rgb, const float* depth, const int* rows, const float* dy, float* dw, int B,
                                   int H, int W, int Cout, int KH, int KW, int stride, int pad) {
    ObsView o = mkobs(rgb, depth, rows, H, W);
    ObsConvWgradProb p;
    HAB_TRY(build(p, mk(B, H, W, o.C, Cout, KH, KW, stride, pad), o, dy, dw, nullptr));
    host_igemm(p);
    return 0;
}
extern "C" int hc_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* y, int ldy, int M, int N,
                             int K, int relu, int accumulate) {
    LinearFwdProb p;
    HAB_TRY(build(p, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, accumulate));
    host_igemm(p);
    return 0;
}
extern "C" int hc_linear_dgrad(const float* dy, int lddy, const float* w, int ldw, const float* mask, int ldmask, int mask_cols,
                               float* dx, int lddx, int M, int n_in, int n_out, int accumulate) {
    LinearDgradProb p;
    HAB_TRY(build(p, dy, lddy, w, ldw, mask, ldmask, mask_cols, dx, lddx, M, n_in, n_out, accumulate));
    host_igemm(p);
    return 0;
}
extern "C" int hc_linear_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int lddw, int M, int n_out, int n_in,
                               int perm_c, int perm_hw, int accumulate) {
    LinearWgradProb p;
    HAB_TRY(build(p, dy, lddy, x, ldx, dw, lddw, M, n_out, n_in, perm_c, perm_hw, accumulate));
    host_igemm(p);
    return 0;
}

// Host restatements of the repack kernels' index formulas (gemm_ops.hip) for the same purpose.
extern "C" void hc_repack_conv(const float* w, float* wf, float* wd, int Cout, int Cin, int KH, int KW, int cpad) {
    for (int co = 0; co < Cout; ++co)
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw)
                for (int ci = 0; ci < cpad; ++ci) {
                    const float v = ci < Cin ? w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
                    if (wf) wf[(((size_t)co * KH + kh) * KW + kw) * cpad + ci] = v;
                    if (wd && ci < Cin) wd[(((size_t)ci * KH + kh) * KW + kw) * Cout + co] = v;
                }
}
extern "C" void hc_repack_flatten(const float* w, float* wp, int N, int C, int HW) {
    for (int n = 0; n < N; ++n)
        for (int hw = 0; hw < HW; ++hw)
            for (int c = 0; c < C; ++c) wp[((size_t)n * HW + hw) * C + c] = w[((size_t)n * C + c) * HW + hw];
}
