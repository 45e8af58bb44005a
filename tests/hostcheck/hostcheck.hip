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

// ---------------------------------------------------------------------------------------------------------------------
// Host emulation of the LDS-DMA staged kernels (igemm_dma.h): the SAME dma_a_tile / dma_a_row / dma_tap / dma_b_* functors produce
// the buffer window, the per-row byte offset + tap-validity mask and the per-K-tile scalar offsets; a gather is
// `window_base + row_offset + tap_offset + 16 * quad`, zero when the tap's mask bit is clear or the offset falls outside the
// window (the hardware's buffer range check).  The epilogue goes through the vector (epi_*4) interface the kernels use, with the
// transposed accumulator's quad order.  tile_rows = BM of the kernel (the window is per M-tile).
// ---------------------------------------------------------------------------------------------------------------------
template <class P>
static float dma_load(const DmaTile& t, uint32_t off) {  // 4-byte element of a raw buffer load with range check
    if (off + 4 > t.records) return 0.f;
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(t.base) + off);
}
template <class P>
static void host_dma_igemm(const P& p, int tile_rows, int tile_cols) {
    constexpr uint32_t OOB = 0x80000000u;
    const int ntk = (p.K + IGEMM_BK - 1) / IGEMM_BK;
    std::vector<float> acc((size_t)4);
    for (int m0 = 0; m0 < p.M; m0 += tile_rows) {
        const DmaTile ta = p.dma_a_tile(m0);
        for (int n0 = 0; n0 < p.N; n0 += tile_cols) {
            const DmaTile tb = p.dma_b_tile(n0);
            for (int m = m0; m < std::min(p.M, m0 + tile_rows); ++m) {
                uint32_t amask;
                const uint32_t aoff = p.dma_a_row(ta, m, amask);
                const typename P::EpiRow er = p.epi_row(m);
                for (int nq = n0; nq < std::min(p.N, n0 + tile_cols); nq += 4) {
                    double s[4] = {0, 0, 0, 0};
                    for (int e = 0; e < 4; ++e) {
                        const int n = nq + e;
                        uint32_t bok;
                        uint32_t boff = p.dma_b_row(tb, n, bok);
                        if (!bok) boff = OOB;
                        for (int kt = 0; kt < ntk; ++kt) {
                            int tap;
                            uint32_t sa, sb;
                            p.dma_tap(kt * IGEMM_BK, tap, sa, sb);
                            const bool a_ok = (amask >> tap) & 1u;
                            for (int kk = 0; kk < IGEMM_BK && kt * IGEMM_BK + kk < p.K; ++kk) {
                                const float a = a_ok ? dma_load<P>(ta, aoff + sa + 4u * kk) : 0.f;
                                const float b = boff == OOB ? 0.f : dma_load<P>(tb, boff + sb + 4u * kk);
                                s[e] += (double)a * (double)b;
                            }
                        }
                    }
                    const typename P::EpiCol4 ec = p.epi_col4(nq);
                    f32x4 v;
                    v[0] = (float)s[0]; v[1] = (float)s[1]; v[2] = (float)s[2]; v[3] = (float)s[3];
                    p.epi_store4(er, ec, p.epi_fetch4(er, ec), v);
                }
            }
        }
    }
}

extern "C" int hc_conv2d_fwd_dma(const float* x, const float* wf, const float* bias, float* y, int B, int H, int W, int C, int Cout,
                                 int KH, int KW, int stride, int pad, int relu) {
    ConvFwdProb p;
    HAB_TRY(build(p, mk(B, H, W, C, Cout, KH, KW, stride, pad), x, wf, bias, y, relu));
    if (!p.dma_ok()) return -1;
    host_dma_igemm(p, 128, 64);
    return 0;
}
// per-class (merged = 0) or merged-stride-class (merged = 1) data gradient through the DMA interface
extern "C" int hc_conv2d_dgrad_dma(const float* dy, const float* wd, const float* mask, const float* add, float* dx, int B, int H,
                                   int W, int C, int Cout, int KH, int KW, int stride, int pad, int merged) {
    const ConvDesc d = mk(B, H, W, C, Cout, KH, KW, stride, pad);
    if (merged) {
        ConvDgradMergedProb q;
        HAB_TRY(check_conv(d));
        q.g = make_geom(d);
        if (!ConvDgradMergedProb::applicable(q.g)) return -2;
        q.dy = dy; q.w = wd; q.mask = mask; q.add = add; q.dx = dx;
        q.finish();
        if (merged == 2) {  // register-staged interface of the merged problem (the split-bf16 kernel's gathers)
            host_igemm(q);
            return 0;
        }
        if (!q.dma_ok()) return -1;
        host_dma_igemm(q, 128, 128);
        return 0;
    }
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            ConvDgradProb p;
            HAB_TRY(build(p, d, dy, wd, mask, add, dx, ph, pw));
            if (p.Hc <= 0 || p.Wc <= 0) continue;
            if (p.K <= 0) {
                for (int m = 0; m < p.M; ++m)
                    for (int n = 0; n < p.N; ++n) p.store(m, n, 0.f);
                continue;
            }
            if (!p.dma_ok()) return -1;
            host_dma_igemm(p, 256, 32);
        }
    return 0;
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
