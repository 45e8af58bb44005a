// prob_build.h -- construction of igemm problem descriptors from layer descriptions.  Shared by the
// device launchers (gemm_ops.hip) and the CPU index-math checker (tests/hostcheck/hostcheck.hip).
#pragma once
#include "ops.h"
#include "problems.h"

namespace hab {

inline ConvGeom make_geom(const ConvDesc& d) {
    ConvGeom g;
    g.B = d.B; g.H = d.H; g.W = d.W; g.C = d.C; g.Cout = d.Cout; g.KH = d.KH; g.KW = d.KW; g.stride = d.stride; g.pad = d.pad;
    g.finish();
    return g;
}

inline int check_conv(const ConvDesc& d) {
    if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.C <= 0 || d.Cout <= 0 || d.KH <= 0 || d.KW <= 0 || d.stride <= 0 || d.pad < 0)
        return HAB_ERR_ARG;
    if ((long long)d.B * d.H * d.W >= (1ll << 31)) return HAB_ERR_UNSUPPORTED;
    if ((d.H + 2 * d.pad - d.KH) < 0 || (d.W + 2 * d.pad - d.KW) < 0) return HAB_ERR_ARG;
    return HAB_OK;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Conditions of the 4-tap observation gather (problems.h): four horizontally adjacent taps start on a
// pixel index that is a multiple of 4, so their rgb bytes are 3 aligned dwords and their depth one aligned float4.
inline int obs_quad_ok(const ConvDesc& d, const ObsView& obs) {
    return obs.rgb && obs.depth && obs.C == 4 && d.KW % 4 == 0 && d.stride % 4 == 0 && d.pad == 0 && d.W % 4 == 0 &&
           ((uintptr_t)obs.rgb & 3) == 0 && aligned16(obs.depth);
}

inline int build(ConvFwdProb& p, const ConvDesc& d, const float* x, const float* wf, const float* bias, float* y, int relu) {
    HAB_TRY(check_conv(d));
    if (d.C % 4) return HAB_ERR_UNSUPPORTED;
    p.g = make_geom(d);
    p.M = d.B * p.g.Ho * p.g.Wo; p.N = d.Cout; p.K = d.KH * d.KW * d.C;
    p.x = x; p.w = wf; p.bias = bias; p.y = y; p.relu = relu;
    return HAB_OK;
}
inline int build(ObsConvFwdProb& p, const ConvDesc& d, const ObsView& obs, const float* wf, const float* bias, float* y, int relu) {
    HAB_TRY(check_conv(d));
    if (d.C != obs.C) return HAB_ERR_UNSUPPORTED;
    p.g = make_geom(d);
    p.M = d.B * p.g.Ho * p.g.Wo; p.N = d.Cout; p.K = d.KH * d.KW * d.C;
    p.obs = obs; p.w = wf; p.bias = bias; p.y = y; p.relu = relu;
    p.quad = obs_quad_ok(d, obs);
    return HAB_OK;
}
inline int build(ConvDgradProb& p, const ConvDesc& d, const float* dy, const float* wd, const float* mask, const float* add,
                 float* dx, int ph = 0, int pw = 0) {
    HAB_TRY(check_conv(d));
    if (d.C % 4 || d.Cout % 4) return HAB_ERR_UNSUPPORTED;
    if (ph < 0 || pw < 0 || ph >= d.stride || pw >= d.stride) return HAB_ERR_ARG;
    p.g = make_geom(d);
    p.dy = dy; p.w = wd; p.mask = mask; p.add = add; p.dx = dx;
    p.set_class(ph, pw);
    return HAB_OK;
}
inline int build(ConvWgradProb& p, const ConvDesc& d, const float* x, const float* dy, float* dw, float* dbias = nullptr) {
    HAB_TRY(check_conv(d));
    if (d.C % 4 || d.Cout % 4) return HAB_ERR_UNSUPPORTED;
    p.g = make_geom(d);
    p.M = d.KH * d.KW * d.C; p.N = d.Cout; p.K = d.B * p.g.Ho * p.g.Wo;
    p.x = x; p.dy = dy; p.dw = dw; p.colsum = dbias;
    p.Creal = d.Creal > 0 ? d.Creal : d.C;
    return HAB_OK;
}
inline int build(ObsConvWgradProb& p, const ConvDesc& d, const ObsView& obs, const float* dy, float* dw, float* dbias = nullptr) {
    HAB_TRY(check_conv(d));
    if (d.C != obs.C || d.Cout % 4) return HAB_ERR_UNSUPPORTED;
    p.g = make_geom(d);
    p.M = d.KH * d.KW * d.C; p.N = d.Cout; p.K = d.B * p.g.Ho * p.g.Wo;
    p.obs = obs; p.dy = dy; p.dw = dw; p.colsum = dbias;
    p.quad = obs_quad_ok(d, obs);
    return HAB_OK;
}
inline int build(LinearFwdProb& p, const float* x, int ldx, const float* w, int ldw, const float* bias, float* y, int ldy,
                 int M, int N, int K, int relu, int accumulate) {
    if (M <= 0 || N <= 0 || K <= 0 || !x || !w || !y) return HAB_ERR_ARG;
    p.M = M; p.N = N; p.K = K; p.x = x; p.ldx = ldx; p.w = w; p.ldw = ldw; p.bias = bias; p.y = y; p.ldy = ldy;
    p.relu = relu; p.accumulate = accumulate;
    p.vec = (ldx % 4 == 0 && ldw % 4 == 0 && K % 4 == 0 && aligned16(x) && aligned16(w)) ? 1 : 0;
    return HAB_OK;
}
inline int build(LinearDgradProb& p, const float* dy, int lddy, const float* w, int ldw, const float* mask, int ldmask,
                 int mask_cols, float* dx, int lddx, int M, int Nin, int Kout, int accumulate) {
    if (M <= 0 || Nin <= 0 || Kout <= 0 || !dy || !w || !dx) return HAB_ERR_ARG;
    if (!aligned16(w)) return HAB_ERR_ARG;
    p.M = M; p.N = Nin; p.K = Kout; p.dy = dy; p.lddy = lddy; p.w = w; p.ldw = ldw; p.mask = mask; p.ldmask = ldmask;
    p.mask_cols = mask_cols; p.dx = dx; p.lddx = lddx; p.accumulate = accumulate;
    p.vec_a = (lddy % 4 == 0 && Kout % 4 == 0 && aligned16(dy)) ? 1 : 0;
    return HAB_OK;
}
inline int build(LinearWgradProb& p, const float* dy, int lddy, const float* x, int ldx, float* dw, int lddw, int Mrows,
                 int Nout, int Kin, int perm_c, int perm_hw, int accumulate) {
    if (Mrows <= 0 || Nout <= 0 || Kin <= 0 || !dy || !x || !dw) return HAB_ERR_ARG;
    if (!aligned16(dy) || !aligned16(x)) return HAB_ERR_ARG;
    p.M = Nout; p.N = Kin; p.K = Mrows; p.dy = dy; p.lddy = lddy; p.x = x; p.ldx = ldx; p.dw = dw; p.lddw = lddw;
    p.perm_c = perm_c; p.perm_hw = perm_hw; p.dPermC = FastDiv(perm_c > 0 ? perm_c : 1); p.accumulate = accumulate;
    return HAB_OK;
}

}  // namespace hab
