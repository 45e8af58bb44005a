// rnn_gates.h -- the element-wise cell arithmetic of the recurrent encoder, shared by the step-per-launch kernels (rnn.hip) and the
// persistent kernels (rnn_persist.h).  Floating-point contraction is OFF inside these functions: with it on, whether a*b + c becomes one
// fused multiply-add is the optimiser's choice per inlining site, and the two kernel families would differ in the last bit; with it off the
// results are fixed by the expressions, and the two families are bit-identical (tests/test_gpu_rnn_persist.py).
// Reference: torch's GRU / LSTM cells as used by rl/models/rnn_state_encoder.py:385-420.
#pragma once
#include "hab_common.h"

namespace hab {

// GRU forward: pre-activations gi (input side, incl. b_ih), gh (hidden side, incl. b_hh) of gates r, z, n; hp = state entering the step
__device__ __forceinline__ float gru_cell_fwd(const float gi0, const float gi1, const float gi2, const float gh0, const float gh1, const float gh2,
                                              const float hp, float& rg, float& zg, float& ng) {
#pragma clang fp contract(off)
    rg = sigmoidf_(gi0 + gh0);
    zg = sigmoidf_(gi1 + gh1);
    const float rn = rg * gh2;
    ng = tanhf(gi2 + rn);
    const float a = (1.0f - zg) * ng;
    const float b = zg * hp;
    return a + b;
}
// LSTM forward: gates i, f, g, o from the summed pre-activations; returns h', writes c'
__device__ __forceinline__ float lstm_cell_fwd(const float p0, const float p1, const float p2, const float p3, const float cp, float& ig, float& fg,
                                               float& gg, float& og, float& cn) {
#pragma clang fp contract(off)
    ig = sigmoidf_(p0);
    fg = sigmoidf_(p1);
    gg = tanhf(p2);
    og = sigmoidf_(p3);
    const float a = fg * cp;
    const float b = ig * gg;
    cn = a + b;
    return og * tanhf(cn);
}
// GRU backward of one element: dh = gradient wrt h' -> pre-activation gradients (input side: dr, dz, dn; hidden side n: dn * r) and the
// direct term dh * z for the step before
__device__ __forceinline__ void gru_cell_bwd(const float dh, const float rg, const float z, const float n, const float hp, const float hn, float& dr_pre,
                                             float& dz_pre, float& dn_pre, float& dhn_pre, float& direct) {
#pragma clang fp contract(off)
    const float dn = dh * (1.0f - z);
    const float dz = dh * (hp - n);
    const float nn = n * n;
    dn_pre = dn * (1.0f - nn);
    const float dr = dn_pre * hn;
    dr_pre = (dr * rg) * (1.0f - rg);
    dz_pre = (dz * z) * (1.0f - z);
    dhn_pre = dn_pre * rg;
    direct = dh * z;
}
// LSTM backward of one element: dh, the cell carry dcc from the step after (added only when `carried`), saved gates and cells ->
// pre-activation gradients of i, f, g, o and the cell carry for the step before
__device__ __forceinline__ void lstm_cell_bwd(const float dh, const bool carried, const float dcc, const float ig, const float fg, const float gg,
                                              const float og, const float cn, const float cp, float& di_pre, float& df_pre, float& dg_pre,
                                              float& do_pre, float& carry) {
#pragma clang fp contract(off)
    const float tc = tanhf(cn);
    const float tc2 = tc * tc;
    float dc = (dh * og) * (1.0f - tc2);
    if (carried) dc = dc + dcc;
    const float d_o = dh * tc;
    const float di = dc * gg, dg = dc * ig, df = dc * cp;
    di_pre = (di * ig) * (1.0f - ig);
    df_pre = (df * fg) * (1.0f - fg);
    const float g2 = gg * gg;
    dg_pre = dg * (1.0f - g2);
    do_pre = (d_o * og) * (1.0f - og);
    carry = dc * fg;
}

}  // namespace hab
