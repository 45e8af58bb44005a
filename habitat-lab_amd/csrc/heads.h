// heads.h -- argument blocks of the head / glue kernels (heads.hip).
#pragma once
#include "hab_common.h"

namespace hab {

struct HeadsArgs {
    int B, H, A, mode;           // mode 0 evaluate, 1 sample (exp noise), 2 deterministic
    const float* feats; int feats_ld;  // [B][feats_ld], first H columns (feats_ld = 0 -> H)
    const float* w_actor; const float* b_actor;    // [A][H], [A]
    const float* w_critic; const float* b_critic;  // [1][H], [1]
    const int64_t* actions_in;   // mode 0: arena (rows,1) gathered through rows
    const int* rows;
    const float* noise;          // mode 1: [B][A]
    int64_t* actions_out;        // modes 1/2: [B]
    float* value; float* logp; float* entropy;  // [B]
    float* probs; float* logits_n;              // [B][8] saved for backward / sampling checks (may be null)
};

struct HeadsBwdArgs {
    int B, H, A;
    const float* d_value; const float* d_logp; const float* d_entropy;  // [B]
    const int64_t* actions; const int* rows;
    const float* probs; const float* logits_n;  // [B][8]
    const float* w_actor; const float* w_critic;
    float* dfeat;   // [B][H]
    float* dzv;     // [B][8]  d logits (cols < A), zero padded
    float* dv_out;  // [B]
};

// Gaussian head (GaussianNet + CustomNormal, utils/common.py:99-175): K = A (std parameter) or 2A linear outputs.
struct GaussHeadsArgs {
    int B, H, A, K, mode, flags;       // mode 0 evaluate (actions given), 1 rsample with N(0,1) noise, 2 deterministic (mean)
    float min_std, max_std;
    const float* feats; int feats_ld;
    const float* w; const float* b;    // mu_maybe_std [K][H], [K]
    const float* std_param;            // [A] or null
    const float* w_critic; const float* b_critic;
    const float* actions_in; const int* rows;   // mode 0: arena (rows, A) float
    const float* noise;                // mode 1: [B][A]
    float* actions_out;                // modes 1/2: [B][A]
    float* value; float* logp; float* entropy;
    float* saved;                      // [B][16] per frame: mu[4], std[4], dmu/dz[4], dstd/draw[4] -- kept for backward (may be null)
};
struct GaussHeadsBwdArgs {
    int B, H, A, K;
    const float* d_value; const float* d_logp; const float* d_entropy;
    const float* actions; const int* rows;
    const float* saved;
    const float* w; const float* w_critic;
    float* dfeat;   // [B][H]
    float* dz;      // [B][8]: d mu_maybe_std outputs (cols < K, rest 0); with a std parameter cols A..2A-1 hold d std_param per frame
    float* dv_out;  // [B]
};
int gauss_heads_forward(const GaussHeadsArgs& a, hipStream_t stream);
int gauss_heads_backward(const GaussHeadsBwdArgs& a, hipStream_t stream);
int heads_forward(const HeadsArgs& a, hipStream_t stream);
int heads_backward(const HeadsBwdArgs& a, hipStream_t stream);
int gather_cols(const float* src, int src_ld, const int* rows, float* dst, int dst_ld, int col0, int ncols, int npad, int B,
                hipStream_t stream);
int masked_rows(const float* src, int src_stride, const uint8_t* masks, float* dst, int n, int H, hipStream_t stream);
int copy_rows(const float* src, const int* idx, int src_ld, float* dst, int dst_stride, int n, int H, hipStream_t stream);

}  // namespace hab
