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

int heads_forward(const HeadsArgs& a, hipStream_t stream);
int heads_backward(const HeadsBwdArgs& a, hipStream_t stream);
int gather_cols(const float* src, int src_ld, const int* rows, float* dst, int dst_ld, int col0, int ncols, int npad, int B,
                hipStream_t stream);
int masked_rows(const float* src, int src_stride, const uint8_t* masks, float* dst, int n, int H, hipStream_t stream);
int copy_rows(const float* src, const int* idx, int src_ld, float* dst, int dst_stride, int n, int H, hipStream_t stream);

}  // namespace hab
