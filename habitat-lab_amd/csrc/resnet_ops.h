// resnet_ops.h -- argument blocks / launchers of the encoder-side HBM-bound kernels (resnet_ops.hip).
#pragma once
#include "hab_common.h"

namespace hab {

struct GnArgs {
    const float* x; float* y;           // [B][HW][C]
    const float* gamma; const float* beta;
    const float* residual;              // optional, added before the ReLU
    float* mean; float* rstd;           // [B][groups] saved for backward
    int B, HW, C, groups, relu;
    float eps;
};
struct GnBwdArgs {
    const float* x; const float* dy; const float* relu_out;  // relu_out: output of the fused ReLU (mask), or null
    float* dx;
    float* dy_masked;                   // optional: dy * (relu_out > 0) written out (gradient of the residual branch)
    const float* gamma; const float* mean; const float* rstd;
    float* chan_sums;                   // [B][2][C]: per-frame sum dy', sum dy'*xhat
    int B, HW, C, groups;
};
struct EmbedArgs {
    const float* goal; const int64_t* prev_actions; const uint8_t* masks; const int* rows;
    const float* w_t; const float* b_t; const float* emb;
    float* out; int ld, col0, B;
    float* saved;  // [B][4]: rho, cos(-phi), sin(-phi), token (as float) -- kept for the backward pass
};
struct EmbedBwdArgs {
    const float* saved;  // [B][4] written by the forward
    const float* dout; int ld, col0, B, num_tokens;
    float* dw_t; float* db_t; float* demb;
};

int ingest_pool(const uint8_t* rgb, const float* depth, const int* rows, float* y, int B, int H, int W, int cpad, int depth_first,
                hipStream_t s);
int chan_moment(const float* x, long long npix, int cpad, int mode, const float* mean, float* out, double* scratch, int scratch_len,
                hipStream_t s);
int rmv_update(float* r_mean, float* r_var, float* r_count, const float* b_mean, const float* b_var, float n, int C, hipStream_t s);
int rmv_normalize(float* x, long long npix, int cpad, int C, const float* mean, const float* var, hipStream_t s);
int groupnorm_forward(const GnArgs& a, hipStream_t s);
int groupnorm_backward(const GnBwdArgs& a, hipStream_t s);
int maxpool_forward(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, hipStream_t s);
int maxpool_backward(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, hipStream_t s);
int embed_forward(const EmbedArgs& a, hipStream_t s);
int embed_backward(const EmbedBwdArgs& a, float* ws, size_t ws_floats, hipStream_t s);

}  // namespace hab
