// ops.h -- internal C++ interface between the kernel translation units and the policy engine.
#pragma once
#include "hab_common.h"

namespace hab {

struct ConvDesc {
    int B, H, W, C, Cout, KH, KW, stride, pad;
    int Creal = 0;  // weight-gradient only: number of real input channels when C is zero-padded (0 = C)
    int Ho() const { return (H + 2 * pad - KH) / stride + 1; }
    int Wo() const { return (W + 2 * pad - KW) / stride + 1; }
};
struct ObsView;

// gemm_ops.hip
int conv_fwd(const ConvDesc& d, const float* x, const float* wf, const float* bias, float* y, int relu, float* ws,
             size_t ws_floats, hipStream_t stream);
// wimg: bf16 weight image of wf for the patch-resident kernel (obs_conv_weight_image; obs_conv_weight_image_floats() floats), or null
int obs_conv_fwd(const ConvDesc& d, const ObsView& obs, const float* wf, const float* bias, float* y, int relu, float* ws,
                 size_t ws_floats, hipStream_t stream, const void* wimg = nullptr);
int64_t obs_conv_weight_image_floats();
int obs_conv_weight_image(const float* wf, int Cout, int KH, int KW, int C, void* img, hipStream_t stream);  // 1: no image for this filter
int conv_dgrad(const ConvDesc& d, const float* dy, const float* wd, const float* mask, const float* add, float* dx,
               float* ws, size_t ws_floats, hipStream_t stream);
int conv_wgrad(const ConvDesc& d, const float* x, const float* dy, float* dw_oihw, float* dbias, float* ws, size_t ws_floats,
               hipStream_t stream);
int obs_conv_wgrad(const ConvDesc& d, const ObsView& obs, const float* dy, float* dw_oihw, float* dbias, float* ws,
                   size_t ws_floats, hipStream_t stream);
int linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* y, int ldy, int M, int N, int K,
               int relu, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
int linear_dgrad(const float* dy, int lddy, const float* w, int ldw, const float* mask, int ldmask, int mask_cols, float* dx,
                 int lddx, int M, int Nin, int Kout, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
int linear_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int lddw, int Mrows, int Nout, int Kin,
                 int perm_c, int perm_hw, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
int colsum(const float* a, int lda, int M, int N, float* out, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
int repack_conv(const float* w_oihw, float* wf, float* wd, int Cout, int Cin, int KH, int KW, int cpad, hipStream_t stream);
int repack_flatten(const float* w, float* wp, int N, int C, int HW, hipStream_t stream);
int transpose2d(const float* w, float* wt, int R, int C, hipStream_t stream);
int pad_rows(const float* w, float* wp, int R, int C, int ld, hipStream_t stream);  // wp[r][0 .. ld) = w[r][0 .. C) then zeros

// rnn.hip -- packed-sequence recurrent encoder (GRU / LSTM)
struct PackInfo {          // device copies of build_pack_info_from_dones (rnn_state_encoder.py:35-168)
    const int* select_inds;        // [P]   packed position -> frame (t*n + j)
    const int* step_offsets;       // host  [max_len+1] prefix sums of num_seqs_at_step
    const int* num_seqs_at_step;   // host  [max_len]
    const int* frag_env;           // [F]   rnn_state_batch_inds: fragment -> env column j
    const int* frag_start;         // [F]   sequence_starts: fragment -> frame of its first step
    int P, F, max_len, n_envs;
};

struct RnnLayerParams {
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;  // reference layout [G*H][in], [G*H][H]
    const float* w_hh_t;                                                          // [H][G*H] transposed copy
    float* dw_ih; float* dw_hh; float* db_ih; float* db_hh;
    int in_dim;
    // inference step with the input projection inside the step kernel (rnn.hip): W_ih with rows of w_ih_ld floats (a multiple of 16,
    // zero beyond in_dim; 16-byte aligned rows) -- null: the projection is a separate contraction
    const float* w_ih_pad = nullptr;
    int w_ih_ld = 0;
};

struct RnnWork {  // per-layer activations saved by the forward for BPTT; all [P][...] in FRAME order
    float* gi;        // [P][G*H] input projection (+b_ih)
    float* gates;     // [P][G*H] post-nonlinearity gates (GRU: r,z,n ; LSTM: i,f,g,o)
    float* hn;        // GRU only: [P][H] W_hn h + b_hn
    float* hprev;     // [P][H] hidden state entering the step
    float* cprev;     // LSTM: [P][H]
    float* c;         // LSTM: [P][H] cell state after the step
    float* out;       // [P][H]
    float* dgi;       // [P][G*H] grads wrt input-side pre-activations
    float* dgh;       // [P][G*H] grads wrt hidden-side pre-activations (GRU; LSTM aliases dgi)
};

int rnn_frag_init(const float* h0, const int* env_rows, int env_stride, const uint8_t* masks, const int* mask_rows,
                  const int* frag_env, const int* frag_start, int F, int H, float* hinit, hipStream_t stream,
                  const int* env_first = nullptr);
int rnn_seq_layer_forward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx,
                          const float* hinit, const float* cinit, const PackInfo& pk, float* ws, size_t ws_floats,
                          hipStream_t stream);
// the L layers as a wavefront (one launch per packed step for all layers): 1 = form not applicable
int rnn_seq_wave_forward(int rnn_type, int H, int L, const RnnLayerParams* lp, const RnnWork* wk, const float* x, int ldx,
                         const float* const* hinit, const float* const* cinit, const PackInfo& pk, float* ws, size_t ws_floats,
                         hipStream_t stream);
int rnn_seq_wave_backward(int rnn_type, int H, int L, const RnnLayerParams* lp, const float* const* w_ih_t, const RnnWork* wk, const float* x0,
                          int ldx0, const float* dout, float* dx0, int lddx0, const float* dx_mask, int ldmask, int mask_cols,
                          const PackInfo& pk, float* scratch /* 2*L*F*H floats */, float* ws, size_t ws_floats, hipStream_t stream);
int rnn_step_layer_forward(int rnn_type, int H, const RnnLayerParams& lp, const float* x, int ldx, const float* h_in, int h_in_stride,
                           const float* c_in, int c_in_stride, const uint8_t* masks, int n, float* gi_scratch, float* h_out,
                           int h_out_stride, float* c_out, int c_out_stride, float* ws, size_t ws_floats, hipStream_t stream);
int rnn_seq_layer_backward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx,
                           const float* dout, float* dx, int lddx, const float* dx_mask, int ldmask, int mask_cols,
                           const PackInfo& pk, float* scratch /* 3*F*H floats */, float* ws, size_t ws_floats,
                           hipStream_t stream);
int matrix_path_bits();  // the current mask of hab_set_matrix_path (gemm_ops.hip)
// time-major form (regular T x n minibatch), chunkable in time: see rnn.hip
int rnn_tm_prepare(const uint8_t* masks, const int* rows, int B, uint8_t* frame_mask, int* iota, hipStream_t stream);
int rnn_tm_layer_forward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx, const float* hinit,
                         const float* cinit, const uint8_t* frame_mask, int n, int T_total, int t0, int t1, float* ws, size_t ws_floats,
                         hipStream_t stream);
int rnn_tm_layer_backward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* dout, float* dx, int lddx,
                          const float* dx_mask, int ldmask, int mask_cols, const uint8_t* frame_mask, const int* iota, int n, int T, int t0,
                          int t1, float* scratch, float* ws, size_t ws_floats, hipStream_t stream);
int rnn_tm_layer_param_grads(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx, int P, float* ws,
                             size_t ws_floats, hipStream_t stream);

}  // namespace hab
