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
    float* scratch = nullptr;           // optional: partial statistics of the chunk-parallel path for frames > 128 KB
    size_t scratch_floats = 0;          //   (>= B * ceil(HW*C / 16384) * groups * 2 floats), else the streaming kernel runs
};
struct GnBwdArgs {
    const float* x; const float* dy; const float* relu_out;  // relu_out: output of the fused ReLU (mask), or null
    float* dx;
    float* dy_masked;                   // optional: dy * (relu_out > 0) written out (gradient of the residual branch)
    const float* gamma; const float* mean; const float* rstd;
    float* chan_sums;                   // [B][2][C]: per-frame sum dy', sum dy'*xhat
    int B, HW, C, groups;
    float* scratch = nullptr;           // optional, >= B * ceil(HW*C / 8192) * 2 * C floats (chunk-parallel path)
    size_t scratch_floats = 0;
    // Stem form (groupnorm_relu_maxpool_forward): the layer's output went through ReLU + MaxPool2d(3, 2, 1) and was never stored.
    // dy is then GATHERED from the gradient of the pooled tensor (pool_dy [B][Ho][Wo][C], arg-max bytes pool_idx) and the ReLU mask is
    // recomputed from x, mean, rstd, gamma, beta with the forward's arithmetic; `dy` / `relu_out` are ignored.  Chunk-parallel path only.
    const float* pool_dy = nullptr; const uint8_t* pool_idx = nullptr; const float* beta = nullptr;
    int pH = 0, pW = 0;                 // spatial size of x (pooled: (pH + 1) / 2 x (pW + 1) / 2)
};
// 1-D sensor embeddings of PointNavResNetNet.forward (resnet_policy.py:662-753), each 32 wide, written side by side into
// the RNN input.  Slot kinds:
enum { EMB_POLAR = 0,   // pointgoal_with_gps_compass (rho, phi) -> Linear(3,32)([rho, cos(-phi), sin(-phi)])   :662-692
       EMB_TOKEN = 1,   // objectgoal id -> Embedding(n_categories, 32)                                          :715-717
       EMB_COSSIN = 2,  // compass x -> Linear(2,32)([cos x, sin x])                                             :719-729
       EMB_LIN2 = 3,    // gps (x, y) -> Linear(2,32)                                                            :731-734
       EMB_PREV = 4,    // previous action -> Embedding(A+1, 32)(mask ? a + 1 : 0)                               :747-753
       EMB_PREVLIN = 5,  // continuous previous action (A <= 4 floats) -> Linear(A, 32)(mask * a); ntok carries A      :754-757
       EMB_LINN = 6 };   // pointgoal (PointGoalSensor) / proximity: Linear(d, 32) on the raw d <= 4 floats; ntok carries d :694-700
constexpr int EMB_MAX_SLOTS = 7;
struct EmbedSlot {
    int kind;
    const void* in;      // sensor buffer (arena rows): f32 [rows][2] / int64 [rows][1] / f32 [rows][1] / f32 [rows][2] / int64 [rows][1]
    const float* w;      // Linear weight [32][nfeat] or embedding table [ntok][32]
    const float* b;      // Linear bias or null
    float* dw; float* db;  // gradients (backward only)
    int ntok;
};
__host__ __device__ inline int emb_nfeat(const EmbedSlot& sl) {
    return sl.kind == EMB_POLAR ? 3 : (sl.kind == EMB_COSSIN || sl.kind == EMB_LIN2) ? 2 : (sl.kind == EMB_PREVLIN || sl.kind == EMB_LINN) ? sl.ntok : 0;
}
struct EmbedArgs {
    EmbedSlot slot[EMB_MAX_SLOTS];
    int nslots;
    const uint8_t* masks; const int* rows;
    float* out; int ld, col0, B;
    float* saved;  // [B][nslots][4]: features / token (as float) -- kept for the backward pass
};
struct EmbedBwdArgs {
    EmbedSlot slot[EMB_MAX_SLOTS];
    int nslots;
    const float* saved;
    const float* dout; int ld, col0, B;
};

int ingest_pool(const uint8_t* rgb, const float* depth, const int32_t* semantic, const int* rows, float* y, int B, int H, int W, int cpad,
                int c_rgb, int c_depth, int c_sem, hipStream_t s, const float* norm_mean = nullptr, const float* norm_var = nullptr,
                const float* pivot = nullptr, double* mom_partial = nullptr, int* mom_blocks = nullptr);
// fused moments of the ingest (training-mode RunningMeanAndVar): mom_partial holds [<= INGEST_MOM_MAX_BLOCKS][16] doubles
constexpr int INGEST_MOM_MAX_BLOCKS = 2048;
int moment_finish_mean(const double* partial, int nblocks, int cpad, const float* pivot, int creal, long long npix, double* sums,
                       float* mean_out, float* count_out, float count_val, hipStream_t s);
int moment_finish_var(const double* sums, const float* pivot, int creal, int cpad, const float* mean_sum, float mean_div, long long npix,
                      float* var_out, hipStream_t s);
int chan_moment(const float* x, long long npix, int cpad, int mode, const float* mean, float* out, double* scratch, int scratch_len,
                hipStream_t s, float mean_div = 1.f, float* count_out = nullptr, float count_val = 0.f);
int rmv_update(float* r_mean, float* r_var, float* r_count, const float* b_mean, const float* b_var, float n, int C, hipStream_t s,
               const float* n_dev = nullptr, float div = 1.f, float* aff = nullptr);
int rmv_normalize(float* x, long long npix, int cpad, int C, const float* mean, const float* var, hipStream_t s);
// conv_gn_ops.hip -- convolution (bias-free) + GroupNorm (+ residual, + ReLU) in one launch for the small-batch passes (conv_gn_slab.h)
struct ConvGnArgs {
    const float* x;                 // [B][H][W][C]
    const unsigned short* w_planes; // forward-packed weight [Cout][KH*KW*C] as three fragment-ordered bf16 planes (weight_planes)
    const float* gamma; const float* beta; const float* residual;
    float* y;
    float* raw = nullptr; float* mean = nullptr; float* rstd = nullptr;  // kept for a backward pass when given
    int B, H, W, C, Cout, KH, KW, stride, pad, groups, relu;
    float eps;
};
int conv_gn_fused_ok(int C, int Cout, int H, int W, int KH, int KW, int stride, int pad, int groups);
int weight_planes(const float* w, int Cout, int K, unsigned short* planes, hipStream_t s);  // fragment-ordered bf16 planes of the exact split
int conv_gn_fused(const ConvGnArgs& a, hipStream_t s);  // 1: geometry not covered
// stem convolution 7x7 / 2 / 3, 4 -> 32 channels with the input strip resident in LDS (stem_conv_strip.h)
constexpr int STEM_PLANE_FLOATS = 3 * 14 * 1024 / 4;  // fragment-ordered bf16 planes of the filter, in floats of the packed arena
int stem_conv_ok(int H, int W, int C, int Cout, int KH, int KW, int stride, int pad);
int stem_weight_planes(const float* wf, unsigned short* planes, hipStream_t s);
int stem_conv_forward(const float* x, const unsigned short* planes, float* y, int B, int H, int W, hipStream_t s, const float* norm = nullptr,
                      float* gn_part = nullptr, int gn_groups = 0);  // gn_part: [B][ceil(Ho / STEM_STAT_ROWS)][groups][2] partial statistics
constexpr int STEM_STAT_ROWS = 8;  // 1: not covered
// its weight gradient, both operands resident in LDS, transpose reads (stem_wgrad_strip.h); ws >= 256 * 7 * 1024 floats; 1: not covered
int stem_conv_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int creal, float* ws, size_t ws_floats, hipStream_t s,
                    const float* norm = nullptr);
int stem_wgrad_ok(int H, int W, int C, int Cout, int KH, int KW, int stride, int pad);
int groupnorm_forward(const GnArgs& a, hipStream_t s);
int groupnorm_backward(const GnBwdArgs& a, hipStream_t s);
// GroupNorm + ReLU + MaxPool2d(3, 2, 1) in one pass over the GroupNorm input; the normalised frame is never written.  idx (nullable):
// arg-max bytes for the backward pass; a.mean / a.rstd (nullable) are written when given.  1: frame not on the chunk-parallel path.
int groupnorm_relu_maxpool_forward(const GnArgs& a, int H, int W, float* pool, uint8_t* idx, hipStream_t s, const float* ext_part = nullptr,
                                   int ext_rows = 0);
bool groupnorm_pool_fusable(int B, int HW, int C, int groups, size_t scratch_floats);  // forward AND backward chunk-parallel forms exist
int groupnorm_relu_materialize(const GnArgs& a, hipStream_t s);  // y = relu(GroupNorm(x)) from the SAVED a.mean / a.rstd (debug taps)
int maxpool_forward(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, hipStream_t s);
int maxpool_backward(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, hipStream_t s);
// Squeeze-and-excitation gate of SEBottleneck (resnet.py:92-113,155-187): pooled[b][c] = mean_hw x;  y = relu(gate[b][c] * x + residual);
// backward pieces: dm = dy * (y > 0) (written out: it is also the residual branch's gradient), dgate[b][c] = sum_hw dm * x, and
// dx = dm * gate[b][c] + dpool[b][c] / HW (the squeeze's gradient spread over the frame).
int se_pool(const float* x, float* pooled, int B, int HW, int C, hipStream_t s);
int se_apply_forward(const float* x, const float* gate, const float* residual, float* y, int B, int HW, int C, hipStream_t s);
int se_backward_reduce(const float* dy, const float* y, const float* x, float* dm, float* dgate, int B, int HW, int C, hipStream_t s);
int se_backward_apply(const float* dm, const float* gate, const float* dpool, float* dx, int B, int HW, int C, hipStream_t s);
int sigmoid_inplace(float* z, long long n, hipStream_t s);
int sigmoid_grad(const float* g, const float* dg, float* dz, long long n, hipStream_t s);
// Grouped 3x3 convolution of the ResNeXt bottlenecks (resnet.py:72-89, groups = cardinality): the weight (Cout, Cin/groups, KH, KW)
// is expanded to the dense block-diagonal kernel layouts (zeros off the diagonal) so that the dense contraction kernels serve it;
// the dense weight gradient is computed into scratch and its block diagonal is gathered back.
int repack_conv_grouped(const float* w, float* wf, float* wd, int Cout, int Cin, int groups, int KH, int KW, hipStream_t s);
int gather_grouped_wgrad(const float* dense_oihw, float* dw, int Cout, int Cin, int groups, int KH, int KW, hipStream_t s);
int embed_forward(const EmbedArgs& a, hipStream_t s);
int embed_backward(const EmbedBwdArgs& a, float* ws, size_t ws_floats, hipStream_t s);

}  // namespace hab
