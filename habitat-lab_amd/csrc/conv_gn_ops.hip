// conv_gn_ops.hip -- launchers / C ABI of the fused convolution + GroupNorm kernel of the small-batch ResNet passes (conv_gn_slab.h).
#include "conv_gn_slab.h"
#include "conv1x1_gn_stream.h"
#include "stem_conv_strip.h"
#include "stem_wgrad_strip.h"
#include "resnet_ops.h"
#include "../../include/habitat_amd.h"

namespace hab {

int conv_gn_fused_ok(int C, int Cout, int H, int W, int KH, int KW, int stride, int pad, int groups) {
    return conv_gn_slab_covers(C, Cout, H, W, KH, KW, stride, pad, groups) ||
           conv1x1_gn_stream_covers(C, Cout, H, W, KH, KW, stride, pad, groups);
}

int weight_planes(const float* w, int Cout, int K, unsigned short* planes, hipStream_t s) { return cgs_split_weights(w, Cout, K, planes, s); }

int conv_gn_fused(const ConvGnArgs& q, hipStream_t s) {
    CgsArgs a{};
    a.x = q.x; a.wq = q.w_planes; a.gamma = q.gamma; a.beta = q.beta; a.residual = q.residual; a.y = q.y;
    a.raw = q.raw; a.mean = q.mean; a.rstd = q.rstd;
    a.B = q.B; a.H = q.H; a.W = q.W; a.C = q.C; a.Cout = q.Cout; a.KH = q.KH; a.KW = q.KW; a.stride = q.stride; a.pad = q.pad;
    a.groups = q.groups; a.relu = q.relu; a.eps = q.eps;
    const int rc = conv_gn_slab(a, s);
    if (rc != 1 || q.KH != 1 || q.KW != 1 || q.pad != 0) return rc;
    // 1x1 convolutions whose frames do not fit LDS (the bottleneck net's 32 x 32-pixel layers): activations streamed from memory
    C1gArgs b{};
    b.x = q.x; b.wq = q.w_planes; b.gamma = q.gamma; b.beta = q.beta; b.residual = q.residual; b.y = q.y;
    b.raw = q.raw; b.mean = q.mean; b.rstd = q.rstd;
    b.B = q.B; b.H = q.H; b.W = q.W; b.C = q.C; b.Cout = q.Cout; b.stride = q.stride; b.groups = q.groups; b.relu = q.relu; b.eps = q.eps;
    return conv1x1_gn_stream(b, s);
}

int stem_conv_ok(int H, int W, int C, int Cout, int KH, int KW, int stride, int pad) { return stem_conv_strip_covers(H, W, C, Cout, KH, KW, stride, pad); }
int stem_weight_planes(const float* wf, unsigned short* planes, hipStream_t s) { return stem_split_weights(wf, planes, s); }
int stem_conv_forward(const float* x, const unsigned short* planes, float* y, int B, int H, int W, hipStream_t s, const float* norm, float* gn_part,
                      int gn_groups) {
    static_assert(STEM_STAT_ROWS == STEM_TH, "statistics chunks are the kernel's strips");
    return stem_conv_strip(x, planes, y, B, H, W, s, norm, gn_part, gn_groups);
}
int stem_conv_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int creal, float* ws, size_t ws_floats, hipStream_t s,
                    const float* norm) {
    return stem_wgrad_strip(x, dy, dw_oihw, B, H, W, creal, ws, ws_floats, s, norm);
}
int stem_wgrad_ok(int H, int W, int C, int Cout, int KH, int KW, int stride, int pad) { return stem_wgrad_strip_covers(H, W, C, Cout, KH, KW, stride, pad); }

}  // namespace hab

using namespace hab;

extern "C" int hab_stem_split_weights(const float* w_fwd, uint16_t* planes, hipStream_t stream) { return stem_weight_planes(w_fwd, planes, stream); }
extern "C" int hab_stem_conv_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int creal, float* ws, size_t ws_floats,
                                   const float* norm, hipStream_t stream) {
    const int rc = stem_conv_wgrad(x, dy, dw_oihw, B, H, W, creal, ws, ws_floats, stream, norm);
    return rc == 1 ? HAB_ERR_UNSUPPORTED : rc;
}
extern "C" int hab_stem_conv_fwd(const float* x, const uint16_t* w_planes, float* y, int B, int H, int W, const float* norm, float* gn_part,
                                 int gn_groups, hipStream_t stream) {
    const int rc = stem_conv_forward(x, w_planes, y, B, H, W, stream, norm, gn_part, gn_groups);
    return rc == 1 ? HAB_ERR_UNSUPPORTED : rc;
}

extern "C" int hab_split_weight_planes(const float* w_fwd, int Cout, int K, uint16_t* planes, hipStream_t stream) {
    return weight_planes(w_fwd, Cout, K, planes, stream);
}

extern "C" int hab_conv_gn_fwd(const float* x, const uint16_t* w_planes, const float* gamma, const float* beta, const float* residual,
                               float* y, float* raw, float* mean, float* rstd, int B, int H, int W, int C, int Cout, int KH, int KW,
                               int stride, int pad, int groups, int relu, float eps, hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return HAB_ERR_ARG;
    ConvGnArgs q;
    q.x = x; q.w_planes = w_planes; q.gamma = gamma; q.beta = beta; q.residual = residual; q.y = y; q.raw = raw; q.mean = mean; q.rstd = rstd;
    q.B = B; q.H = H; q.W = W; q.C = C; q.Cout = Cout; q.KH = KH; q.KW = KW; q.stride = stride; q.pad = pad; q.groups = groups;
    q.relu = relu; q.eps = eps;
    const int rc = conv_gn_fused(q, stream);
    return rc == 1 ? HAB_ERR_UNSUPPORTED : rc;
}
