// gemm_ops.hip -- launchers that map conv / linear layers (fwd, dgrad, wgrad) onto igemm tiles,
// plus the small HBM-bound helpers around them (weight repack, column sums for bias gradients).
#include "igemm.h"
#include "igemm_dma.h"
#include "igemm_dma_wgrad.h"
#include "igemm_bf3.h"
#include "igemm_bf3_ws.h"
#include "dense_bf3.h"
#include "obs_conv_bf3.h"
#include "obs_conv_bf3_ws.h"
#include "obs_conv_patch.h"
#include "wgrad3x3_bf3.h"
#include "conv2_fwd_strip.h"
#include "conv2_dgrad_strip.h"
#include "obs_wgrad_bf3.h"
#include "conv_patch_bf3.h"
#include "wgrad3x3_patch.h"
#include "prob_build.h"
#include <stdlib.h>

#include <atomic>

#include "../../include/habitat_amd.h"

namespace hab {

// Tile choice by output width N (= Cout / out-features) and height M: tall-skinny tiles because the policy's
// layers have N in 32..512 and M in 1e2..1e7.  Weight-gradient problems have small M x N and a
// huge reduction: the 96-row tile (3 waves) avoids padding M = 288 / 576 / 1152 (3x3 taps x 32..128 channels).
static bool no_dma() { static const bool v = hab_env_flag("HAB_NO_DMA"); return v; }
static bool no_merged_dgrad() { static const bool v = hab_env_flag("HAB_NO_MERGED_DGRAD"); return v; }
static bool no_patch() { static const bool v = hab_env_flag("HAB_NO_PATCH"); return v; }
// split-bf16 matrix-pipe path (igemm_bf3.h) for the r-contiguous x r-contiguous contractions
// bit 0: r-contiguous x r-contiguous problems, bit 1: observation-ingest convolution (obs_conv_bf3.h), bit 2: problems with an
// i/j-contiguous operand (weight gradients, Linear data gradient), bit 3: prefer it over the fp32 patch / DMA weight-gradient kernels,
// bit 4: input-patch-resident stride-1 3x3 convolutions (conv_patch_bf3.h), bit 5: producer / consumer waves where they won
// (igemm_bf3_ws.h: long-K 128 x 128 forward-form tiles; obs_conv_bf3_ws.h: the observation-ingest convolution), bit 6: the
// observation-ingest convolution with the input patch resident in LDS (obs_conv_patch.h), bit 7: strip-resident 3x3 weight gradients
// (wgrad3x3_bf3.h), bit 8 / 9: SimpleCNN conv2 strip kernels, bit 10: the plain dense GEMM kernel for large Linear layers (dense_bf3.h), bit 11 (tests): that kernel for every shape it applies to,
// bit 12 (tests): the time-major recurrence as one launch per step instead of the persistent kernels (rnn_persist.h)
static std::atomic<int> g_bf3_mode{-1};  // engines of several inference-worker threads dispatch concurrently
static int bf3_mode() {
    int m = g_bf3_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        static const int from_env = hab_env_int("HAB_BF3", 2047);
        int expected = -1;
        g_bf3_mode.compare_exchange_strong(expected, from_env, std::memory_order_relaxed);
        m = g_bf3_mode.load(std::memory_order_relaxed);
    }
    return m;
}
int matrix_path_bits() { return bf3_mode(); }
extern "C" int hab_set_matrix_path(int mode) {
    const int prev = bf3_mode();
    if (mode >= 0) g_bf3_mode.store(mode, std::memory_order_relaxed);
    return prev;
}


// Producer / consumer waves (igemm_bf3_ws.h, matrix-path bit 5, on by default).  Measured on the MI355X against igemm_bf3_kernel, same
// arithmetic and bit-identical results (profiles/r03_ws_vs_base_layers.txt): the 128 x 128 forward-form tiles with a long reduction gain
// (3x3 256->256 at 4x4: 128 -> 163 TFLOP/s-eq forward and data gradient, fc 25088->512 forward 132 -> 150), every other shape loses
// (32- / 64-column tiles -10..-45 %, i/j-contiguous operands -30 %, K = 256 merged data gradient -18 %: the longer prologue and the
// halved MFMA wave count are not paid back).  It is therefore selected only where it won.
template <class P, int TM, int TN, int WM, int WN>
static int bf3_launch(const P& p, float* ws, size_t ws_floats, int target_blocks, hipStream_t stream) {
    constexpr bool WS_SHAPE = P::A_RC && P::B_RC && TM == 2 && TN == 2 && WM == 2 && WN == 2;
    if constexpr (WS_SHAPE) {
        constexpr int ws_min_k = 2048;
        if ((bf3_mode() & 32) && p.K >= ws_min_k) return igemm_bf3_ws_launch<P, TM, TN, WM, WN, 4>(p, ws, ws_floats, target_blocks, stream);
    }
    return igemm_bf3_launch<P, TM, TN, WM, WN>(p, ws, ws_floats, target_blocks, stream);
}

template <class P>
static int run_igemm(const P& p, float* ws, size_t ws_floats, hipStream_t stream, int target_blocks = 1024) {
    // Split-K target.  Forward-form problems (activations x weights) split only until there is one workgroup per CU: their
    // split-K second pass re-reads M*N*splits floats and, at rollout batch sizes, costs more than the idle CUs it fills
    // (measured: target 1024 -> 256 is +4 % on C2, +6 % on C3, all of it in the 64-frame rollout forward passes).  Weight
    // gradients (tiny M*N, K in the millions) keep the deeper split.
    if (P::A_RC || P::B_RC) target_blocks = 256;
    constexpr bool WG = !P::A_RC && !P::B_RC && AKv<P>::value == 4;  // weight-gradient form
    if constexpr (P::A_RC && P::B_RC) {
        if ((bf3_mode() & 1) && p.M > 64) {
            if (p.N <= 32) return bf3_launch<P, 2, 1, 4, 1>(p, ws, ws_floats, target_blocks, stream);
            if (p.N <= 64) {
                // 256 x 64 tiles halve the weight traffic per output row: +12 % on the 3x3 64 -> 64 convolutions of ResNet layer2
                // (0.089 -> 0.079 ms at 512 frames, forward and data gradient), nothing on SimpleCNN conv2 (K = 512: 0.308 vs 0.312 ms)
                if (p.M >= 256 * 512 && p.K >= 576) return bf3_launch<P, 2, 2, 4, 1>(p, ws, ws_floats, target_blocks, stream);
                return bf3_launch<P, 1, 2, 4, 1>(p, ws, ws_floats, target_blocks, stream);
            }
            return bf3_launch<P, 2, 2, 2, 2>(p, ws, ws_floats, target_blocks, stream);
        }
    } else if constexpr (AKv<P>::value == 4) {  // an i/j-contiguous operand: register-transposed staging (igemm_bf3.h)
        if ((bf3_mode() & 4) && p.M > 64) {
            if (p.N <= 32) {
                if constexpr (WG) {
                    if (p.M % 96 == 0 || cdiv(p.M, 96) * 96 < cdiv(p.M, 256) * 256)
                        return bf3_launch<P, 1, 1, 3, 1>(p, ws, ws_floats, target_blocks, stream);
                }
                return bf3_launch<P, 2, 1, 4, 1>(p, ws, ws_floats, target_blocks, stream);
            }
            if (p.N <= 64) {
                if constexpr (WG) {
                    if (p.M % 96 == 0 || cdiv(p.M, 96) * 96 < cdiv(p.M, 128) * 128)
                        return bf3_launch<P, 1, 2, 3, 1>(p, ws, ws_floats, target_blocks, stream);
                }
                return bf3_launch<P, 1, 2, 4, 1>(p, ws, ws_floats, target_blocks, stream);
            }
            return bf3_launch<P, 2, 2, 2, 2>(p, ws, ws_floats, target_blocks, stream);
        }
    }
    if constexpr (std::is_same_v<P, ConvFwdProb> || std::is_same_v<P, ConvDgradProb>) {
        if (p.dma_ok() && p.M > 64 && !no_dma()) {  // LDS-DMA staged variant (igemm_dma.h)
            if (p.N <= 32) return igemm_dma_launch<P, 2, 1, 4, 1, false>(p, ws, ws_floats, target_blocks, stream);
            if (p.N <= 64) return igemm_dma_launch<P, 1, 2, 4, 1, false>(p, ws, ws_floats, target_blocks, stream);
            if (p.N <= 128) return igemm_dma_launch<P, 2, 2, 2, 2, true>(p, ws, ws_floats, target_blocks, stream);
        }
    }
    if (p.N <= 32) {
        if (p.M <= 64) return igemm_launch<P, 1, 1, 2, 1>(p, ws, ws_floats, target_blocks, stream);
        if constexpr (WG) {
            if (p.M % 96 == 0 || cdiv(p.M, 96) * 96 < cdiv(p.M, 256) * 256)
                return igemm_launch<P, 1, 1, 3, 1>(p, ws, ws_floats, target_blocks, stream);
        }
        return igemm_launch<P, 2, 1, 4, 1>(p, ws, ws_floats, target_blocks, stream);
    } else if (p.N <= 64) {
        if (p.M <= 64) return igemm_launch<P, 1, 1, 2, 2>(p, ws, ws_floats, target_blocks, stream);
        if constexpr (WG) {
            if (p.M % 96 == 0 || cdiv(p.M, 96) * 96 < cdiv(p.M, 128) * 128)
                return igemm_launch<P, 1, 2, 3, 1>(p, ws, ws_floats, target_blocks, stream);
        }
        return igemm_launch<P, 1, 2, 4, 1>(p, ws, ws_floats, target_blocks, stream);
    } else {
        if (p.M <= 64) return igemm_launch<P, 1, 2, 2, 2>(p, ws, ws_floats, target_blocks, stream);
        return igemm_launch<P, 2, 2, 2, 2>(p, ws, ws_floats, target_blocks, stream);
    }
}

int conv_fwd(const ConvDesc& d, const float* x, const float* wf, const float* bias, float* y, int relu, float* ws,
             size_t ws_floats, hipStream_t stream) {
    ConvFwdProb p;
    HAB_TRY(build(p, d, x, wf, bias, y, relu));
    if ((bf3_mode() & 256) && (bf3_mode() & 1)) {  // SimpleCNN conv2: input strip in LDS, filter slices in registers (conv2_fwd_strip.h)
        const int rc = conv2_fwd_strip(p, ws, ws_floats, stream);
        if (rc != 1) return rc;
    }
    if ((bf3_mode() & 16) && d.stride == 1 && d.KH == 3 && d.KW == 3 && p.M > 64) {  // input patch resident in LDS (conv_patch_bf3.h)
        PatchGeom gq{};
        gq.in = x; gq.Hi = d.H; gq.Wi = d.W; gq.Ci = d.C; gq.Ho = p.g.Ho; gq.Wo = p.g.Wo; gq.KH = 3; gq.KW = 3;
        gq.off_h = -d.pad; gq.off_w = -d.pad; gq.flip = 0;
        const int rc = conv_patch_bf3_dispatch(p, gq, ws, ws_floats, stream);
        if (rc != 1) return rc;
    }
    return run_igemm(p, ws, ws_floats, stream);
}
int64_t obs_conv_weight_image_floats() { return ((int64_t)OCP_W_ELEMS + 1) / 2 + 4; }
int obs_conv_weight_image(const float* wf, int Cout, int KH, int KW, int C, void* img, hipStream_t stream) {
    return obs_conv_patch_weight_image(wf, Cout, KH, KW, C, reinterpret_cast<unsigned short*>(img), stream);
}
int obs_conv_fwd(const ConvDesc& d, const ObsView& obs, const float* wf, const float* bias, float* y, int relu, float* ws,
                 size_t ws_floats, hipStream_t stream, const void* wimg) {
    ObsConvFwdProb p;
    HAB_TRY(build(p, d, obs, wf, bias, y, relu));
    if ((bf3_mode() & 64) && (bf3_mode() & 2) && p.quad) {  // observation patch resident in LDS (obs_conv_patch.h): 8x8 / 4 RGB-D -> 32 only
        const int rc = obs_conv_patch_launch(p, ws, ws_floats, stream, reinterpret_cast<const unsigned short*>(wimg));
        if (rc != 1) return rc;
    }
    if ((bf3_mode() & 32) && (bf3_mode() & 2) && p.quad && p.M > 64) {  // producer / consumer waves (obs_conv_bf3_ws.h): 133 -> 148-152 TFLOP/s-eq at 1024 frames
        const int rc = obs_conv_bf3_ws_launch(p, ws, ws_floats, stream);
        if (rc != 1) return rc;
    }
    if ((bf3_mode() & 2) && p.quad && p.M > 64) {  // uint8 x split-bf16 weights on the matrix pipe (obs_conv_bf3.h)
        const int rc = obs_conv_bf3_launch<2>(p, ws, ws_floats, stream);
        if (rc != 1) return rc;
    }
    return run_igemm(p, ws, ws_floats, stream);
}
// dX pixels whose class has no taps (possible only when stride > kernel size) receive no contribution:
// they are written by the zero-tap path below so that dx is always fully defined.
__global__ void dgrad_empty_class_kernel(ConvDgradProb p) {
    const long long total = (long long)p.g.B * p.Hc * p.Wc * p.N;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
        p.store((int)(e / p.N), (int)(e % p.N), 0.f);
}

int conv_dgrad(const ConvDesc& d, const float* dy, const float* wd, const float* mask, const float* add, float* dx,
               float* ws, size_t ws_floats, hipStream_t stream) {
    if ((bf3_mode() & 512) && (bf3_mode() & 1)) {  // SimpleCNN conv2: dY strip in LDS, filter slices in registers (conv2_dgrad_strip.h)
        const int rc = conv2_dgrad_strip(d, dy, wd, mask, add, dx, stream);
        if (rc != 1) return rc;
    }
    if (d.stride > 1 && !no_dma() && !no_merged_dgrad()) {
        // kernel size a multiple of the stride: the stride classes share their dY gather -> one contraction with N = s*s*Cin
        ConvDgradMergedProb q;
        HAB_TRY(check_conv(d));
        q.g = make_geom(d);
        if (ConvDgradMergedProb::applicable(q.g)) {
            q.dy = dy; q.w = wd; q.mask = mask; q.add = add; q.dx = dx;
            q.finish();
            if ((bf3_mode() & 1) && q.M > 64) {  // split-bf16 matrix path, register-staged (igemm_bf3.h)
                if (q.N <= 32) return bf3_launch<ConvDgradMergedProb, 2, 1, 4, 1>(q, ws, ws_floats, 256, stream);
                if (q.N <= 64) return bf3_launch<ConvDgradMergedProb, 1, 2, 4, 1>(q, ws, ws_floats, 256, stream);
                return bf3_launch<ConvDgradMergedProb, 2, 2, 2, 2>(q, ws, ws_floats, 256, stream);
            }
            if (q.dma_ok()) {
                if (q.N <= 32) return igemm_dma_launch<ConvDgradMergedProb, 2, 1, 4, 1, false>(q, ws, ws_floats, 1024, stream);
                if (q.N <= 64) return igemm_dma_launch<ConvDgradMergedProb, 1, 2, 4, 1, false>(q, ws, ws_floats, 1024, stream);
                // single LDS buffer: K is short (KH/s * KW/s taps), 4 resident workgroups overlap each other's prologue / epilogue
                // (measured 79 vs 68 TFLOP/s double-buffered on SimpleCNN conv2 with the ReLU-mask epilogue)
                return igemm_dma_launch<ConvDgradMergedProb, 2, 2, 2, 2, false>(q, ws, ws_floats, 1024, stream);
            }
        }
    }
    for (int ph = 0; ph < d.stride; ++ph)
        for (int pw = 0; pw < d.stride; ++pw) {
            ConvDgradProb p;
            HAB_TRY(build(p, d, dy, wd, mask, add, dx, ph, pw));
            if (p.Hc <= 0 || p.Wc <= 0) continue;
            if ((bf3_mode() & 16) && d.stride == 1 && d.KH == 3 && d.KW == 3 && p.M > 64 && p.K > 0) {  // conv_patch_bf3.h, taps flipped
                PatchGeom gq{};
                gq.in = dy; gq.Hi = p.g.Ho; gq.Wi = p.g.Wo; gq.Ci = d.Cout; gq.Ho = d.H; gq.Wo = d.W; gq.KH = 3; gq.KW = 3;
                gq.off_h = d.pad - 2; gq.off_w = d.pad - 2; gq.flip = 1;
                const int rc = conv_patch_bf3_dispatch(p, gq, ws, ws_floats, stream);
                if (rc == HAB_OK) continue;
                if (rc != 1) return rc;
            }
            if (p.K <= 0) {
                p.M = d.B * p.Hc * p.Wc;
                dgrad_empty_class_kernel<<<1024, 256, 0, stream>>>(p);
                HAB_LAUNCH_CHECK();
                continue;
            }
            HAB_TRY(run_igemm(p, ws, ws_floats, stream));
        }
    return HAB_OK;
}
int conv_wgrad(const ConvDesc& d, const float* x, const float* dy, float* dw_oihw, float* dbias, float* ws, size_t ws_floats,
               hipStream_t stream) {
    ConvWgradProb p;
    HAB_TRY(build(p, d, x, dy, dw_oihw, dbias));
    if (ws && (bf3_mode() & 8) && (bf3_mode() & 128) && wgrad3x3_bf3_shape(p)) {  // strip-resident, transpose reads (wgrad3x3_bf3.h)
        const int rc = wgrad3x3_bf3(p, ws, ws_floats, stream);
        if (rc != 1) return rc;
    }
    if (ws && wgrad3x3_patch_ok(p) && !no_patch() && !(bf3_mode() & 8)) {  // 3x3/1/1 with W in {16, 32}: patch-resident kernel
        ConvWgradProb q = p;
        q.colsum = nullptr;
        if (dbias) HAB_TRY(colsum(dy, p.N, p.K, p.N, dbias, 0, ws, ws_floats, stream));
        return wgrad3x3_patch(q, ws, ws_floats, stream);
    }
    // LDS-DMA staged variant (igemm_dma_wgrad.h): measured faster only for unpadded convolutions with Cout <= 32 (SimpleCNN conv3:
    // 51 -> 64 TFLOP/s); with padding the per-pixel scalar decode + border tests cost more than the VGPR staging they replace.
    if (ws && d.pad == 0 && p.N <= 32 && !no_dma() && !(bf3_mode() & 8)) {
        ConvWgradProb q = p;
        q.colsum = nullptr;  // the DMA path never sees dY in registers: the bias gradient is a separate column sum
        if (wgrad_dma_ok(q)) {
            if (dbias) HAB_TRY(colsum(dy, p.N, p.K, p.N, dbias, 0, ws, ws_floats, stream));
            if ((q.M % 288 == 0) || (cdiv(q.M, 288) * 288 < cdiv(q.M, 256) * 256)) return igemm_dma_wgrad_launch<3, 3, 1>(q, ws, ws_floats, 1024, stream);
            return igemm_dma_wgrad_launch<2, 4, 1>(q, ws, ws_floats, 1024, stream);
        }
    }
    return run_igemm(p, ws, ws_floats, stream);
}
int obs_conv_wgrad(const ConvDesc& d, const ObsView& obs, const float* dy, float* dw_oihw, float* dbias, float* ws,
                   size_t ws_floats, hipStream_t stream) {
    ObsConvWgradProb p;
    HAB_TRY(build(p, d, obs, dy, dw_oihw, dbias));
    if ((bf3_mode() & 2) && p.quad && p.K > 4096) {  // uint8 x split-bf16 dY on the matrix pipe (obs_wgrad_bf3.h)
        const int rc = obs_wgrad_bf3_launch(p, ws, ws_floats, stream);
        if (rc != 1) return rc;
    }
    return run_igemm(p, ws, ws_floats, stream);
}
// Large dense layers on dense_bf3.h (matrix-path bit 10).  Returns 1 when the kernel does not apply.
static bool dense_on() { return (bf3_mode() & 1024) && (bf3_mode() & 1); }
static bool dense_worth(long long M, long long N, long long K) {
    if (bf3_mode() & 2048) return true;  // matrix-path bit 11 (tests): every applicable shape
    // the shapes it was built and measured for: >= ~10 GFLOP contractions (SimpleCNN's visual fc and its gradients); smaller layers keep
    // the igemm tiles (their launch is shorter than this kernel's 256 x 128 x K prologue / epilogue)
    static const long long min_flop = (long long)hab_env_int("HAB_DENSE_MIN_MFLOP", 4000) * 1000000LL;
    return 2 * M * N * K >= min_flop;
}
int linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* y, int ldy, int M, int N, int K,
               int relu, int accumulate, float* ws, size_t ws_floats, hipStream_t stream) {
    LinearFwdProb p;
    HAB_TRY(build(p, x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, accumulate));
    if (dense_on() && dense_worth(M, N, K)) {
        DenseArgs g{};
        g.M = M; g.N = N; g.K = K; g.a = x; g.lda = ldx; g.b = w; g.ldb = ldw; g.c = y; g.ldc = ldy; g.bias = bias; g.relu = relu;
        g.accumulate = accumulate; g.partial = ws;
        g.a_bytes = ((long long)(M - 1) * ldx + K) * 4; g.b_bytes = ((long long)(N - 1) * ldw + K) * 4;
        if (dense_bf3_ok(g, false, false)) {
            g.nsplit = ws ? dense_bf3_splits(g, ws_floats) : 1;
            HAB_TRY((dense_bf3_launch<0, 0>(g, stream)));
            if (g.nsplit > 1) {
                igemm_splitk_reduce<LinearFwdProb>(p, ws, g.nsplit, stream);
                HAB_LAUNCH_CHECK();
            }
            return HAB_OK;
        }
    }
    return run_igemm(p, ws, ws_floats, stream);
}
int linear_dgrad(const float* dy, int lddy, const float* w, int ldw, const float* mask, int ldmask, int mask_cols, float* dx,
                 int lddx, int M, int Nin, int Kout, int accumulate, float* ws, size_t ws_floats, hipStream_t stream) {
    LinearDgradProb p;
    HAB_TRY(build(p, dy, lddy, w, ldw, mask, ldmask, mask_cols, dx, lddx, M, Nin, Kout, accumulate));
    if (dense_on() && !mask && dense_worth(M, Nin, Kout)) {  // dX[M][Nin] = dY[M][Kout] W[Kout][Nin]: W is the k-strided operand
        DenseArgs g{};
        g.M = M; g.N = Nin; g.K = Kout; g.a = dy; g.lda = lddy; g.b = w; g.ldb = ldw; g.c = dx; g.ldc = lddx; g.accumulate = accumulate;
        g.partial = ws;
        g.a_bytes = ((long long)(M - 1) * lddy + Kout) * 4; g.b_bytes = ((long long)(Kout - 1) * ldw + Nin) * 4;
        if (dense_bf3_ok(g, false, true)) {
            g.nsplit = ws ? dense_bf3_splits(g, ws_floats) : 1;
            HAB_TRY((dense_bf3_launch<0, 1>(g, stream)));
            if (g.nsplit > 1) {
                igemm_splitk_reduce<LinearDgradProb>(p, ws, g.nsplit, stream);
                HAB_LAUNCH_CHECK();
            }
            return HAB_OK;
        }
    }
    return run_igemm(p, ws, ws_floats, stream);
}
int linear_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int lddw, int Mrows, int Nout, int Kin,
                 int perm_c, int perm_hw, int accumulate, float* ws, size_t ws_floats, hipStream_t stream) {
    LinearWgradProb p;
    HAB_TRY(build(p, dy, lddy, x, ldx, dw, lddw, Mrows, Nout, Kin, perm_c, perm_hw, accumulate));
    if (dense_on() && dense_worth(Nout, Kin, Mrows)) {  // dW[Nout][Kin] = dY^T X: both operands are indexed by the frame = k-strided
        DenseArgs g{};
        g.M = Nout; g.N = Kin; g.K = Mrows; g.a = dy; g.lda = lddy; g.b = x; g.ldb = ldx; g.c = dw; g.ldc = lddw; g.accumulate = accumulate;
        g.perm_c = perm_c; g.perm_hw = perm_hw; g.partial = ws;
        g.a_bytes = ((long long)(Mrows - 1) * lddy + Nout) * 4; g.b_bytes = ((long long)(Mrows - 1) * ldx + Kin) * 4;
        if (dense_bf3_ok(g, true, true)) {
            g.nsplit = ws ? dense_bf3_splits(g, ws_floats) : 1;
            HAB_TRY((dense_bf3_launch<1, 1>(g, stream)));
            if (g.nsplit > 1) {  // (the reduction pass applies the flatten permutation and `accumulate`: LinearWgradProb::store)
                igemm_splitk_reduce<LinearWgradProb>(p, ws, g.nsplit, stream);
                HAB_LAUNCH_CHECK();
            }
            return HAB_OK;
        }
    }
    return run_igemm(p, ws, ws_floats, stream);
}

// ---------------------------------------------------------------------------------------------
// Column sums (bias gradients): out[n] = sum_m a[m][n] * (mask ? mask[m][n] > 0 : 1).
// Stage 1: each block reduces a row range into partial[block][N]; stage 2: fixed-order sum.
// ---------------------------------------------------------------------------------------------
// grid = (column blocks of cw = min(64, N) columns, row chunks); 256 threads = (256 / cw) row lanes x cw columns.
__global__ void __launch_bounds__(256) colsum_stage1(const float* __restrict__ a, int lda, int M, int N, int rows_per_block,
                                                     float* __restrict__ partial) {
    const int cw = N < 64 ? N : 64;
    const int rl = 256 / cw;
    const int c = threadIdx.x % cw, r = threadIdx.x / cw;
    const int n = blockIdx.x * cw + c;
    __shared__ float sm[256];
    const int m_begin = blockIdx.y * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float s0 = 0.f, s1 = 0.f;
    if (r < rl && n < N) {
        int m = m_begin + r;
        for (; m + rl < m_end; m += 2 * rl) {
            s0 += a[(size_t)m * lda + n];
            s1 += a[(size_t)(m + rl) * lda + n];
        }
        if (m < m_end) s0 += a[(size_t)m * lda + n];
    }
    sm[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (r == 0 && n < N) {
        float t = 0.f;
        for (int q = 0; q < rl; ++q) t += sm[q * cw + c];
        partial[(size_t)blockIdx.y * N + n] = t;
    }
}
__global__ void __launch_bounds__(256) colsum_stage2(const float* __restrict__ partial, int nblocks, int N, float* __restrict__ out,
                                                     int accumulate) {
    // 64 columns per block, 4 row lanes
    __shared__ float sm[4][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    // four independent chains per row lane: up to 64 partials per lane were one dependent chain of L2 round trips (15 us for a 32-column
    // sum that moves 32 KB; 80 such launches per ResNet18 minibatch, 200 per ResNet50 one)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int b = r;
        for (; b + 12 < nblocks; b += 16) {
            s0 += partial[(size_t)b * N + n];
            s1 += partial[(size_t)(b + 4) * N + n];
            s2 += partial[(size_t)(b + 8) * N + n];
            s3 += partial[(size_t)(b + 12) * N + n];
        }
        for (; b < nblocks; b += 4) s0 += partial[(size_t)b * N + n];
    }
    const float s = (s0 + s1) + (s2 + s3);
    sm[r][c] = s;
    __syncthreads();
    if (r == 0 && n < N) {
        const float t = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
        out[n] = accumulate ? out[n] + t : t;
    }
}

int colsum(const float* a, int lda, int M, int N, float* out, int accumulate, float* ws, size_t ws_floats, hipStream_t stream) {
    if (M <= 0 || N <= 0 || !a || !out || !ws) return HAB_ERR_ARG;
    const int cw = N < 64 ? N : 64;
    const int colblocks = cdiv(N, cw);
    // enough row chunks for ~1024 workgroups, at least 16 rows each
    int chunks = cdiv(1024, colblocks);
    if (chunks > cdiv(M, 16)) chunks = cdiv(M, 16);
    while (chunks > 1 && (size_t)chunks * N > ws_floats) chunks >>= 1;
    if ((size_t)chunks * N > ws_floats) return HAB_ERR_ARG;
    const int rpb = cdiv(M, chunks);
    chunks = cdiv(M, rpb);
    colsum_stage1<<<dim3(colblocks, chunks), 256, 0, stream>>>(a, lda, M, N, rpb, ws);
    HAB_LAUNCH_CHECK();
    colsum_stage2<<<cdiv(N, 64), 256, 0, stream>>>(ws, chunks, N, out, accumulate);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight repack: OIHW master -> Wf[co][(kh,kw,ci)] and Wd[ci][(kh,kw,co)].  cpad >= Cin pads the
// packed input-channel dimension with zeros (stem conv: 4/5 obs channels -> 4/8).
// ---------------------------------------------------------------------------------------------
__global__ void repack_conv_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int Cout,
                                   int Cin, int KH, int KW, int cpad) {
    const int total = Cout * cpad * KH * KW;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        // e enumerates Wf: [co][kh][kw][ci(cpad)]
        const int ci = e % cpad;
        int t = e / cpad;
        const int kw = t % KW; t /= KW;
        const int kh = t % KH;
        const int co = t / KH;
        const float v = (ci < Cin) ? w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
        if (wf) wf[e] = v;
        if (wd && ci < Cin) wd[(((size_t)ci * KH + kh) * KW + kw) * Cout + co] = v;
    }
}

int repack_conv(const float* w_oihw, float* wf, float* wd, int Cout, int Cin, int KH, int KW, int cpad, hipStream_t stream) {
    if (!w_oihw || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || cpad < Cin) return HAB_ERR_ARG;
    if (wd && cpad != Cin) return HAB_ERR_UNSUPPORTED;
    const int total = Cout * cpad * KH * KW;
    repack_conv_kernel<<<min(1024, cdiv(total, 256)), 256, 0, stream>>>(w_oihw, wf, wd, Cout, Cin, KH, KW, cpad);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Linear weight after nn.Flatten of an NCHW tensor: Wp[n][hw*C + c] = W[n][c*HW + hw].
__global__ void repack_flatten_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int C, int HW) {
    const size_t total = (size_t)N * C * HW;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t t = e / C;
        const int hw = (int)(t % HW);
        const size_t n = t / HW;
        wp[e] = w[(n * C + c) * HW + hw];
    }
}
int repack_flatten(const float* w, float* wp, int N, int C, int HW, hipStream_t stream) {
    if (!w || !wp || N <= 0 || C <= 0 || HW <= 0) return HAB_ERR_ARG;
    const size_t total = (size_t)N * C * HW;
    repack_flatten_kernel<<<(int)fmin(4096.0, (double)cdivl((long long)total, 256)), 256, 0, stream>>>(w, wp, N, C, HW);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Transposed copy: wt[c][r] = w[r][c]   (RNN recurrent weights for the BPTT mat-vec)
__global__ void transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int C) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < C) ? w[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = bx + j, r = by + threadIdx.x;
        if (c < C && r < R) wt[(size_t)c * R + r] = tile[threadIdx.x][j];
    }
}
__global__ void pad_rows_kernel(const float* __restrict__ w, float* __restrict__ wp, int R, int C, int ld) {
    const long long total = (long long)R * ld;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % ld);
        const long long r = e / ld;
        wp[e] = c < C ? w[r * C + c] : 0.f;
    }
}
int pad_rows(const float* w, float* wp, int R, int C, int ld, hipStream_t stream) {
    if (!w || !wp || R <= 0 || C <= 0 || ld < C) return HAB_ERR_ARG;
    const long long total = (long long)R * ld;
    pad_rows_kernel<<<(int)fmin(2048.0, (double)cdivl(total, 256)), 256, 0, stream>>>(w, wp, R, C, ld);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
int transpose2d(const float* w, float* wt, int R, int C, hipStream_t stream) {
    if (!w || !wt || R <= 0 || C <= 0) return HAB_ERR_ARG;
    transpose_kernel<<<dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, stream>>>(w, wt, R, C);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab

// ------------------------------------------- C ABI -------------------------------------------
using namespace hab;

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

extern "C" int hab_conv2d_fwd(const float* x, const float* w_fwd, const float* bias, float* y, int B, int H, int W, int C,
                              int Cout, int KH, int KW, int stride, int pad, int relu, float* ws, size_t ws_floats,
                              hipStream_t stream) {
    if (!x || !w_fwd || !y) return HAB_ERR_ARG;
    return conv_fwd(mk(B, H, W, C, Cout, KH, KW, stride, pad), x, w_fwd, bias, y, relu, ws, ws_floats, stream);
}
extern "C" int hab_obs_conv2d_fwd(const uint8_t* rgb, const float* depth, const int* rows, const float* w_fwd,
                                  const float* bias, float* y, int B, int H, int W, int Cout, int KH, int KW, int stride,
                                  int pad, int relu, float* ws, size_t ws_floats, hipStream_t stream) {
    if ((!rgb && !depth) || !w_fwd || !y) return HAB_ERR_ARG;
    ObsView o = mkobs(rgb, depth, rows, H, W);
    return obs_conv_fwd(mk(B, H, W, o.C, Cout, KH, KW, stride, pad), o, w_fwd, bias, y, relu, ws, ws_floats, stream);
}
extern "C" int hab_conv2d_dgrad(const float* dy, const float* w_dgrad, const float* relu_mask, const float* add, float* dx,
                                int B, int H, int W, int C, int Cout, int KH, int KW, int stride, int pad, float* ws,
                                size_t ws_floats, hipStream_t stream) {
    if (!dy || !w_dgrad || !dx) return HAB_ERR_ARG;
    return conv_dgrad(mk(B, H, W, C, Cout, KH, KW, stride, pad), dy, w_dgrad, relu_mask, add, dx, ws, ws_floats, stream);
}
extern "C" int hab_conv2d_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, int B, int H, int W, int C, int Cout,
                                int KH, int KW, int stride, int pad, float* ws, size_t ws_floats, hipStream_t stream) {
    if (!x || !dy || !dw_oihw) return HAB_ERR_ARG;
    return conv_wgrad(mk(B, H, W, C, Cout, KH, KW, stride, pad), x, dy, dw_oihw, dbias, ws, ws_floats, stream);
}
extern "C" int hab_obs_conv2d_wgrad(const uint8_t* rgb, const float* depth, const int* rows, const float* dy, float* dw_oihw,
                                    float* dbias, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, float* ws,
                                    size_t ws_floats, hipStream_t stream) {
    if ((!rgb && !depth) || !dy || !dw_oihw) return HAB_ERR_ARG;
    ObsView o = mkobs(rgb, depth, rows, H, W);
    return obs_conv_wgrad(mk(B, H, W, o.C, Cout, KH, KW, stride, pad), o, dy, dw_oihw, dbias, ws, ws_floats, stream);
}
extern "C" int hab_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* y, int ldy, int M,
                              int N, int K, int relu, int accumulate, float* ws, size_t ws_floats, hipStream_t stream) {
    return linear_fwd(x, ldx, w, ldw, bias, y, ldy, M, N, K, relu, accumulate, ws, ws_floats, stream);
}
extern "C" int hab_linear_dgrad(const float* dy, int lddy, const float* w, int ldw, const float* relu_mask, int ldmask,
                                float* dx, int lddx, int M, int n_in, int n_out, int accumulate, float* ws, size_t ws_floats,
                                hipStream_t stream) {
    return linear_dgrad(dy, lddy, w, ldw, relu_mask, ldmask, n_in, dx, lddx, M, n_in, n_out, accumulate, ws, ws_floats, stream);
}
extern "C" int hab_linear_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int lddw, int M, int n_out,
                                int n_in, int perm_c, int perm_hw, int accumulate, float* ws, size_t ws_floats,
                                hipStream_t stream) {
    return linear_wgrad(dy, lddy, x, ldx, dw, lddw, M, n_out, n_in, perm_c, perm_hw, accumulate, ws, ws_floats, stream);
}
extern "C" int hab_debug_dense_trace(long long* out320) {  // development: see dn_trace (dense_bf3.h)
    return (int)hipMemcpyFromSymbol(out320, HIP_SYMBOL(hab::dn_trace), sizeof(long long) * 8 * 8 * 5);
}
extern "C" int hab_colsum(const float* a, int lda, int M, int N, float* out, int accumulate, float* ws, size_t ws_floats,
                          hipStream_t stream) {
    return colsum(a, lda, M, N, out, accumulate, ws, ws_floats, stream);
}
extern "C" int hab_repack_conv_weight(const float* w_oihw, float* w_fwd, float* w_dgrad, int Cout, int Cin, int KH, int KW,
                                      int cin_padded, hipStream_t stream) {
    return repack_conv(w_oihw, w_fwd, w_dgrad, Cout, Cin, KH, KW, cin_padded, stream);
}
extern "C" int hab_repack_flatten_weight(const float* w, float* w_packed, int N, int C, int HW, hipStream_t stream) {
    return repack_flatten(w, w_packed, N, C, HW, stream);
}
extern "C" int hab_transpose2d(const float* w, float* wt, int R, int C, hipStream_t stream) {
    return transpose2d(w, wt, R, C, stream);
}
