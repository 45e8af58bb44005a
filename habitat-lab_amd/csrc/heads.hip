// heads.hip -- actor / critic heads (K12) and the small per-frame glue kernels of the policy.
// One wavefront per frame: the A+1 dot products over the 512 recurrent features are lane-partial
// sums folded with wave shuffles; log-softmax, entropy, sampling and log-prob gather are fused.
// (utils/common.py:64-96 CategoricalNet/CustomFixedCategorical, rl/ppo/policy.py:335-352,416-424)
#include "ops.h"
#include "heads.h"
#include "../../include/habitat_amd.h"

namespace hab {

constexpr int MAX_A = 8;

// mode 0: evaluate (actions given through act_rows gather)   mode 1: sample with exp noise   mode 2: deterministic mode()
__global__ void __launch_bounds__(256) heads_fwd_kernel(const HeadsArgs a) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= a.B) return;
    const float* x = a.feats + (size_t)f * (a.feats_ld ? a.feats_ld : a.H);
    float acc[MAX_A + 1];
#pragma unroll
    for (int k = 0; k <= MAX_A; ++k) acc[k] = 0.f;
    if ((a.H & 255) == 0) {
        // 16-byte loads, all of a 256-wide slice requested before any is used: one memory round trip per slice instead of one per
        // 64 elements (the kernel is a chain of latencies: 64 waves, 5 dot products of 512)
        for (int j = lane * 4; j < a.H; j += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + j);
            f32x4 wv[MAX_A + 1];
#pragma unroll
            for (int k = 0; k < MAX_A; ++k)
                wv[k] = k < a.A ? *reinterpret_cast<const f32x4*>(a.w_actor + (size_t)k * a.H + j) : f32x4{0.f, 0.f, 0.f, 0.f};
            wv[MAX_A] = *reinterpret_cast<const f32x4*>(a.w_critic + j);
#pragma unroll
            for (int k = 0; k <= MAX_A; ++k)
                acc[k] += (xv[0] * wv[k][0] + xv[1] * wv[k][1]) + (xv[2] * wv[k][2] + xv[3] * wv[k][3]);
        }
    } else {
        for (int j = lane; j < a.H; j += 64) {
            const float xv = x[j];
#pragma unroll
            for (int k = 0; k < MAX_A; ++k)
                if (k < a.A) acc[k] += xv * a.w_actor[(size_t)k * a.H + j];
            acc[MAX_A] += xv * a.w_critic[j];
        }
    }
#pragma unroll
    for (int k = 0; k <= MAX_A; ++k) acc[k] = wave_sum(acc[k]);
    // every lane now holds all sums; lane 0 finishes
    if (lane != 0) return;
    float z[MAX_A], mx = -INFINITY;
    for (int k = 0; k < a.A; ++k) { z[k] = acc[k] + a.b_actor[k]; mx = fmaxf(mx, z[k]); }
    float se = 0.f;
    for (int k = 0; k < a.A; ++k) se += expf(z[k] - mx);
    const float lse = mx + logf(se);
    float ln[MAX_A], p[MAX_A], mx2 = -INFINITY;
    for (int k = 0; k < a.A; ++k) { ln[k] = z[k] - lse; mx2 = fmaxf(mx2, ln[k]); }
    float s2 = 0.f;
    for (int k = 0; k < a.A; ++k) { p[k] = expf(ln[k] - mx2); s2 += p[k]; }
    float ent = 0.f;
    for (int k = 0; k < a.A; ++k) { p[k] = p[k] / s2; ent -= p[k] * ln[k]; }
    int act;
    if (a.mode == 0) {
        act = (int)a.actions_in[a.rows ? a.rows[f] : f];
    } else {
        float best = -INFINITY;
        act = 0;
        for (int k = 0; k < a.A; ++k) {
            const float s = (a.mode == 2) ? p[k] : __fdiv_rn(p[k], a.noise[(size_t)f * a.A + k]);
            if (s > best) { best = s; act = k; }
        }
        a.actions_out[f] = act;
    }
    a.value[f] = acc[MAX_A] + a.b_critic[0];
    a.logp[f] = ln[act];
    if (a.entropy) a.entropy[f] = ent;
    if (a.probs)
        for (int k = 0; k < a.A; ++k) { a.probs[(size_t)f * MAX_A + k] = p[k]; a.logits_n[(size_t)f * MAX_A + k] = ln[k]; }
}

int heads_forward(const HeadsArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.A <= 0 || a.A > MAX_A || !a.feats || !a.value || !a.logp) return HAB_ERR_ARG;
    heads_fwd_kernel<<<cdiv(a.B, 4), 256, 0, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Backward through heads for one frame per wave: dz, dv -> dfeat[f][:], and dzv[f][0..A] (A = value slot)
__global__ void __launch_bounds__(256) heads_bwd_kernel(const HeadsBwdArgs a) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= a.B) return;
    const float dlp = a.d_logp[f], den = a.d_entropy[f], dv = a.d_value[f];
    const int act = (int)a.actions[a.rows ? a.rows[f] : f];
    float dz[MAX_A];
    float ent = 0.f;
    for (int k = 0; k < a.A; ++k) ent -= a.probs[(size_t)f * MAX_A + k] * a.logits_n[(size_t)f * MAX_A + k];
    for (int k = 0; k < a.A; ++k) {
        const float p = a.probs[(size_t)f * MAX_A + k], ln = a.logits_n[(size_t)f * MAX_A + k];
        dz[k] = dlp * ((k == act ? 1.0f : 0.0f) - p) - den * p * (ln + ent);
    }
    if (lane == 0) {
        for (int k = 0; k < a.A; ++k) a.dzv[(size_t)f * MAX_A + k] = dz[k];
        for (int k = a.A; k < MAX_A; ++k) a.dzv[(size_t)f * MAX_A + k] = 0.f;
        a.dv_out[f] = dv;
    }
    for (int j = lane; j < a.H; j += 64) {
        float s = dv * a.w_critic[j];
        for (int k = 0; k < a.A; ++k) s += dz[k] * a.w_actor[(size_t)k * a.H + j];
        a.dfeat[(size_t)f * a.H + j] = s;
    }
}

int heads_backward(const HeadsBwdArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.A <= 0 || a.A > MAX_A) return HAB_ERR_ARG;
    heads_bwd_kernel<<<cdiv(a.B, 4), 256, 0, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Gaussian action head.  One wavefront per frame: K + 1 dot products over the recurrent features, then per action dimension
//   mu = tanh(z) | z;   raw = std_param | z[A + a];   raw = clamp(raw, min, max);   std = exp(raw) | raw;   std = softplus(std) | std
//   x = mu + std * eps (rsample) | mu | given;   log_prob = sum_a -(x - mu)^2 / (2 std^2) - log std - log sqrt(2 pi)
//   entropy = sum_a 0.5 + 0.5 log(2 pi) + log std            (torch.distributions.Normal, in its operation order)
// ------------------------------------------------------------------------------------------------------------------------
constexpr int GA = 4;  // max action dimensions of the Gaussian head

__device__ inline float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // F.softplus (beta 1, threshold 20)

__global__ void __launch_bounds__(256) gauss_heads_fwd_kernel(const GaussHeadsArgs a) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= a.B) return;
    const float* x = a.feats + (size_t)f * (a.feats_ld ? a.feats_ld : a.H);
    float acc[2 * GA + 1];
#pragma unroll
    for (int k = 0; k <= 2 * GA; ++k) acc[k] = 0.f;
    for (int j = lane; j < a.H; j += 64) {
        const float xv = x[j];
#pragma unroll
        for (int k = 0; k < 2 * GA; ++k)
            if (k < a.K) acc[k] += xv * a.w[(size_t)k * a.H + j];
        acc[2 * GA] += xv * a.w_critic[j];
    }
#pragma unroll
    for (int k = 0; k <= 2 * GA; ++k) acc[k] = wave_sum(acc[k]);
    if (lane != 0) return;
    const float LOG_SQRT_2PI = 0.91893853320467267f;  // math.log(math.sqrt(2 * math.pi))
    float logp = 0.f, ent = 0.f;
    for (int d = 0; d < a.A; ++d) {
        const float zmu = acc[d] + a.b[d];
        const float mu = (a.flags & HAB_GAUSS_TANH_MU) ? tanhf(zmu) : zmu;
        const float dmu = (a.flags & HAB_GAUSS_TANH_MU) ? 1.0f - mu * mu : 1.0f;
        const float raw = a.std_param ? a.std_param[d] : acc[a.A + d] + a.b[a.A + d];
        float sd = raw, dsd = 1.0f;
        if (a.flags & HAB_GAUSS_CLAMP_STD) {
            if (!(raw >= a.min_std && raw <= a.max_std)) dsd = 0.0f;  // torch.clamp passes the gradient inside [min, max]
            sd = fminf(fmaxf(raw, a.min_std), a.max_std);
        }
        if (a.flags & HAB_GAUSS_USE_LOG_STD) { sd = expf(sd); dsd *= sd; }
        if (a.flags & HAB_GAUSS_USE_SOFTPLUS) { dsd *= 1.0f / (1.0f + expf(-sd)); sd = softplusf_(sd); }
        float xv;
        if (a.mode == 0) xv = a.actions_in[(size_t)(a.rows ? a.rows[f] : f) * a.A + d];
        else {
            xv = a.mode == 1 ? mu + sd * a.noise[(size_t)f * a.A + d] : mu;
            a.actions_out[(size_t)f * a.A + d] = xv;
        }
        const float var = sd * sd, diff = xv - mu;
        logp += -(diff * diff) / (2.0f * var) - logf(sd) - LOG_SQRT_2PI;
        ent += 0.5f + LOG_SQRT_2PI + logf(sd);
        if (a.saved) {
            float* sv = a.saved + (size_t)f * 16;
            sv[d] = mu; sv[4 + d] = sd; sv[8 + d] = dmu; sv[12 + d] = dsd;
        }
    }
    a.value[f] = acc[2 * GA] + a.b_critic[0];
    a.logp[f] = logp;
    if (a.entropy) a.entropy[f] = ent;
}
int gauss_heads_forward(const GaussHeadsArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.A <= 0 || a.A > GA || (a.K != a.A && a.K != 2 * a.A) || !a.feats || !a.value || !a.logp || !a.w || !a.b) return HAB_ERR_ARG;
    if ((a.K == a.A) != (a.std_param != nullptr)) return HAB_ERR_ARG;
    if (a.mode == 0 ? !a.actions_in : (!a.actions_out || (a.mode == 1 && !a.noise))) return HAB_ERR_ARG;
    gauss_heads_fwd_kernel<<<cdiv(a.B, 4), 256, 0, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// dL/dmu = g_lp (x - mu) / std^2;  dL/dstd = g_lp ((x - mu)^2 / std^3 - 1 / std) + g_ent / std;  chained to the linear outputs
__global__ void __launch_bounds__(256) gauss_heads_bwd_kernel(const GaussHeadsBwdArgs a) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= a.B) return;
    const float dlp = a.d_logp[f], den = a.d_entropy[f], dv = a.d_value[f];
    const float* sv = a.saved + (size_t)f * 16;
    float dz[2 * GA];
#pragma unroll
    for (int k = 0; k < 2 * GA; ++k) dz[k] = 0.f;
    for (int d = 0; d < a.A; ++d) {
        const float mu = sv[d], sd = sv[4 + d];
        const float xv = a.actions[(size_t)(a.rows ? a.rows[f] : f) * a.A + d];
        const float diff = xv - mu, var = sd * sd;
        dz[d] = dlp * (diff / var) * sv[8 + d];
        dz[a.A + d] = (dlp * (diff * diff / (var * sd) - 1.0f / sd) + den / sd) * sv[12 + d];
    }
    if (lane == 0) {
        for (int k = 0; k < 8; ++k) a.dz[(size_t)f * 8 + k] = k < 2 * a.A ? dz[k] : 0.f;
        a.dv_out[f] = dv;
    }
    for (int j = lane; j < a.H; j += 64) {
        float s = dv * a.w_critic[j];
        for (int k = 0; k < a.K; ++k) s += dz[k] * a.w[(size_t)k * a.H + j];
        a.dfeat[(size_t)f * a.H + j] = s;
    }
}
int gauss_heads_backward(const GaussHeadsBwdArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.A <= 0 || a.A > GA || !a.saved || !a.actions || !a.dz || !a.dfeat) return HAB_ERR_ARG;
    gauss_heads_bwd_kernel<<<cdiv(a.B, 4), 256, 0, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// dst[f][col0 + c] = src[rows[f]][c]  (c < ncols), zero-fills pad columns [col0+ncols, col0+ncols+npad)
__global__ void gather_cols_kernel(const float* __restrict__ src, int src_ld, const int* __restrict__ rows, float* __restrict__ dst,
                                   int dst_ld, int col0, int ncols, int npad, int B) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = ncols + npad;
    if (e >= B * per) return;
    const int f = e / per, c = e % per;
    dst[(size_t)f * dst_ld + col0 + c] = (c < ncols) ? src[(size_t)(rows ? rows[f] : f) * src_ld + c] : 0.f;
}
int gather_cols(const float* src, int src_ld, const int* rows, float* dst, int dst_ld, int col0, int ncols, int npad, int B,
                hipStream_t stream) {
    if (!src || !dst || B <= 0 || ncols <= 0) return HAB_ERR_ARG;
    gather_cols_kernel<<<cdiv(B * (ncols + npad), 256), 256, 0, stream>>>(src, src_ld, rows, dst, dst_ld, col0, ncols, npad, B);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// hidden (n, L, H) slice copy helpers for the rollout step: dst[q][u] = mask[q] ? src[q*stride + u] : 0
__global__ void masked_rows_kernel(const float* __restrict__ src, int src_stride, const uint8_t* __restrict__ masks,
                                   float* __restrict__ dst, int n, int H) {
    const int q = blockIdx.x;
    const bool keep = masks[q] != 0;
    for (int u = threadIdx.x; u < H; u += blockDim.x) dst[(size_t)q * H + u] = keep ? src[(size_t)q * src_stride + u] : 0.f;
}
int masked_rows(const float* src, int src_stride, const uint8_t* masks, float* dst, int n, int H, hipStream_t stream) {
    if (!src || !masks || !dst || n <= 0) return HAB_ERR_ARG;
    masked_rows_kernel<<<n, 128, 0, stream>>>(src, src_stride, masks, dst, n, H);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// strided row copy: dst[q*dst_stride + u] = src[idx ? idx[q] : q][u]
__global__ void copy_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int src_ld, float* __restrict__ dst,
                                 int dst_stride, int n, int H) {
    const int q = blockIdx.x;
    const float* s = src + (size_t)(idx ? idx[q] : q) * src_ld;
    for (int u = threadIdx.x; u < H; u += blockDim.x) dst[(size_t)q * dst_stride + u] = s[u];
}
int copy_rows(const float* src, const int* idx, int src_ld, float* dst, int dst_stride, int n, int H, hipStream_t stream) {
    if (!src || !dst || n <= 0) return HAB_ERR_ARG;
    copy_rows_kernel<<<n, 128, 0, stream>>>(src, idx, src_ld, dst, dst_stride, n, H);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
