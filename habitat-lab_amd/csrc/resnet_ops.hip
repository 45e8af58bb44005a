// resnet_ops.hip -- the HBM-bound kernels around the conv contractions of the GroupNorm-ResNet encoder
// (K1-K3, K5-K8 of SURVEY.md): observation ingest (uint8 scale + 2x2 average pool + channel concat, read in
// place through rows[]), RunningMeanAndVar statistics and normalisation, GroupNorm forward / backward
// (optionally fused with the residual add and ReLU), 3x3/2 max-pool forward / backward, and the tiny goal /
// previous-action embeddings.  All NHWC fp32, coalesced along C; reductions use wave shuffles + LDS.
#include "ops.h"
#include "resnet_ops.h"
#include "../../include/habitat_amd.h"

namespace hab {

// ------------------------------------------------------------------------------------------------------
// Ingest (resnet_policy.py:259-271): per visual key permute -> uint8 * fp32(1/255) -> cat -> avg_pool2d(2).
// Output y[f][h/2][w/2][cpad]: channels rgb(3), depth(1), zero padding up to cpad.
// ------------------------------------------------------------------------------------------------------
// MOM (training mode, RunningMeanAndVar): the same pass also accumulates, per channel, S1 = sum (x - p) and S2 = sum (x - p)^2 in double
// about a pivot p (the running mean: any value works), one partial pair per workgroup in a fixed order -- the batch mean is p + S1 / n and
// the variance about ANY later mean m (the all-reduced one under DD-PPO) is (S2 - 2 (m - p) S1 + n (m - p)^2) / n, so the two extra read
// passes over the pooled tensor (chan_moment modes 0 and 1) are not needed.  mom_partial: [gridDim.x][2][8] doubles.
template <bool MOM>
__global__ void __launch_bounds__(256) ingest_pool_kernel(const uint8_t* __restrict__ rgb, const float* __restrict__ depth,
                                                          const int32_t* __restrict__ semantic, const int* __restrict__ rows,
                                                          float* __restrict__ y, int B, int H, int W, int cpad, int c_rgb, int c_depth,
                                                          int c_sem, const float* __restrict__ nmean, const float* __restrict__ nvar, int creal,
                                                          const float* __restrict__ pivot, double* __restrict__ mom_partial) {
#pragma clang fp contract(off)  // the reference rounds the uint8 scaling and every addition of the 2x2 average separately
    // optional RunningMeanAndVar normalisation of the result (evaluation mode: a fixed affine per channel; rmv_normalize_kernel's
    // arithmetic: fma(x, inv_std, -mean * inv_std))
    __shared__ float na[8], nb[8];
    if (nmean) {
        if (threadIdx.x < 8) {
            const int c = threadIdx.x;
            float a_ = 0.f, b_ = 0.f;
            if (c < creal) { b_ = rsqrtf(fmaxf(nvar[c], 1e-2f)); a_ = -nmean[c] * b_; }
            na[c] = a_; nb[c] = b_;
        }
        __syncthreads();
    }
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo;
    const float inv255 = (float)(1.0 / 255.0);
    // moments: fp32 accumulators over runs of 32 pixels of this thread (about the pivot the first moment is a short random walk and the
    // second a sum of 32 positive terms: relative error of a run ~1e-7, unbiased), flushed into doubles
    double s1[MOM ? 8 : 1], s2[MOM ? 8 : 1];
    float f1[MOM ? 8 : 1], f2[MOM ? 8 : 1], pv[MOM ? 8 : 1];
    int run = 0;
    if constexpr (MOM) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { s1[c] = 0.0; s2[c] = 0.0; f1[c] = 0.f; f2[c] = 0.f; pv[c] = c < creal ? pivot[c] : 0.f; }
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int wo = (int)(e % Wo);
        const long long t = e / Wo;
        const int ho = (int)(t % Ho);
        const int f = (int)(t / Ho);
        const size_t srow = rows ? rows[f] : f;
        float out[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) out[c] = 0.f;
        if (rgb) {
            float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
                    const uint8_t* p = rgb + ((srow * H + 2 * ho + dh) * W + 2 * wo + dw) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) { const float m = (float)p[c] * inv255; s[c] = s[c] + m; }
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c_rgb + c] = s[c] * 0.25f;
        }
        if (depth) {
            float s = 0.f;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) s = s + depth[(srow * H + 2 * ho + dh) * W + 2 * wo + dw];
            out[c_depth] = s * 0.25f;
        }
        if (semantic) {  // int32 ids are concatenated by type promotion (torch.cat) and averaged like any channel
            float s = 0.f;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) s = s + (float)semantic[(srow * H + 2 * ho + dh) * W + 2 * wo + dw];
            out[c_sem] = s * 0.25f;
        }
        if (nmean) {
#pragma unroll
            for (int c = 0; c < 8; ++c) out[c] = __builtin_fmaf(out[c], nb[c], na[c]);
        }
        if constexpr (MOM) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < cpad) { const float d = out[c] - pv[c]; f1[c] = f1[c] + d; f2[c] = __builtin_fmaf(d, d, f2[c]); }
            if (++run == 32) {
                run = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < cpad) { s1[c] += (double)f1[c]; s2[c] += (double)f2[c]; f1[c] = 0.f; f2[c] = 0.f; }
            }
        }
        float* o = y + (size_t)e * cpad;
        for (int c = 0; c < cpad; c += 4) *reinterpret_cast<f32x4*>(o + c) = *reinterpret_cast<const f32x4*>(out + c);
    }
    if constexpr (MOM) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < cpad) { s1[c] += (double)f1[c]; s2[c] += (double)f2[c]; }
        // fixed-order reduction: xor tree inside the wave, then the four waves in order
        __shared__ double red[4][16];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < cpad) {
#pragma unroll
                for (int off = 32; off; off >>= 1) { s1[c] += __shfl_xor(s1[c], off); s2[c] += __shfl_xor(s2[c], off); }
                if (lane == 0) { red[wave][c] = s1[c]; red[wave][8 + c] = s2[c]; }
            }
        __syncthreads();
        if (threadIdx.x < 16) {
            const int c = threadIdx.x & 7;
            double tsum = 0.0;
            if (c < cpad) tsum = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
            mom_partial[(size_t)blockIdx.x * 16 + threadIdx.x] = tsum;
        }
    }
}

int ingest_pool(const uint8_t* rgb, const float* depth, const int32_t* semantic, const int* rows, float* y, int B, int H, int W, int cpad,
                int c_rgb, int c_depth, int c_sem, hipStream_t s, const float* norm_mean, const float* norm_var, const float* pivot,
                double* mom_partial, int* mom_blocks) {
    if ((!rgb && !depth && !semantic) || !y || B <= 0 || H < 2 || W < 2 || (cpad != 4 && cpad != 8)) return HAB_ERR_ARG;
    if ((norm_mean == nullptr) != (norm_var == nullptr)) return HAB_ERR_ARG;
    if (mom_partial && (!pivot || !mom_blocks || norm_mean)) return HAB_ERR_ARG;  // moments are those of the un-normalised tensor
    const int n = (rgb ? 3 : 0) + (depth ? 1 : 0) + (semantic ? 1 : 0);
    if (n > cpad || (rgb && (c_rgb < 0 || c_rgb + 3 > n)) || (depth && (c_depth < 0 || c_depth >= n)) || (semantic && (c_sem < 0 || c_sem >= n)))
        return HAB_ERR_ARG;
    const long long total = (long long)B * (H / 2) * (W / 2);
    if (mom_partial) {
        const int blocks = (int)fmin((double)INGEST_MOM_MAX_BLOCKS, (double)cdivl(total, 256));
        *mom_blocks = blocks;
        ingest_pool_kernel<true><<<blocks, 256, 0, s>>>(rgb, depth, semantic, rows, y, B, H, W, cpad, c_rgb, c_depth, c_sem, nullptr, nullptr,
                                                        n, pivot, mom_partial);
    } else {
        ingest_pool_kernel<false><<<(int)fmin(8192.0, (double)cdivl(total, 256)), 256, 0, s>>>(rgb, depth, semantic, rows, y, B, H, W, cpad,
                                                                                                c_rgb, c_depth, c_sem, norm_mean, norm_var, n,
                                                                                                nullptr, nullptr);
    }
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Finish of the fused moments (ingest_pool_kernel<true>): one workgroup; thread (q, v) sums the partials b = q, q + 16, ... of value v
// (v < 8: S1 of channel v, else S2 of channel v - 8), the 16 q's are then added in order.  sums[0..15] keeps S1 / S2 for the variance;
// mean_out[c] = p + S1 / n.
__global__ void __launch_bounds__(256) moment_finish_mean_kernel(const double* __restrict__ partial, int nblocks, int cpad,
                                                                 const float* __restrict__ pivot, int creal, double inv_n,
                                                                 double* __restrict__ sums, float* __restrict__ mean_out,
                                                                 float* __restrict__ count_out, float count_val) {
    __shared__ double sm[16][16];
    const int v = threadIdx.x & 15, q = threadIdx.x >> 4;
    double t = 0.0;
    for (int b = q; b < nblocks; b += 16) t += partial[(size_t)b * 16 + v];
    sm[q][v] = t;
    __syncthreads();
    if (threadIdx.x < 16) {
        double tot = 0.0;
        for (int k = 0; k < 16; ++k) tot += sm[k][v];
        sums[v] = tot;
        if (v < 8 && v < cpad) mean_out[v] = v < creal ? (float)((double)pivot[v] + tot * inv_n) : 0.f;
    }
    if (threadIdx.x == 0 && count_out) *count_out = count_val;
}
// var_out[c] = biased variance of this rank's pixels about m = mean_sum[c] / mean_div (the cross-rank mean under DD-PPO)
__global__ void moment_finish_var_kernel(const double* __restrict__ sums, const float* __restrict__ pivot, int creal, int cpad,
                                         const float* __restrict__ mean_sum, float mean_div, double n, float* __restrict__ var_out) {
    const int c = threadIdx.x;
    if (c >= cpad) return;
    float out = 0.f;
    if (c < creal) {
        const float m = mean_sum[c] / mean_div;  // `new_mean /= world_size` (exact IEEE division)
        const double d = (double)m - (double)pivot[c];
        out = (float)((sums[8 + c] - 2.0 * d * sums[c] + n * d * d) / n);
    }
    var_out[c] = out;
}
int moment_finish_mean(const double* partial, int nblocks, int cpad, const float* pivot, int creal, long long npix, double* sums,
                       float* mean_out, float* count_out, float count_val, hipStream_t s) {
    if (!partial || nblocks < 1 || !pivot || !sums || !mean_out || npix <= 0 || cpad > 8 || creal > cpad) return HAB_ERR_ARG;
    moment_finish_mean_kernel<<<1, 256, 0, s>>>(partial, nblocks, cpad, pivot, creal, 1.0 / (double)npix, sums, mean_out, count_out, count_val);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
int moment_finish_var(const double* sums, const float* pivot, int creal, int cpad, const float* mean_sum, float mean_div, long long npix,
                      float* var_out, hipStream_t s) {
    if (!sums || !pivot || !mean_sum || !var_out || npix <= 0 || cpad > 8 || creal > cpad || !(mean_div >= 1.f)) return HAB_ERR_ARG;
    moment_finish_var_kernel<<<1, 64, 0, s>>>(sums, pivot, creal, cpad, mean_sum, mean_div, (double)npix, var_out);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// RunningMeanAndVar (running_mean_and_var.py:24-78).  Channel moments over (B, H, W) of an NHWC tensor
// with cpad channels: stage 1 -> per-block partial sums (double), stage 2 -> fixed-order reduce.
// mode 0: sum_c x            -> out[c] = mean_c
// mode 1: sum_c (x - m_c)^2  -> out[c] = biased variance about the supplied mean (m = stats_in[c])
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) chan_moment_stage1(const float* __restrict__ x, long long npix, int cpad, int mode,
                                                          const float* __restrict__ mean, float mean_div, double* __restrict__ partial) {
    __shared__ double sm[256];
    const int c = threadIdx.x % cpad, lane = threadIdx.x / cpad, nl = 256 / cpad;
    const float m = (mode == 1) ? mean[c] / mean_div : 0.f;  // `new_mean /= world_size` of the summed means (exact IEEE division)
    double s = 0.0;
    for (long long p = (long long)blockIdx.x * nl + lane; p < npix; p += (long long)gridDim.x * nl) {
        const float v = x[p * cpad + c];
        if (mode == 0) s += (double)v;
        else { const float d = v - m; s += (double)(d * d); }
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0) {
        double t = 0.0;
        for (int q = 0; q < nl; ++q) t += sm[q * cpad + c];
        partial[(size_t)blockIdx.x * cpad + c] = t;
    }
}
__global__ void chan_moment_stage2(const double* __restrict__ partial, int nblocks, int cpad, double inv_n, float* __restrict__ out,
                                   float* __restrict__ count_out, float count_val) {
    const int c = threadIdx.x;
    if (c == 0 && count_out) *count_out = count_val;  // new_count = n frames of THIS rank; summed over ranks with the means
    if (c >= cpad) return;
    double t = 0.0;
    for (int b = 0; b < nblocks; ++b) t += partial[(size_t)b * cpad + c];
    out[c] = (float)(t * inv_n);
}
int chan_moment(const float* x, long long npix, int cpad, int mode, const float* mean, float* out, double* scratch, int scratch_len,
                hipStream_t s, float mean_div, float* count_out, float count_val) {
    if (!x || !out || !scratch || npix <= 0 || (256 % cpad) || (mode == 1 && !mean) || !(mean_div >= 1.f)) return HAB_ERR_ARG;
    int blocks = (int)fmin(1024.0, (double)cdivl(npix, 256 / cpad * 8));
    if (blocks * cpad > scratch_len) blocks = scratch_len / cpad;
    if (blocks < 1) return HAB_ERR_ARG;
    chan_moment_stage1<<<blocks, 256, 0, s>>>(x, npix, cpad, mode, mean, mean_div, scratch);
    HAB_LAUNCH_CHECK();
    chan_moment_stage2<<<1, 64, 0, s>>>(scratch, blocks, cpad, 1.0 / (double)npix, out, count_out, count_val);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Chan merge of (running mean, var, count) with the batch moments, all on device (one thread per channel).
// new_count = batch size n (number of frames, as in the reference -- not the number of pixels).
// DD-PPO (running_mean_and_var.py:38-49): b_mean / b_var hold the SUMS over ranks (divided here by `div` = world size) and the
// batch count is the all-reduced number of frames *n_dev -- ranks may hold different numbers of frames (preempted rollouts).
__global__ void rmv_update_kernel(float* __restrict__ r_mean, float* __restrict__ r_var, float* __restrict__ r_count,
                                  const float* __restrict__ b_mean, const float* __restrict__ b_var, float n_host,
                                  const float* __restrict__ n_dev, float div, int C, float* __restrict__ aff) {
    const int c = threadIdx.x;
    const float count = r_count[0];
    const float n = n_dev ? n_dev[0] : n_host;
    if (c < C) {
        const float mean = r_mean[c], var = r_var[c], nm = b_mean[c] / div, nv = b_var[c] / div;
        const float m_a = var * count, m_b = nv * n;
        const float d = nm - mean;
        const float M2 = m_a + m_b + d * d * count * n / (count + n);
        const float v_new = M2 / (count + n), m_new = (count * mean + n * nm) / (count + n);
        r_var[c] = v_new;
        r_mean[c] = m_new;
        if (aff) {  // the normalisation as the consumers apply it: fma(x, aff[c], aff[8 + c]) (rmv_normalize_kernel's arithmetic)
            const float inv = rsqrtf(fmaxf(v_new, 1e-2f));
            aff[c] = inv;
            aff[8 + c] = -m_new * inv;
        }
    } else if (aff && c < 8) { aff[c] = 0.f; aff[8 + c] = 0.f; }
    __syncthreads();
    if (c == 0) r_count[0] = count + n;
}
int rmv_update(float* r_mean, float* r_var, float* r_count, const float* b_mean, const float* b_var, float n, int C, hipStream_t s,
               const float* n_dev, float div, float* aff) {
    if (!r_mean || !r_var || !r_count || !b_mean || !b_var || C <= 0 || C > 64 || !(div >= 1.f) || (aff && C > 8)) return HAB_ERR_ARG;
    rmv_update_kernel<<<1, 64, 0, s>>>(r_mean, r_var, r_count, b_mean, b_var, n, n_dev, div, C, aff);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// x = addcmul(-mean*inv_std, x, inv_std), inv_std = rsqrt(max(var, 1e-2)), in place; padded channels stay 0.
__global__ void __launch_bounds__(256) rmv_normalize_kernel(float* __restrict__ x, long long n_elems, int cpad, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ var) {
    __shared__ float a[8], b[8];
    if (threadIdx.x < cpad) {
        const int c = threadIdx.x;
        if (c < C) {
            const float inv = rsqrtf(fmaxf(var[c], 1e-2f));
            b[c] = inv;
            a[c] = -mean[c] * inv;
        } else { a[c] = 0.f; b[c] = 0.f; }
    }
    __syncthreads();
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n_elems; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % cpad);
        x[e] = __builtin_fmaf(x[e], b[c], a[c]);
    }
}
int rmv_normalize(float* x, long long npix, int cpad, int C, const float* mean, const float* var, hipStream_t s) {
    if (!x || !mean || !var || npix <= 0 || cpad > 8 || C > cpad) return HAB_ERR_ARG;
    const long long n = npix * cpad;
    rmv_normalize_kernel<<<(int)fmin(8192.0, (double)cdivl(n, 256)), 256, 0, s>>>(x, n, cpad, C, mean, var);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// GroupNorm (resnet.py:51-57,207-219; eps 1e-5) on NHWC.  One workgroup (256 threads) per frame; the frame's
// tensor (8 KB .. 512 KB) is read from HBM once and re-read from L2.
// forward : two-pass statistics (mean, then squared deviations about it); y = (x-mu)*rstd*gamma + beta
//           [+ residual] [ReLU].  mu / rstd saved for backward.
// backward: dy' = dy * (relu_out > 0); per-channel sums S1 = sum dy', S2 = sum dy'*xhat are written per frame
//           (reduced over frames afterwards -> dbeta, dgamma); dx = rstd*(g - mean_g(g) - xhat*mean_g(g*xhat)),
//           g = dy'*gamma.  Optionally dy' itself is written out (gradient of the residual branch).
// Thread mapping: C4 = C/4 float4 columns.  C4 <= 256: thread t owns column t % C4 and pixel lane t / C4 of
// np = 256 / C4; C4 > 256: thread t owns columns t, t+256, ... and all pixels.  Reductions are fixed-order.
// ------------------------------------------------------------------------------------------------------
struct GnMap {
    int C4, np, ncc;  // float4 columns, pixel lanes, column chunks
    __device__ GnMap(int C) { C4 = C >> 2; np = C4 >= 256 ? 1 : 256 / C4; ncc = C4 >= 256 ? C4 / 256 : 1; }
    __device__ int col(int t, int cc) const { return C4 >= 256 ? cc * 256 + t : t % C4; }
    __device__ int lane(int t) const { return C4 >= 256 ? 0 : t / C4; }
};

// per-channel reduction over the pixel lanes: red[t*4+k] (this chunk) -> ch[c]
__device__ inline void gn_fold(const GnMap& m, int cc, const float* red, float* ch, int t) {
    const int cols = m.C4 >= 256 ? 256 : m.C4;
    for (int c = t; c < cols * 4; c += 256) {
        const int c4 = c >> 2, k = c & 3;
        float u = 0.f;
        for (int q = 0; q < m.np; ++q) u += red[(q * cols + c4) * 4 + k];
        ch[(m.C4 >= 256 ? cc * 1024 : 0) + c] = u;
    }
}

__global__ void __launch_bounds__(256) groupnorm_fwd_kernel(const GnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, HW = a.HW, G = a.groups, cpg = C / G;
    const GnMap m(C);
    const int f = blockIdx.x, t = threadIdx.x;
    const int pl = m.lane(t);
    const bool act = pl < m.np;
    float* red = sm;            // [256][4]
    float* ch = sm + 1024;      // [C]
    float* mu_s = ch + C;       // [G]
    float* rs_s = mu_s + G;     // [G]
    const float* x = a.x + (size_t)f * HW * C;
    const float inv_m = 1.0f / (float)(HW * cpg);
    // pass 1: channel sums -> group means
    for (int cc = 0; cc < m.ncc; ++cc) {
        const int col = m.col(t, cc);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (act)
            for (int p = pl; p < HW; p += m.np) s += *reinterpret_cast<const f32x4*>(x + (size_t)p * C + col * 4);
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + t * 4) = act ? s : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        gn_fold(m, cc, red, ch, t);
    }
    __syncthreads();
    if (t < G) {
        float u = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) u += ch[c];
        mu_s[t] = u * inv_m;
    }
    __syncthreads();
    // pass 2: squared deviations -> rstd
    for (int cc = 0; cc < m.ncc; ++cc) {
        const int col = m.col(t, cc);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            f32x4 mu;
#pragma unroll
            for (int k = 0; k < 4; ++k) mu[k] = mu_s[(col * 4 + k) / cpg];
            for (int p = pl; p < HW; p += m.np) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(x + (size_t)p * C + col * 4) - mu;
                s += d * d;
            }
        }
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + t * 4) = act ? s : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        gn_fold(m, cc, red, ch, t);
    }
    __syncthreads();
    if (t < G) {
        float u = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) u += ch[c];
        const float rs = rsqrtf(u * inv_m + a.eps);
        rs_s[t] = rs;
        a.mean[(size_t)f * G + t] = mu_s[t];
        a.rstd[(size_t)f * G + t] = rs;
    }
    __syncthreads();
    if (!act) return;
    float* y = a.y + (size_t)f * HW * C;
    const float* res = a.residual ? a.residual + (size_t)f * HW * C : nullptr;
    for (int cc = 0; cc < m.ncc; ++cc) {
        const int col = m.col(t, cc);
        f32x4 sc, sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = col * 4 + k, g = c / cpg;
            sc[k] = rs_s[g] * a.gamma[c];
            sh[k] = a.beta[c] - mu_s[g] * sc[k];
        }
        for (int p = pl; p < HW; p += m.np) {
            const size_t o = (size_t)p * C + col * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(x + o) * sc + sh;
            if (res) v += *reinterpret_cast<const f32x4*>(res + o);
            if (a.relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
            }
            *reinterpret_cast<f32x4*>(y + o) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Register-resident GroupNorm: frames of <= 128 KB (every ResNet18 layer behind the stem's max-pool).  The frame is loaded
// ONCE into VGPRs (NV float4 per thread, all loads in flight together), the exact two-pass statistics and the apply run from
// registers: 1 read + 1 write of HBM per activation instead of 3 + 1 (forward) / 2x3 + 1 (backward), and one memory round
// trip instead of three dependent ones when only 64 frames are in flight (rollout).  NT threads per frame; thread t owns the
// float4s t, t + NT, ... (coalesced); C/4 is a power of two dividing NT, so the thread's channel column is fixed.
// Per-channel reductions: xor-shuffle over the lanes of a wave that share the column, then a fixed-order LDS fold.
// ------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void gn_col_reduce(f32x4 s, int C4, float* red, float* ch, int C, int t) {
    const int lane = t & 63, wave = t >> 6;
    int nq;
    __syncthreads();  // red / ch consumers of the previous round are done
    if (C4 < 64) {
        for (int off = C4; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += __shfl_xor(s[k], off, 64);
        }
        if (lane < C4) *reinterpret_cast<f32x4*>(red + (wave * C4 + lane) * 4) = s;
        nq = NT / 64;
    } else {
        *reinterpret_cast<f32x4*>(red + t * 4) = s;
        nq = NT / C4;
    }
    __syncthreads();
    for (int c = t; c < C; c += NT) {
        const int c4 = c >> 2, k = c & 3;
        float u = 0.f;
        for (int q = 0; q < nq; ++q) u += red[(q * C4 + c4) * 4 + k];
        ch[c] = u;
    }
    __syncthreads();
}

template <int NT, int NV>
__global__ void __launch_bounds__(NT) groupnorm_fwd_reg_kernel(const GnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, G = a.groups, cpg = C / G, C4 = C >> 2;
    const int F4 = a.HW * C4;
    const int f = blockIdx.x, t = threadIdx.x, col = t & (C4 - 1);
    float* red = sm;             // [NT][4]
    float* ch = sm + NT * 4;     // [C]
    float* mu_s = ch + C;        // [G]
    float* rs_s = mu_s + G;      // [G]
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (size_t)f * a.HW * C);
    const float inv_m = 1.0f / (float)(a.HW * cpg);
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = t + j * NT;
        v[j] = i < F4 ? x4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) s += v[j];
    gn_col_reduce<NT>(s, C4, red, ch, C, t);
    if (t < G) {
        float u = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) u += ch[c];
        mu_s[t] = u * inv_m;
    }
    __syncthreads();
    f32x4 mu;
#pragma unroll
    for (int k = 0; k < 4; ++k) mu[k] = mu_s[(col * 4 + k) / cpg];
    s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const f32x4 d = v[j] - mu;
        if (t + j * NT < F4) s += d * d;
    }
    gn_col_reduce<NT>(s, C4, red, ch, C, t);
    if (t < G) {
        float u = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) u += ch[c];
        const float rs = rsqrtf(u * inv_m + a.eps);
        rs_s[t] = rs;
        a.mean[(size_t)f * G + t] = mu_s[t];
        a.rstd[(size_t)f * G + t] = rs;
    }
    __syncthreads();
    f32x4 sc, sh;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = col * 4 + k, g = c / cpg;
        sc[k] = rs_s[g] * a.gamma[c];
        sh[k] = a.beta[c] - mu_s[g] * sc[k];
    }
    f32x4* y4 = reinterpret_cast<f32x4*>(a.y + (size_t)f * a.HW * C);
    const f32x4* r4 = a.residual ? reinterpret_cast<const f32x4*>(a.residual + (size_t)f * a.HW * C) : nullptr;
    if (r4) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = t + j * NT;
            if (i < F4) v[j] = v[j] * sc + sh + r4[i];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = v[j] * sc + sh;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = t + j * NT;
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[j][k] = v[j][k] > 0.f ? v[j][k] : 0.f;
        }
        if (i < F4) y4[i] = v[j];
    }
}

template <int NT, int NV>
__global__ void __launch_bounds__(NT) groupnorm_bwd_reg_kernel(const GnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, G = a.groups, cpg = C / G, C4 = C >> 2;
    const int F4 = a.HW * C4;
    const int f = blockIdx.x, t = threadIdx.x, col = t & (C4 - 1);
    float* red = sm;            // [NT][4]
    float* c1 = sm + NT * 4;    // [C]  per-channel S1 = sum dy'
    float* c2 = c1 + C;         // [C]  per-channel S2 = sum dy' * xhat
    float* g1 = c2 + C;         // [G]
    float* g2 = g1 + G;
    const size_t base = (size_t)f * a.HW * C;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + base);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(a.dy + base);
    const f32x4* ro4 = a.relu_out ? reinterpret_cast<const f32x4*>(a.relu_out + base) : nullptr;
    f32x4* dym4 = a.dy_masked ? reinterpret_cast<f32x4*>(a.dy_masked + base) : nullptr;
    const float* mean = a.mean + (size_t)f * G;
    const float* rstd = a.rstd + (size_t)f * G;
    f32x4 xh[NV], dv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = t + j * NT;
        const bool ok = i < F4;
        xh[j] = ok ? x4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        dv[j] = ok ? dy4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (ro4) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = t + j * NT;
            if (i < F4) {
                const f32x4 rv = ro4[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[j][k] = rv[k] > 0.f ? dv[j][k] : 0.f;
            }
        }
    }
    f32x4 mu, rs, ga;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int c = col * 4 + k, g = c / cpg; mu[k] = mean[g]; rs[k] = rstd[g]; ga[k] = a.gamma[c]; }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = t + j * NT;
        if (dym4 && i < F4) dym4[i] = dv[j];
        xh[j] = (xh[j] - mu) * rs;
        if (i >= F4) xh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        s1 += dv[j];
        s2 += dv[j] * xh[j];
    }
    gn_col_reduce<NT>(s1, C4, red, c1, C, t);
    gn_col_reduce<NT>(s2, C4, red, c2, C, t);
    for (int c = t; c < C; c += NT) {
        a.chan_sums[((size_t)f * 2 + 0) * C + c] = c1[c];   // -> dbeta after the reduction over frames
        a.chan_sums[((size_t)f * 2 + 1) * C + c] = c2[c];   // -> dgamma
    }
    if (t < G) {
        float u = 0.f, w = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) { u += a.gamma[c] * c1[c]; w += a.gamma[c] * c2[c]; }
        const float inv_m = 1.0f / (float)(a.HW * cpg);
        g1[t] = u * inv_m; g2[t] = w * inv_m;
    }
    __syncthreads();
    f32x4 m1, m2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int g = (col * 4 + k) / cpg; m1[k] = g1[g]; m2[k] = g2[g]; }
    f32x4* dx4 = reinterpret_cast<f32x4*>(a.dx + base);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = t + j * NT;
        if (i < F4) dx4[i] = rs * (dv[j] * ga - m1 - xh[j] * m2);
    }
}

// ------------------------------------------------------------------------------------------------------
// Chunk-parallel GroupNorm for frames that do not fit one workgroup's registers (> 128 KB: the 64x64x32 stem output, ResNet50's
// 32x32x128 and 16x16x256 bottleneck outputs).  A frame is cut into chunks of NT*NV float4 (whole pixels); kernel 1 keeps its
// chunk in registers and writes per-(chunk, group) partial statistics -- count is implied, (mean_k, M2_k) from an exact local
// two-pass -- kernel 2 merges them with Chan's formula  M2 = sum M2_k + sum n_k (mean_k - mean)^2  (as accurate as a global
// two-pass, fixed order) and applies.  HBM traffic 2 reads + 1 write instead of 3 + 1, and chunks x frames workgroups instead of
// `frames` (a 32-frame rollout batch used 32 of 256 CUs).  Backward: kernel 1 = per-channel partial sums, kernel 2 = dx.
// ------------------------------------------------------------------------------------------------------
constexpr int GNC_NT = 256, GNC_NV_F = 16, GNC_NV_B = 8;
// backward: a workgroup walks GNC_REPS_B consecutive sub-chunks of GNC_NT * GNC_NV_B float4 (round 6).  Both backward kernels are
// memory-latency-bound (SQ_WAIT_ANY 0.64 / 0.77 of wave cycles, profiles/r06_c3_sq_counters.txt) and every workgroup starts with a serial
// prologue -- merging the frame's partial sums (dx kernel), two LDS column reductions (sums kernel); with 2048 float4 per workgroup the
// stem's backward launched 65 536 workgroups per pass.
constexpr int GNC_REPS_B = 4;
static int gnc_reps_b() { static const int v = std::max(1, hab_env_int("HAB_GN_REPS", GNC_REPS_B)); return v; }

template <int NT, int NV>
__global__ void __launch_bounds__(NT) gn_chunk_stats_kernel(const GnArgs a, int nchunks, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, G = a.groups, cpg = C / G, C4 = C >> 2;
    const int F4 = a.HW * C4;
    const int f = blockIdx.x / nchunks, ck = blockIdx.x % nchunks, t = threadIdx.x, col = t & (C4 - 1);
    const int base = ck * NT * NV;
    float* red = sm;
    float* ch = sm + NT * 4;
    float* mu_s = ch + C;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (size_t)f * a.HW * C);
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = base + t + j * NT;
        v[j] = i < F4 ? x4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int len = min(NT * NV, F4 - base);             // float4s of this chunk (a multiple of C4: whole pixels)
    const float inv_n = 1.0f / (float)((len / C4) * cpg);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) s += v[j];
    gn_col_reduce<NT>(s, C4, red, ch, C, t);
    if (t < G) {
        float u = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) u += ch[c];
        mu_s[t] = u * inv_n;
    }
    __syncthreads();
    f32x4 mu;
#pragma unroll
    for (int k = 0; k < 4; ++k) mu[k] = mu_s[(col * 4 + k) / cpg];
    s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const f32x4 d = v[j] - mu;
        if (base + t + j * NT < F4) s += d * d;
    }
    gn_col_reduce<NT>(s, C4, red, ch, C, t);
    if (t < G) {
        float u = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) u += ch[c];
        float* o = part + ((size_t)blockIdx.x * G + t) * 2;
        o[0] = mu_s[t];
        o[1] = u;
    }
}

template <int NT, int NV>
__global__ void __launch_bounds__(NT) gn_chunk_apply_kernel(const GnArgs a, int nchunks, const float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, G = a.groups, cpg = C / G, C4 = C >> 2;
    const int F4 = a.HW * C4;
    const int f = blockIdx.x / nchunks, ck = blockIdx.x % nchunks, t = threadIdx.x, col = t & (C4 - 1);
    const int base = ck * NT * NV;
    float* mu_s = sm;
    float* rs_s = sm + G;
    if (t < G) {
        const float* p = part + ((size_t)f * nchunks * G + t) * 2;
        const float n = (float)(a.HW * cpg);
        float mean = 0.f;
        for (int k = 0; k < nchunks; ++k) {
            const float nk = (float)((min(NT * NV, F4 - k * NT * NV) / C4) * cpg);
            mean += nk * p[(size_t)k * G * 2];
        }
        mean /= n;
        float m2 = 0.f;
        for (int k = 0; k < nchunks; ++k) {
            const float nk = (float)((min(NT * NV, F4 - k * NT * NV) / C4) * cpg);
            const float d = p[(size_t)k * G * 2] - mean;
            m2 += p[(size_t)k * G * 2 + 1] + nk * d * d;
        }
        const float rs = rsqrtf(m2 / n + a.eps);
        mu_s[t] = mean;
        rs_s[t] = rs;
        if (ck == 0) { a.mean[(size_t)f * G + t] = mean; a.rstd[(size_t)f * G + t] = rs; }
    }
    __syncthreads();
    f32x4 sc, sh;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = col * 4 + k, g = c / cpg;
        sc[k] = rs_s[g] * a.gamma[c];
        sh[k] = a.beta[c] - mu_s[g] * sc[k];
    }
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (size_t)f * a.HW * C);
    f32x4* y4 = reinterpret_cast<f32x4*>(a.y + (size_t)f * a.HW * C);
    const f32x4* r4 = a.residual ? reinterpret_cast<const f32x4*>(a.residual + (size_t)f * a.HW * C) : nullptr;
    f32x4 v[NV], rr[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = base + t + j * NT;
        v[j] = i < F4 ? x4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        rr[j] = (r4 && i < F4) ? r4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = base + t + j * NT;
        f32x4 o = v[j] * sc + sh;
        if (r4) o += rr[j];
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
        }
        if (i < F4) y4[i] = o;
    }
}

// y = x * sc + sh as ONE fused multiply-add per element: the stem's ReLU mask is recomputed by the backward pass from the same
// operands, so forward and backward must round identically whatever the compiler's contraction choices
__device__ __forceinline__ f32x4 gn_affine4(const f32x4 x, const f32x4 sc, const f32x4 sh) {
    return f32x4{__builtin_fmaf(x[0], sc[0], sh[0]), __builtin_fmaf(x[1], sc[1], sh[1]), __builtin_fmaf(x[2], sc[2], sh[2]),
                 __builtin_fmaf(x[3], sc[3], sh[3])};
}

// Second kernel of the chunk-parallel forward for the stem: merge the chunk statistics, then GroupNorm-apply + ReLU +
// MaxPool2d(3, 2, 1) in one pass -- the normalised frame (512 KB per frame at 256^2 observations) is neither written nor re-read.
// thread = (pooled pixel, 4 channels).  idx (nullable): window offset of the first maximum in scan order (ATen), for the backward
// pass; a.mean / a.rstd (nullable): the merged statistics.  Bit-identical to GroupNorm -> ReLU -> max-pool over a stored tensor.
// Replaces resnet.py:207-220 (GroupNorm, ReLU, MaxPool2d of `conv1`).
template <int NT, int NV>
__global__ void __launch_bounds__(256) gn_chunk_apply_pool_kernel(const GnArgs a, int nchunks, const float* __restrict__ part, int H, int W,
                                                                  float* __restrict__ pool, uint8_t* __restrict__ idx, int blocks_per_frame,
                                                                  int chunk4 /* float4s per chunk of `part` */) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, G = a.groups, cpg = C / G, C4 = C >> 2;
    const int F4 = a.HW * C4;
    const int f = blockIdx.x / blocks_per_frame, bk = blockIdx.x % blocks_per_frame, t = threadIdx.x;
    float* mu_s = sm;
    float* rs_s = sm + G;
    if (t < G) {  // the merge of gn_chunk_apply_kernel (same order, same arithmetic)
        const float* p = part + ((size_t)f * nchunks * G + t) * 2;
        const float n = (float)(a.HW * cpg);
        float mean = 0.f;
        for (int k = 0; k < nchunks; ++k) {
            const float nk = (float)((min(chunk4, F4 - k * chunk4) / C4) * cpg);
            mean += nk * p[(size_t)k * G * 2];
        }
        mean /= n;
        float m2 = 0.f;
        for (int k = 0; k < nchunks; ++k) {
            const float nk = (float)((min(chunk4, F4 - k * chunk4) / C4) * cpg);
            const float d = p[(size_t)k * G * 2] - mean;
            m2 += p[(size_t)k * G * 2 + 1] + nk * d * d;
        }
        const float rs = rsqrtf(m2 / n + a.eps);
        mu_s[t] = mean;
        rs_s[t] = rs;
        if (bk == 0 && a.mean) { a.mean[(size_t)f * G + t] = mean; a.rstd[(size_t)f * G + t] = rs; }
    }
    __syncthreads();
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int total = Ho * Wo * C4;
    const float* x = a.x + (size_t)f * a.HW * C;
    for (int e = bk * 256 + t; e < total; e += blocks_per_frame * 256) {
        const int c4 = e % C4, pp = e / C4, wo = pp % Wo, ho = pp / Wo;
        f32x4 sc, sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c4 * 4 + k, g = c / cpg;
            sc[k] = rs_s[g] * a.gamma[c];
            sh[k] = a.beta[c] - mu_s[g] * sc[k];
        }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * 2 - 1 + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                f32x4 v = gn_affine4(*reinterpret_cast<const f32x4*>(x + ((size_t)h * W + w) * C + c4 * 4), sc, sh);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = v[k] > 0.f ? v[k] : 0.f;  // ReLU
                    if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = kh * 3 + kw; }
                }
            }
        }
        const size_t o = ((size_t)f * Ho * Wo + pp) * C + c4 * 4;
        *reinterpret_cast<f32x4*>(pool + o) = best;
        if (idx) *reinterpret_cast<uint32_t*>(idx + o) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}

// y = relu(GroupNorm(x)) from the saved statistics (materialises the stem output for the debug taps when the fused forward skipped it)
__global__ void __launch_bounds__(256) gn_relu_from_stats_kernel(const GnArgs a) {
    const int C = a.C, cpg = C / a.groups, C4 = C >> 2;
    const long long total = (long long)a.B * a.HW * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        const int f = (int)(e / ((long long)a.HW * C4));
        f32x4 sc, sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c4 * 4 + k, g = c / cpg;
            sc[k] = a.rstd[(size_t)f * a.groups + g] * a.gamma[c];
            sh[k] = a.beta[c] - a.mean[(size_t)f * a.groups + g] * sc[k];
        }
        f32x4 v = gn_affine4(*reinterpret_cast<const f32x4*>(a.x + e * 4), sc, sh);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        *reinterpret_cast<f32x4*>(a.y + e * 4) = v;
    }
}

// Gradient reaching pixel `pix` of the (never stored) stem output through ReLU + MaxPool2d(3, 2, 1): the sum over the <= 4 pooling
// windows that contain the pixel and chose it (maxpool_bwd_kernel's gather, same order), masked by relu'(y), y recomputed from x.
// (pidx_f / pdy_f: the frame's slices of pool_idx / pool_dy -- the element offsets below are 32-bit)
__device__ __forceinline__ f32x4 gn_pool_dy(const GnBwdArgs& a, const uint8_t* __restrict__ pidx_f, const float* __restrict__ pdy_f, int h, int w,
                                            int c4, const f32x4 xv, const f32x4 sc, const f32x4 sh) {
    const int H = a.pH, W = a.pW, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    // The candidate windows are (h / 2, (h + 1) / 2) x (w / 2, (w + 1) / 2): one per axis for even coordinates, two for odd ones.  All four
    // (index word, gradient) pairs are fetched up front from clamped coordinates -- eight independent loads instead of a data-dependent
    // loop nest whose loads wait for each other -- and folded in the loop nest's order, so the sum is bit-identical to maxpool_bwd_kernel's.
    const int ho0 = h >> 1, ho1 = (h + 1) >> 1, wo0 = w >> 1, wo1 = (w + 1) >> 1;
    const bool vh[2] = {ho0 < Ho, ho1 != ho0 && ho1 < Ho}, vw[2] = {wo0 < Wo, wo1 != wo0 && wo1 < Wo};
    const int hh[2] = {min(ho0, Ho - 1), min(ho1, Ho - 1)}, ww[2] = {min(wo0, Wo - 1), min(wo1, Wo - 1)};
    uint32_t id[2][2];
    f32x4 d[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = (hh[p] * Wo + ww[q]) * a.C + c4 * 4;
            id[p][q] = *reinterpret_cast<const uint32_t*>(pidx_f + o);
            d[p][q] = *reinterpret_cast<const f32x4*>(pdy_f + o);
        }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kh = h - ((p ? ho1 : ho0) * 2 - 1), kw = w - ((q ? wo1 : wo0) * 2 - 1);
            const uint32_t me = (uint32_t)(kh * 3 + kw);
            const bool ok = vh[p] && vw[q];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok && ((id[p][q] >> (8 * k)) & 0xffu) == me) s[k] += d[p][q][k];
        }
    const f32x4 y = gn_affine4(xv, sc, sh);
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = y[k] > 0.f ? s[k] : 0.f;
    return s;
}

// backward, kernel 1: per-channel partial sums S1 = sum dy', S2 = sum dy' * xhat of the chunk (+ optional dy' write-out)
template <int NT, int NV>
__global__ void __launch_bounds__(NT) gn_chunk_bwd_sums_kernel(const GnBwdArgs a, int nchunks, float* __restrict__ part, const int reps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, G = a.groups, cpg = C / G, C4 = C >> 2;
    const int F4 = a.HW * C4;
    const int f = blockIdx.x / nchunks, ck = blockIdx.x % nchunks, t = threadIdx.x, col = t & (C4 - 1);
    float* red = sm;
    float* c1 = sm + NT * 4;
    const size_t fb = (size_t)f * a.HW * C;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + fb);
    const bool pooled = a.pool_dy != nullptr;
    const int c4_shift = __builtin_ctz(C4), pstep = NT >> c4_shift;           // pooled form: pixels between two units of a thread
    const int step_h = pooled ? pstep / a.pW : 0, step_w = pooled ? pstep - step_h * a.pW : 0;
    const size_t pool_f = pooled ? (size_t)f * ((a.pH + 1) / 2) * ((a.pW + 1) / 2) * C : 0;   // MaxPool2d(3, 2, 1): Ho = (H + 1) / 2
    const uint8_t* pidx_f = pooled ? a.pool_idx + pool_f : nullptr;
    const float* pdy_f = pooled ? a.pool_dy + pool_f : nullptr;
    const f32x4* dy4 = pooled ? nullptr : reinterpret_cast<const f32x4*>(a.dy + fb);
    const f32x4* ro4 = (a.relu_out && !pooled) ? reinterpret_cast<const f32x4*>(a.relu_out + fb) : nullptr;
    f32x4* dym4 = a.dy_masked ? reinterpret_cast<f32x4*>(a.dy_masked + fb) : nullptr;
    f32x4 mu, rs, psc = {0.f, 0.f, 0.f, 0.f}, psh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int g = (col * 4 + k) / cpg; mu[k] = a.mean[(size_t)f * G + g]; rs[k] = a.rstd[(size_t)f * G + g]; }
    if (pooled) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { psc[k] = rs[k] * a.gamma[col * 4 + k]; psh[k] = a.beta[col * 4 + k] - mu[k] * psc[k]; }
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (int rep = 0; rep < reps; ++rep) {
    const int base = (ck * reps + rep) * NT * NV;
    if (base >= F4) break;
    f32x4 xv[NV], dv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = base + t + j * NT;
        const bool ok = i < F4;
        xv[j] = ok ? x4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (!pooled) dv[j] = ok ? dy4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (pooled) {
        // pixel of unit j: (base + t) / C4 + j * (NT / C4); C4 is a power of two (gn_chunk_cfg), the row / column walk forward by a
        // workgroup-uniform step -- one division per chunk instead of two per element
        int ph, pw;
        { const int pix0 = (base + t) >> c4_shift; ph = pix0 / a.pW; pw = pix0 - ph * a.pW; }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = base + t + j * NT;
            dv[j] = i < F4 ? gn_pool_dy(a, pidx_f, pdy_f, ph, pw, col, xv[j], psc, psh) : f32x4{0.f, 0.f, 0.f, 0.f};
            ph += step_h; pw += step_w;
            if (pw >= a.pW) { pw -= a.pW; ++ph; }
        }
    }
    if (ro4) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = base + t + j * NT;
            if (i < F4) {
                const f32x4 rv = ro4[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[j][k] = rv[k] > 0.f ? dv[j][k] : 0.f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = base + t + j * NT;
        if (dym4 && i < F4) dym4[i] = dv[j];
        f32x4 xh = (xv[j] - mu) * rs;
        if (i >= F4) xh = f32x4{0.f, 0.f, 0.f, 0.f};
        s1 += dv[j];
        s2 += dv[j] * xh;
    }
    }
    float* o = part + (size_t)blockIdx.x * 2 * C;
    gn_col_reduce<NT>(s1, C4, red, c1, C, t);
    for (int c = t; c < C; c += NT) o[c] = c1[c];
    gn_col_reduce<NT>(s2, C4, red, c1, C, t);
    for (int c = t; c < C; c += NT) o[C + c] = c1[c];
}

// backward, kernel 2: merge the partial sums of the frame (fixed order), then dx for the chunk
template <int NT, int NV>
__global__ void __launch_bounds__(NT) gn_chunk_bwd_dx_kernel(const GnBwdArgs a, int nchunks, const float* __restrict__ part, const int reps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, G = a.groups, cpg = C / G, C4 = C >> 2;
    const int F4 = a.HW * C4;
    const int f = blockIdx.x / nchunks, ck = blockIdx.x % nchunks, t = threadIdx.x, col = t & (C4 - 1);
    float* c1 = sm;          // [C]
    float* c2 = sm + C;      // [C]
    float* g1 = c2 + C;      // [G]
    float* g2 = g1 + G;
    const float* p = part + (size_t)f * nchunks * 2 * C;
    for (int c = t; c < C; c += NT) {
        float u = 0.f, w = 0.f;
        for (int k = 0; k < nchunks; ++k) { u += p[(size_t)k * 2 * C + c]; w += p[(size_t)k * 2 * C + C + c]; }
        c1[c] = u; c2[c] = w;
        if (ck == 0) {
            a.chan_sums[((size_t)f * 2 + 0) * C + c] = u;
            a.chan_sums[((size_t)f * 2 + 1) * C + c] = w;
        }
    }
    __syncthreads();
    if (t < G) {
        float u = 0.f, w = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) { u += a.gamma[c] * c1[c]; w += a.gamma[c] * c2[c]; }
        const float inv_m = 1.0f / (float)(a.HW * cpg);
        g1[t] = u * inv_m; g2[t] = w * inv_m;
    }
    __syncthreads();
    f32x4 mu, rs, ga, m1, m2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = col * 4 + k, g = c / cpg;
        mu[k] = a.mean[(size_t)f * G + g]; rs[k] = a.rstd[(size_t)f * G + g]; ga[k] = a.gamma[c]; m1[k] = g1[g]; m2[k] = g2[g];
    }
    const size_t fb = (size_t)f * a.HW * C;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + fb);
    // the masked gradient was written out by kernel 1 when the caller wanted it: read that instead of dy + relu_out
    const bool pooled = a.pool_dy != nullptr && !a.dy_masked;
    const int c4_shift = __builtin_ctz(C4), pstep = NT >> c4_shift;
    const int step_h = pooled ? pstep / a.pW : 0, step_w = pooled ? pstep - step_h * a.pW : 0;
    const size_t pool_f = pooled ? (size_t)f * ((a.pH + 1) / 2) * ((a.pW + 1) / 2) * C : 0;
    const uint8_t* pidx_f = pooled ? a.pool_idx + pool_f : nullptr;
    const float* pdy_f = pooled ? a.pool_dy + pool_f : nullptr;
    const f32x4* dy4 = pooled ? nullptr : reinterpret_cast<const f32x4*>((a.dy_masked ? a.dy_masked : a.dy) + fb);
    const f32x4* ro4 = (a.relu_out && !a.dy_masked && !pooled) ? reinterpret_cast<const f32x4*>(a.relu_out + fb) : nullptr;
    f32x4* dx4 = reinterpret_cast<f32x4*>(a.dx + fb);
    for (int rep = 0; rep < reps; ++rep) {
    const int base = (ck * reps + rep) * NT * NV;
    if (base >= F4) break;
    f32x4 xv[NV], dv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = base + t + j * NT;
        const bool ok = i < F4;
        xv[j] = ok ? x4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (!pooled) dv[j] = ok ? dy4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (pooled) {
        f32x4 psc, psh;
#pragma unroll
        for (int k = 0; k < 4; ++k) { psc[k] = rs[k] * ga[k]; psh[k] = a.beta[col * 4 + k] - mu[k] * psc[k]; }
        int ph, pw;
        { const int pix0 = (base + t) >> c4_shift; ph = pix0 / a.pW; pw = pix0 - ph * a.pW; }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = base + t + j * NT;
            dv[j] = i < F4 ? gn_pool_dy(a, pidx_f, pdy_f, ph, pw, col, xv[j], psc, psh) : f32x4{0.f, 0.f, 0.f, 0.f};
            ph += step_h; pw += step_w;
            if (pw >= a.pW) { pw -= a.pW; ++ph; }
        }
    }
    if (ro4) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = base + t + j * NT;
            if (i < F4) {
                const f32x4 rv = ro4[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[j][k] = rv[k] > 0.f ? dv[j][k] : 0.f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = base + t + j * NT;
        const f32x4 xh = (xv[j] - mu) * rs;
        if (i < F4) dx4[i] = rs * (dv[j] * ga - m1 - xh * m2);
    }
    }
}

// chunk-parallel path applicable?  (power-of-two C/4 that divides the workgroup, scratch large enough)
static bool gn_stream_only() { static const bool v = hab_env_flag("HAB_GN_STREAM"); return v; }
static int gn_chunk_cfg(int B, int HW, int C, size_t need_per_chunk, size_t scratch_floats, int chunk_f4, int& nchunks) {
    const int C4 = C / 4;
    if (C4 < 1 || (C4 & (C4 - 1)) || C4 > GNC_NT || gn_stream_only()) return 0;
    const long long F4 = (long long)HW * C4;
    nchunks = (int)((F4 + chunk_f4 - 1) / chunk_f4);
    if (nchunks < 2 || (size_t)B * nchunks * need_per_chunk > scratch_floats) return 0;
    if ((long long)B * nchunks > 0x7fffffffLL) return 0;
    return 1;
}

// Picks (NT, NV) for the register-resident kernels: 0 = not applicable (frame too large / channel count not a power of two).
static int gn_reg_cfg(int HW, int C, int& nt, int& nv) {
    const int C4 = C / 4;
    if (C4 < 1 || (C4 & (C4 - 1)) || gn_stream_only()) return 0;
    const long long F4 = (long long)HW * C4;
    nt = F4 > 2048 ? 1024 : 256;
    if (C4 > nt) return 0;
    const long long need = (F4 + nt - 1) / nt;
    if (need > 8) return 0;  // 16 float4 per thread spill at 1024 threads (128-VGPR budget)
    nv = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    return 1;
}
#define HAB_GN_REG_DISPATCH(KERNEL, ARGS, LDS)                                                                 \
    do {                                                                                                       \
        if (nt == 256) {                                                                                       \
            switch (nv) {                                                                                      \
            case 1: KERNEL<256, 1><<<ARGS.B, 256, LDS, s>>>(ARGS); break;                                      \
            case 2: KERNEL<256, 2><<<ARGS.B, 256, LDS, s>>>(ARGS); break;                                      \
            case 4: KERNEL<256, 4><<<ARGS.B, 256, LDS, s>>>(ARGS); break;                                      \
            default: KERNEL<256, 8><<<ARGS.B, 256, LDS, s>>>(ARGS); break;                                     \
            }                                                                                                  \
        } else {                                                                                               \
            switch (nv) {                                                                                      \
            case 1: KERNEL<1024, 1><<<ARGS.B, 1024, LDS, s>>>(ARGS); break;                                    \
            case 2: KERNEL<1024, 2><<<ARGS.B, 1024, LDS, s>>>(ARGS); break;                                    \
            case 4: KERNEL<1024, 4><<<ARGS.B, 1024, LDS, s>>>(ARGS); break;                                    \
            default: KERNEL<1024, 8><<<ARGS.B, 1024, LDS, s>>>(ARGS); break;                                   \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

static int gn_check(int B, int C, int groups) {
    if (B <= 0 || C <= 0 || groups <= 0 || C % 4 || C % groups || groups > 256) return HAB_ERR_ARG;
    const int C4 = C / 4;
    if (C4 < 256 ? (256 % C4 != 0) : (C4 % 256 != 0)) return HAB_ERR_UNSUPPORTED;
    return HAB_OK;
}

int groupnorm_forward(const GnArgs& a, hipStream_t s) {
    if (!a.x || !a.y || !a.gamma || !a.beta || !a.mean || !a.rstd) return HAB_ERR_ARG;
    HAB_TRY(gn_check(a.B, a.C, a.groups));
    int nt, nv;
    if (gn_reg_cfg(a.HW, a.C, nt, nv)) {
        const size_t lds_r = (size_t)(nt * 4 + a.C + 2 * a.groups) * sizeof(float);
        HAB_GN_REG_DISPATCH(groupnorm_fwd_reg_kernel, a, lds_r);
        HAB_LAUNCH_CHECK();
        return HAB_OK;
    }
    int nchunks;
    if (a.scratch && gn_chunk_cfg(a.B, a.HW, a.C, (size_t)a.groups * 2, a.scratch_floats, GNC_NT * GNC_NV_F, nchunks)) {
        const size_t lds1 = (size_t)(GNC_NT * 4 + a.C + a.groups) * sizeof(float);
        gn_chunk_stats_kernel<GNC_NT, GNC_NV_F><<<a.B * nchunks, GNC_NT, lds1, s>>>(a, nchunks, a.scratch);
        HAB_LAUNCH_CHECK();
        gn_chunk_apply_kernel<GNC_NT, GNC_NV_F><<<a.B * nchunks, GNC_NT, 2 * a.groups * sizeof(float), s>>>(a, nchunks, a.scratch);
        HAB_LAUNCH_CHECK();
        return HAB_OK;
    }
    const size_t lds = (size_t)(1024 + a.C + 2 * a.groups) * sizeof(float);
    groupnorm_fwd_kernel<<<a.B, 256, lds, s>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// GroupNorm + ReLU + MaxPool2d(3, 2, 1) in one pass; 1: frame not on the chunk-parallel path (the caller runs groupnorm_forward +
// maxpool_forward).
// ext_part / ext_rows: the partial statistics already exist -- (mean, M2) per (frame, chunk of ext_rows image rows, group), written by the
// producing convolution's epilogue (stem_conv_strip.h) -- and the statistics pass over the tensor is skipped.
int groupnorm_relu_maxpool_forward(const GnArgs& a, int H, int W, float* pool, uint8_t* idx, hipStream_t s, const float* ext_part, int ext_rows) {
    if (!a.x || !a.gamma || !a.beta || !pool || a.HW != H * W || a.residual || !a.relu) return HAB_ERR_ARG;
    if ((a.mean == nullptr) != (a.rstd == nullptr)) return HAB_ERR_ARG;
    HAB_TRY(gn_check(a.B, a.C, a.groups));
    int nt, nv, nchunks;
    if (gn_reg_cfg(a.HW, a.C, nt, nv)) return 1;
    const float* part = a.scratch;
    int chunk4 = GNC_NT * GNC_NV_F;
    if (ext_part) {
        if (ext_rows < 1) return HAB_ERR_ARG;
        part = ext_part;
        chunk4 = ext_rows * W * (a.C / 4);
        nchunks = cdiv(H, ext_rows);
    } else {
        if (!a.scratch || !gn_chunk_cfg(a.B, a.HW, a.C, (size_t)a.groups * 2, a.scratch_floats, GNC_NT * GNC_NV_F, nchunks)) return 1;
        const size_t lds1 = (size_t)(GNC_NT * 4 + a.C + a.groups) * sizeof(float);
        gn_chunk_stats_kernel<GNC_NT, GNC_NV_F><<<a.B * nchunks, GNC_NT, lds1, s>>>(a, nchunks, a.scratch);
        HAB_LAUNCH_CHECK();
    }
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int bpf = max(1, min(64, cdiv(Ho * Wo * (a.C / 4), 256 * 4)));
    gn_chunk_apply_pool_kernel<GNC_NT, GNC_NV_F><<<a.B * bpf, 256, 2 * a.groups * sizeof(float), s>>>(a, nchunks, part, H, W, pool, idx, bpf, chunk4);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
bool groupnorm_pool_fusable(int B, int HW, int C, int groups, size_t scratch_floats) {
    int nt, nv, n1, n2;
    if (gn_check(B, C, groups) != HAB_OK || gn_reg_cfg(HW, C, nt, nv)) return false;
    return gn_chunk_cfg(B, HW, C, (size_t)groups * 2, scratch_floats, GNC_NT * GNC_NV_F, n1) &&
           gn_chunk_cfg(B, HW, C, (size_t)C * 2, scratch_floats, GNC_NT * GNC_NV_B, n2);  // (the stem form's backward: one sub-chunk)
}
int groupnorm_relu_materialize(const GnArgs& a, hipStream_t s) {
    if (!a.x || !a.y || !a.gamma || !a.beta || !a.mean || !a.rstd || a.B <= 0 || (a.C & 3) || a.C % a.groups) return HAB_ERR_ARG;
    const long long total = (long long)a.B * a.HW * (a.C / 4);
    gn_relu_from_stats_kernel<<<(int)fmin(32768.0, (double)cdivl(total, 256)), 256, 0, s>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

__global__ void __launch_bounds__(256) groupnorm_bwd_kernel(const GnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, HW = a.HW, G = a.groups, cpg = C / G;
    const GnMap m(C);
    const int f = blockIdx.x, t = threadIdx.x;
    const int pl = m.lane(t);
    const bool act = pl < m.np;
    float* red = sm;            // [256][4]
    float* c1 = sm + 1024;      // [C]  per-channel S1
    float* c2 = c1 + C;         // [C]  per-channel S2
    float* g1 = c2 + C;         // [G]  mean_g(g), mean_g(g*xhat)
    float* g2 = g1 + G;
    const size_t base = (size_t)f * HW * C;
    const float* x = a.x + base;
    const float* dy = a.dy + base;
    const float* ro = a.relu_out ? a.relu_out + base : nullptr;
    float* dym = a.dy_masked ? a.dy_masked + base : nullptr;
    const float* mean = a.mean + (size_t)f * G;
    const float* rstd = a.rstd + (size_t)f * G;
    for (int cc = 0; cc < m.ncc; ++cc) {
        const int col = m.col(t, cc);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            f32x4 mu, rs;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int g = (col * 4 + k) / cpg; mu[k] = mean[g]; rs[k] = rstd[g]; }
            for (int p = pl; p < HW; p += m.np) {
                const size_t o = (size_t)p * C + col * 4;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
                f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
                if (ro) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(ro + o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) dv[k] = rv[k] > 0.f ? dv[k] : 0.f;
                }
                if (dym) *reinterpret_cast<f32x4*>(dym + o) = dv;
                s1 += dv;
                s2 += dv * ((xv - mu) * rs);
            }
        }
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + t * 4) = act ? s1 : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        gn_fold(m, cc, red, c1, t);
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + t * 4) = act ? s2 : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        gn_fold(m, cc, red, c2, t);
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        a.chan_sums[((size_t)f * 2 + 0) * C + c] = c1[c];   // -> dbeta after the reduction over frames
        a.chan_sums[((size_t)f * 2 + 1) * C + c] = c2[c];   // -> dgamma
    }
    if (t < G) {
        float u = 0.f, v = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) { u += a.gamma[c] * c1[c]; v += a.gamma[c] * c2[c]; }
        const float inv_m = 1.0f / (float)(HW * cpg);
        g1[t] = u * inv_m; g2[t] = v * inv_m;
    }
    __syncthreads();
    if (!act) return;
    float* dx = a.dx + base;
    for (int cc = 0; cc < m.ncc; ++cc) {
        const int col = m.col(t, cc);
        f32x4 mu, rs, ga, m1, m2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = col * 4 + k, g = c / cpg;
            mu[k] = mean[g]; rs[k] = rstd[g]; ga[k] = a.gamma[c]; m1[k] = g1[g]; m2[k] = g2[g];
        }
        for (int p = pl; p < HW; p += m.np) {
            const size_t o = (size_t)p * C + col * 4;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
            f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
            if (ro) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(ro + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[k] = rv[k] > 0.f ? dv[k] : 0.f;
            }
            const f32x4 xh = (xv - mu) * rs;
            *reinterpret_cast<f32x4*>(dx + o) = rs * (dv * ga - m1 - xh * m2);
        }
    }
}

int groupnorm_backward(const GnBwdArgs& a, hipStream_t s) {
    if (!a.x || (!a.dy && !a.pool_dy) || !a.dx || !a.gamma || !a.mean || !a.rstd || !a.chan_sums) return HAB_ERR_ARG;
    if (a.dx == a.dy && a.dy_masked) return HAB_ERR_ARG;
    HAB_TRY(gn_check(a.B, a.C, a.groups));
    int nt, nv;
    if (gn_reg_cfg(a.HW, a.C, nt, nv)) {
        if (a.pool_dy) return HAB_ERR_UNSUPPORTED;  // the stem form exists on the chunk-parallel path only (groupnorm_pool_fusable)
        const size_t lds_r = (size_t)(nt * 4 + 2 * a.C + 2 * a.groups) * sizeof(float);
        HAB_GN_REG_DISPATCH(groupnorm_bwd_reg_kernel, a, lds_r);
        HAB_LAUNCH_CHECK();
        return HAB_OK;
    }
    int nchunks;
    // sub-chunks per workgroup: the stem form (gathers through the pooling windows per element) measured SLOWER with them (its statistics
    // kernel 1.22 -> 1.79 ms per 4096 frames: profiles r06, commits 075041b vs b3f3464), the plain form faster (NOTEBOOK R6.8)
    const int reps = a.pool_dy ? 1 : gnc_reps_b();
    // (2 or 4 units per thread instead of 8 for this form -- more, smaller workgroups -- measured slower: C3 247.7 / 251.2 / 263.6 ms)
    if (a.scratch && gn_chunk_cfg(a.B, a.HW, a.C, (size_t)a.C * 2, a.scratch_floats, GNC_NT * GNC_NV_B * reps, nchunks)) {
        const size_t lds1 = (size_t)(GNC_NT * 4 + a.C) * sizeof(float);
        if (a.pool_dy && (!a.pool_idx || !a.beta || a.pH * a.pW != a.HW)) return HAB_ERR_ARG;
        gn_chunk_bwd_sums_kernel<GNC_NT, GNC_NV_B><<<a.B * nchunks, GNC_NT, lds1, s>>>(a, nchunks, a.scratch, reps);
        HAB_LAUNCH_CHECK();
        const size_t lds2 = (size_t)(2 * a.C + 2 * a.groups) * sizeof(float);
        gn_chunk_bwd_dx_kernel<GNC_NT, GNC_NV_B><<<a.B * nchunks, GNC_NT, lds2, s>>>(a, nchunks, a.scratch, reps);
        HAB_LAUNCH_CHECK();
        return HAB_OK;
    }
    if (a.pool_dy) return HAB_ERR_UNSUPPORTED;  // the stem form exists on the chunk-parallel path only (groupnorm_pool_fusable)
    const size_t lds = (size_t)(1024 + 2 * a.C + 2 * a.groups) * sizeof(float);
    groupnorm_bwd_kernel<<<a.B, 256, lds, s>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, padding 1) (resnet.py:220): forward keeps the window offset of the first maximum
// (ATen scan order); backward gathers over the <= 4 windows that contain an input pixel.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int B, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, C4 = C >> 2;
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        long long t = e / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int f = (int)(t / Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * 2 - 1 + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)f * H + h) * W + w) * C + c4 * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = kh * 3 + kw; }  // first maximum in scan order (ATen)
            }
        }
        *reinterpret_cast<f32x4*>(y + e * 4) = best;
        *reinterpret_cast<uint32_t*>(idx + e * 4) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int B, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, C4 = C >> 2;
    const long long total = (long long)B * H * W * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        long long t = e / C4;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int f = (int)(t / H);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        // windows (ho, wo) with ho*2-1 <= h <= ho*2+1
        for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
            if (ho >= Ho) continue;
            const int kh = h - (ho * 2 - 1);
            for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
                if (wo >= Wo) continue;
                const int kw = w - (wo * 2 - 1);
                const size_t o = (((size_t)f * Ho + ho) * Wo + wo) * C + c4 * 4;
                const uint32_t id = *reinterpret_cast<const uint32_t*>(idx + o);
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
                const uint32_t me = (uint32_t)(kh * 3 + kw);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (((id >> (8 * k)) & 0xffu) == me) s[k] += d[k];
            }
        }
        *reinterpret_cast<f32x4*>(dx + e * 4) = s;
    }
}
int maxpool_forward(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, hipStream_t s) {
    if (!x || !y || !idx || B <= 0 || (C & 3)) return HAB_ERR_ARG;
    const long long total = (long long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 4);
    maxpool_fwd_kernel<<<(int)fmin(32768.0, (double)cdivl(total, 256)), 256, 0, s>>>(x, y, idx, B, H, W, C);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
int maxpool_backward(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, hipStream_t s) {
    if (!dy || !idx || !dx || B <= 0 || (C & 3)) return HAB_ERR_ARG;
    const long long total = (long long)B * H * W * (C / 4);
    maxpool_bwd_kernel<<<(int)fmin(32768.0, (double)cdivl(total, 256)), 256, 0, s>>>(dy, idx, dx, B, H, W, C);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// 1-D sensor embeddings (resnet_ops.h: EmbedSlot) written into the RNN input:
//   out[f][col0 + 32*s .. +32) = slot s.   forward keeps features / tokens in saved[f][s][0..3].
// backward: deterministic two-stage reduction (fixed-order frame loops), one output row per (slot, feature | bias | token).
// ------------------------------------------------------------------------------------------------------
__device__ inline void emb_features(const EmbedSlot& sl, int r, int masked, float* v /*[4]*/) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    switch (sl.kind) {
        case EMB_POLAR: {
            const float* g = reinterpret_cast<const float*>(sl.in) + (size_t)r * 2;
            v[0] = g[0]; v[1] = cosf(-g[1]); v[2] = sinf(-g[1]);
            break;
        }
        case EMB_TOKEN: v[3] = (float)reinterpret_cast<const int64_t*>(sl.in)[r]; break;
        case EMB_COSSIN: {
            const float x = reinterpret_cast<const float*>(sl.in)[r];
            v[0] = cosf(x); v[1] = sinf(x);
            break;
        }
        case EMB_LIN2: {
            const float* g = reinterpret_cast<const float*>(sl.in) + (size_t)r * 2;
            v[0] = g[0]; v[1] = g[1];
            break;
        }
        case EMB_PREVLIN: {
            const float* g = reinterpret_cast<const float*>(sl.in) + (size_t)r * sl.ntok;
            for (int d = 0; d < sl.ntok; ++d) v[d] = masked ? g[d] : 0.f;  // masks * prev_actions.float()
            break;
        }
        case EMB_LINN: {
            const float* g = reinterpret_cast<const float*>(sl.in) + (size_t)r * sl.ntok;
            for (int d = 0; d < sl.ntok; ++d) v[d] = g[d];
            break;
        }
        default: v[3] = masked ? (float)(reinterpret_cast<const int64_t*>(sl.in)[r] + 1) : 0.f; break;
    }
}

__global__ void __launch_bounds__(256) embed_fwd_kernel(const EmbedArgs a) {
    const int f = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
    if (f >= a.B) return;
    const int r = a.rows ? a.rows[f] : f;
    const int m = a.masks ? (int)a.masks[r] : 1;
    float* o = a.out + (size_t)f * a.ld + a.col0;
    for (int s = 0; s < a.nslots; ++s) {
        const EmbedSlot& sl = a.slot[s];
        float v[4];
        emb_features(sl, r, m, v);
        const int nf = emb_nfeat(sl);
        float y;
        if (sl.kind == EMB_PREVLIN || sl.kind == EMB_LINN) {
            y = 0.f;
            for (int d = 0; d < nf; ++d) y += sl.w[j * nf + d] * v[d];
            y += sl.b[j];
        } else if (nf == 3) y = (sl.w[j * 3] * v[0] + sl.w[j * 3 + 1] * v[1]) + sl.w[j * 3 + 2] * v[2] + sl.b[j];
        else if (nf == 2) y = sl.w[j * 2] * v[0] + sl.w[j * 2 + 1] * v[1] + sl.b[j];
        else y = sl.w[(size_t)(int)v[3] * 32 + j];
        o[s * 32 + j] = y;
        if (j < 4 && a.saved) a.saved[((size_t)f * a.nslots + s) * 4 + j] = v[j];
    }
}
int embed_forward(const EmbedArgs& a, hipStream_t s) {
    if (!a.out || a.B <= 0 || a.nslots <= 0 || a.nslots > EMB_MAX_SLOTS) return HAB_ERR_ARG;
    for (int i = 0; i < a.nslots; ++i)
        if (!a.slot[i].in || !a.slot[i].w || (emb_nfeat(a.slot[i]) && !a.slot[i].b)) return HAB_ERR_ARG;
    embed_fwd_kernel<<<cdiv(a.B, 8), 256, 0, s>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

struct EmbRowMap { int slot[64]; int sub[64]; int n; };  // output row -> (slot, feature index | nfeat = bias | token)

__global__ void __launch_bounds__(256) embed_bwd_stage1(const EmbedBwdArgs a, const EmbRowMap rm, int frames_per_chunk,
                                                        float* __restrict__ partial) {
    __shared__ float sm[256];
    const int b = blockIdx.x, j = threadIdx.x & 31, fl = threadIdx.x >> 5;
    const int s = rm.slot[b], sub = rm.sub[b];
    const int nf = emb_nfeat(a.slot[s]);
    const int f0 = blockIdx.y * frames_per_chunk, f1 = min(a.B, f0 + frames_per_chunk);
    float acc = 0.f;
    for (int f = f0 + fl; f < f1; f += 8) {
        const float d = a.dout[(size_t)f * a.ld + a.col0 + s * 32 + j];
        const float* sv = a.saved + ((size_t)f * a.nslots + s) * 4;
        if (nf) acc += sub < nf ? d * sv[sub] : d;
        else if (sv[3] == (float)sub) acc += d;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (fl == 0) {
        float t = 0.f;
        for (int q = 0; q < 8; ++q) t += sm[q * 32 + j];
        partial[((size_t)blockIdx.y * gridDim.x + b) * 32 + j] = t;
    }
}
__global__ void __launch_bounds__(32) embed_bwd_stage2(const EmbedBwdArgs a, const EmbRowMap rm, const float* __restrict__ partial,
                                                       int chunks) {
    const int b = blockIdx.x, j = threadIdx.x;
    float acc = 0.f;
    for (int c = 0; c < chunks; ++c) acc += partial[((size_t)c * rm.n + b) * 32 + j];
    const EmbedSlot& sl = a.slot[rm.slot[b]];
    const int nf = emb_nfeat(sl);
    const int sub = rm.sub[b];
    if (nf == 0) sl.dw[(size_t)sub * 32 + j] = acc;
    else if (sub < nf) sl.dw[j * nf + sub] = acc;
    else sl.db[j] = acc;
}
int embed_backward(const EmbedBwdArgs& a, float* ws, size_t ws_floats, hipStream_t s) {
    if (!a.saved || !a.dout || a.B <= 0 || !ws || a.nslots <= 0 || a.nslots > EMB_MAX_SLOTS) return HAB_ERR_ARG;
    EmbRowMap rm;
    rm.n = 0;
    for (int i = 0; i < a.nslots; ++i) {
        const int nf = emb_nfeat(a.slot[i]);
        const int rows = nf ? nf + 1 : a.slot[i].ntok;
        if (!a.slot[i].dw || (nf && !a.slot[i].db) || rows <= 0 || rm.n + rows > 64) return HAB_ERR_ARG;
        for (int q = 0; q < rows; ++q) { rm.slot[rm.n] = i; rm.sub[rm.n] = q; ++rm.n; }
    }
    int chunks = cdiv(a.B, 64);
    if (chunks > 128) chunks = 128;
    if ((size_t)chunks * rm.n * 32 > ws_floats) return HAB_ERR_ARG;
    const int fpc = cdiv(a.B, chunks);
    chunks = cdiv(a.B, fpc);
    embed_bwd_stage1<<<dim3(rm.n, chunks), 256, 0, s>>>(a, rm, fpc, ws);
    HAB_LAUNCH_CHECK();
    embed_bwd_stage2<<<rm.n, 32, 0, s>>>(a, rm, ws, chunks);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// Squeeze-and-excitation (resnet.py:92-113): block per frame, thread per channel quad, fixed-order loops over the frame.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) se_pool_kernel(const float* __restrict__ x, float* __restrict__ pooled, int HW, int C) {
    const int b = blockIdx.x, C4 = C >> 2;
    const f32x4* xb = reinterpret_cast<const f32x4*>(x + (size_t)b * HW * C);
    for (int c = threadIdx.x; c < C4; c += 256) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // fp64 accumulation: the mean of a zero-centred GroupNorm output cancels heavily
        for (int p = 0; p < HW; ++p) {
            const f32x4 v = xb[(size_t)p * C4 + c];
            s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
        }
        f32x4 o;
        o[0] = (float)(s0 / HW); o[1] = (float)(s1 / HW); o[2] = (float)(s2 / HW); o[3] = (float)(s3 / HW);
        *reinterpret_cast<f32x4*>(pooled + (size_t)b * C + c * 4) = o;
    }
}
int se_pool(const float* x, float* pooled, int B, int HW, int C, hipStream_t s) {
    if (!x || !pooled || B <= 0 || HW <= 0 || (C & 3)) return HAB_ERR_ARG;
    se_pool_kernel<<<B, 256, 0, s>>>(x, pooled, HW, C);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
__global__ void __launch_bounds__(256) se_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                           const float* __restrict__ residual, float* __restrict__ y, long long total4,
                                                           int HWC4, int C4) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int b = (int)(e / HWC4), c = (int)(e % C4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gate + ((size_t)b * C4 + c) * 4);
        f32x4 v = g * reinterpret_cast<const f32x4*>(x)[e] + reinterpret_cast<const f32x4*>(residual)[e];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        reinterpret_cast<f32x4*>(y)[e] = v;
    }
}
int se_apply_forward(const float* x, const float* gate, const float* residual, float* y, int B, int HW, int C, hipStream_t s) {
    if (!x || !gate || !residual || !y || B <= 0 || (C & 3)) return HAB_ERR_ARG;
    const long long total4 = (long long)B * HW * (C >> 2);
    se_apply_fwd_kernel<<<(int)fmin(32768.0, (double)cdivl(total4, 256)), 256, 0, s>>>(x, gate, residual, y, total4, HW * (C >> 2), C >> 2);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
__global__ void __launch_bounds__(256) se_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ x, float* __restrict__ dm,
                                                            float* __restrict__ dgate, int HW, int C) {
    const int b = blockIdx.x, C4 = C >> 2;
    const size_t fb = (size_t)b * HW * C4;
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy) + fb;
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y) + fb;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + fb;
    f32x4* dm4 = reinterpret_cast<f32x4*>(dm) + fb;
    for (int c = threadIdx.x; c < C4; c += 256) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int p = 0; p < HW; ++p) {
            const size_t i = (size_t)p * C4 + c;
            f32x4 d = dy4[i];
            const f32x4 yv = y4[i], xv = x4[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) { d[k] = yv[k] > 0.f ? d[k] : 0.f; s[k] += (double)d[k] * (double)xv[k]; }
            dm4[i] = d;
        }
        f32x4 o;
        o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = (float)s[2]; o[3] = (float)s[3];
        *reinterpret_cast<f32x4*>(dgate + (size_t)b * C + c * 4) = o;
    }
}
int se_backward_reduce(const float* dy, const float* y, const float* x, float* dm, float* dgate, int B, int HW, int C, hipStream_t s) {
    if (!dy || !y || !x || !dm || !dgate || B <= 0 || (C & 3)) return HAB_ERR_ARG;
    se_bwd_reduce_kernel<<<B, 256, 0, s>>>(dy, y, x, dm, dgate, HW, C);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
__global__ void __launch_bounds__(256) se_bwd_apply_kernel(const float* __restrict__ dm, const float* __restrict__ gate,
                                                           const float* __restrict__ dpool, float* __restrict__ dx, long long total4,
                                                           int HWC4, int C4, float inv_hw) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int b = (int)(e / HWC4), c = (int)(e % C4);
        const size_t o = ((size_t)b * C4 + c) * 4;
        reinterpret_cast<f32x4*>(dx)[e] = reinterpret_cast<const f32x4*>(dm)[e] * *reinterpret_cast<const f32x4*>(gate + o) +
                                          *reinterpret_cast<const f32x4*>(dpool + o) * inv_hw;
    }
}
int se_backward_apply(const float* dm, const float* gate, const float* dpool, float* dx, int B, int HW, int C, hipStream_t s) {
    if (!dm || !gate || !dpool || !dx || B <= 0 || (C & 3)) return HAB_ERR_ARG;
    const long long total4 = (long long)B * HW * (C >> 2);
    se_bwd_apply_kernel<<<(int)fmin(32768.0, (double)cdivl(total4, 256)), 256, 0, s>>>(dm, gate, dpool, dx, total4, HW * (C >> 2), C >> 2,
                                                                                         1.0f / (float)HW);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
__global__ void sigmoid_kernel(float* __restrict__ z, long long n) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) z[e] = 1.0f / (1.0f + expf(-z[e]));
}
__global__ void sigmoid_grad_kernel(const float* __restrict__ g, const float* __restrict__ dg, float* __restrict__ dz, long long n) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) dz[e] = dg[e] * g[e] * (1.0f - g[e]);
}
int sigmoid_inplace(float* z, long long n, hipStream_t s) {
    if (!z || n <= 0) return HAB_ERR_ARG;
    sigmoid_kernel<<<(int)fmin(4096.0, (double)cdivl(n, 256)), 256, 0, s>>>(z, n);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
int sigmoid_grad(const float* g, const float* dg, float* dz, long long n, hipStream_t s) {
    if (!g || !dg || !dz || n <= 0) return HAB_ERR_ARG;
    sigmoid_grad_kernel<<<(int)fmin(4096.0, (double)cdivl(n, 256)), 256, 0, s>>>(g, dg, dz, n);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Grouped-conv weight (Cout, Cin/groups, KH, KW) -> dense Wf [co][kh][kw][ci] / Wd [ci][kh][kw][co] with zeros off the block diagonal
__global__ void repack_grouped_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int Cout, int Cin,
                                      int groups, int KH, int KW) {
    const int total = Cout * Cin * KH * KW, cig = Cin / groups, cog = Cout / groups;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int ci = e % Cin;
        int t = e / Cin;
        const int kw = t % KW; t /= KW;
        const int kh = t % KH;
        const int co = t / KH;
        const int g = co / cog;
        const int cl = ci - g * cig;
        const float v = (cl >= 0 && cl < cig) ? w[(((size_t)co * cig + cl) * KH + kh) * KW + kw] : 0.f;
        wf[e] = v;
        if (wd) wd[(((size_t)ci * KH + kh) * KW + kw) * Cout + co] = v;
    }
}
int repack_conv_grouped(const float* w, float* wf, float* wd, int Cout, int Cin, int groups, int KH, int KW, hipStream_t s) {
    if (!w || !wf || groups <= 0 || Cout % groups || Cin % groups) return HAB_ERR_ARG;
    const int total = Cout * Cin * KH * KW;
    repack_grouped_kernel<<<min(2048, cdiv(total, 256)), 256, 0, s>>>(w, wf, wd, Cout, Cin, groups, KH, KW);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
__global__ void gather_grouped_kernel(const float* __restrict__ dense, float* __restrict__ dw, int Cout, int Cin, int groups, int KH, int KW) {
    const int cig = Cin / groups, cog = Cout / groups, total = Cout * cig * KH * KW;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        int t = e / (KH * KW);
        const int r = e % (KH * KW);
        const int cl = t % cig;
        const int co = t / cig;
        const int ci = (co / cog) * cig + cl;
        dw[e] = dense[((size_t)co * Cin + ci) * KH * KW + r];
    }
}
int gather_grouped_wgrad(const float* dense_oihw, float* dw, int Cout, int Cin, int groups, int KH, int KW, hipStream_t s) {
    if (!dense_oihw || !dw || groups <= 0 || Cout % groups || Cin % groups) return HAB_ERR_ARG;
    const int total = Cout * (Cin / groups) * KH * KW;
    gather_grouped_kernel<<<min(2048, cdiv(total, 256)), 256, 0, s>>>(dense_oihw, dw, Cout, Cin, groups, KH, KW);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab

// ------------------------------------------- C ABI -------------------------------------------
using namespace hab;

extern "C" int hab_obs_ingest_pool(const uint8_t* rgb, const float* depth, const int32_t* semantic, const int* rows, float* y, int B, int H,
                                   int W, int cpad, int c_rgb, int c_depth, int c_sem, hipStream_t stream) {
    return ingest_pool(rgb, depth, semantic, rows, y, B, H, W, cpad, c_rgb, c_depth, c_sem, stream);
}
extern "C" int hab_channel_moments(const float* x, int64_t npix, int cpad, int mode, const float* mean, float* out, double* scratch,
                                   int scratch_len, hipStream_t stream) {
    return chan_moment(x, npix, cpad, mode, mean, out, scratch, scratch_len, stream);
}
extern "C" int hab_running_mean_var_update(float* r_mean, float* r_var, float* r_count, const float* b_mean, const float* b_var, float n,
                                           int C, hipStream_t stream) {
    return rmv_update(r_mean, r_var, r_count, b_mean, b_var, n, C, stream);
}
extern "C" int hab_running_mean_var_normalize(float* x, int64_t npix, int cpad, int C, const float* mean, const float* var,
                                              hipStream_t stream) {
    return rmv_normalize(x, npix, cpad, C, mean, var, stream);
}
extern "C" int hab_groupnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, const float* residual, float* mean,
                                 float* rstd, int B, int HW, int C, int groups, int relu, float eps, float* ws, int64_t ws_floats,
                                 hipStream_t stream) {
    GnArgs a;
    a.scratch = ws; a.scratch_floats = ws ? (size_t)ws_floats : 0;
    a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.residual = residual; a.mean = mean; a.rstd = rstd;
    a.B = B; a.HW = HW; a.C = C; a.groups = groups; a.relu = relu; a.eps = eps;
    return groupnorm_forward(a, stream);
}
extern "C" int hab_groupnorm_bwd(const float* x, const float* dy, const float* relu_out, float* dx, float* dy_masked, const float* gamma,
                                 const float* mean, const float* rstd, float* chan_sums, int B, int HW, int C, int groups,
                                 float* ws, int64_t ws_floats, hipStream_t stream) {
    GnBwdArgs a;
    a.scratch = ws; a.scratch_floats = ws ? (size_t)ws_floats : 0;
    a.x = x; a.dy = dy; a.relu_out = relu_out; a.dx = dx; a.dy_masked = dy_masked; a.gamma = gamma; a.mean = mean; a.rstd = rstd;
    a.chan_sums = chan_sums; a.B = B; a.HW = HW; a.C = C; a.groups = groups;
    return groupnorm_backward(a, stream);
}
extern "C" int hab_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, hipStream_t stream) {
    return maxpool_forward(x, y, idx, B, H, W, C, stream);
}
extern "C" int hab_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, hipStream_t stream) {
    return maxpool_backward(dy, idx, dx, B, H, W, C, stream);
}
static EmbedSlot mk_slot(const hab_embed_slot& q) {
    EmbedSlot sl;
    sl.kind = q.kind; sl.in = q.input; sl.w = q.weight; sl.b = q.bias; sl.dw = q.d_weight; sl.db = q.d_bias; sl.ntok = q.num_tokens;
    return sl;
}
extern "C" int hab_nav_embed_fwd(const hab_embed_slot* slots, int nslots, const uint8_t* masks, const int* rows, float* out, int ld,
                                 int col0, int B, float* saved, hipStream_t stream) {
    if (!slots || nslots <= 0 || nslots > EMB_MAX_SLOTS) return HAB_ERR_ARG;
    EmbedArgs a;
    for (int i = 0; i < nslots; ++i) a.slot[i] = mk_slot(slots[i]);
    a.nslots = nslots; a.masks = masks; a.rows = rows; a.out = out; a.ld = ld; a.col0 = col0; a.B = B; a.saved = saved;
    return embed_forward(a, stream);
}
extern "C" int hab_nav_embed_bwd(const hab_embed_slot* slots, int nslots, const float* saved, const float* dout, int ld, int col0, int B,
                                 float* ws, size_t ws_floats, hipStream_t stream) {
    if (!slots || nslots <= 0 || nslots > EMB_MAX_SLOTS) return HAB_ERR_ARG;
    EmbedBwdArgs a;
    for (int i = 0; i < nslots; ++i) a.slot[i] = mk_slot(slots[i]);
    a.nslots = nslots; a.saved = saved; a.dout = dout; a.ld = ld; a.col0 = col0; a.B = B;
    return embed_backward(a, ws, ws_floats, stream);
}
