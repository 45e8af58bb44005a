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
__global__ void __launch_bounds__(256) ingest_pool_kernel(const uint8_t* __restrict__ rgb, const float* __restrict__ depth,
                                                          const int* __restrict__ rows, float* __restrict__ y, int B, int H, int W,
                                                          int cpad) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo;
    const float inv255 = (float)(1.0 / 255.0);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int wo = (int)(e % Wo);
        const long long t = e / Wo;
        const int ho = (int)(t % Ho);
        const int f = (int)(t / Ho);
        const size_t srow = rows ? rows[f] : f;
        float out[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) out[c] = 0.f;
        int nc = 0;
        if (rgb) {
            float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
                    const uint8_t* p = rgb + ((srow * H + 2 * ho + dh) * W + 2 * wo + dw) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) s[c] = __fadd_rn(s[c], __fmul_rn((float)p[c], inv255));
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = __fmul_rn(s[c], 0.25f);
            nc = 3;
        }
        if (depth) {
            float s = 0.f;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) s = __fadd_rn(s, depth[(srow * H + 2 * ho + dh) * W + 2 * wo + dw]);
            out[nc] = __fmul_rn(s, 0.25f);
        }
        float* o = y + (size_t)e * cpad;
        for (int c = 0; c < cpad; c += 4) *reinterpret_cast<f32x4*>(o + c) = *reinterpret_cast<const f32x4*>(out + c);
    }
}

int ingest_pool(const uint8_t* rgb, const float* depth, const int* rows, float* y, int B, int H, int W, int cpad, hipStream_t s) {
    if ((!rgb && !depth) || !y || B <= 0 || (H & 1) || (W & 1) || (cpad != 4 && cpad != 8)) return HAB_ERR_ARG;
    const long long total = (long long)B * (H / 2) * (W / 2);
    ingest_pool_kernel<<<(int)fmin(8192.0, (double)cdivl(total, 256)), 256, 0, s>>>(rgb, depth, rows, y, B, H, W, cpad);
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
                                                          const float* __restrict__ mean, double* __restrict__ partial) {
    __shared__ double sm[256];
    const int c = threadIdx.x % cpad, lane = threadIdx.x / cpad, nl = 256 / cpad;
    const float m = (mode == 1) ? mean[c] : 0.f;
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
__global__ void chan_moment_stage2(const double* __restrict__ partial, int nblocks, int cpad, double inv_n, float* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= cpad) return;
    double t = 0.0;
    for (int b = 0; b < nblocks; ++b) t += partial[(size_t)b * cpad + c];
    out[c] = (float)(t * inv_n);
}
int chan_moment(const float* x, long long npix, int cpad, int mode, const float* mean, float* out, double* scratch, int scratch_len,
                hipStream_t s) {
    if (!x || !out || !scratch || npix <= 0 || (256 % cpad) || (mode == 1 && !mean)) return HAB_ERR_ARG;
    int blocks = (int)fmin(1024.0, (double)cdivl(npix, 256 / cpad * 8));
    if (blocks * cpad > scratch_len) blocks = scratch_len / cpad;
    if (blocks < 1) return HAB_ERR_ARG;
    chan_moment_stage1<<<blocks, 256, 0, s>>>(x, npix, cpad, mode, mean, scratch);
    HAB_LAUNCH_CHECK();
    chan_moment_stage2<<<1, 64, 0, s>>>(scratch, blocks, cpad, 1.0 / (double)npix, out);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Chan merge of (running mean, var, count) with the batch moments, all on device (one thread per channel).
// new_count = batch size n (number of frames, as in the reference -- not the number of pixels).
__global__ void rmv_update_kernel(float* __restrict__ r_mean, float* __restrict__ r_var, float* __restrict__ r_count,
                                  const float* __restrict__ b_mean, const float* __restrict__ b_var, float n, int C) {
    const int c = threadIdx.x;
    const float count = r_count[0];
    if (c < C) {
        const float mean = r_mean[c], var = r_var[c], nm = b_mean[c], nv = b_var[c];
        const float m_a = var * count, m_b = nv * n;
        const float d = nm - mean;
        const float M2 = m_a + m_b + d * d * count * n / (count + n);
        r_var[c] = M2 / (count + n);
        r_mean[c] = (count * mean + n * nm) / (count + n);
    }
    __syncthreads();
    if (c == 0) r_count[0] = count + n;
}
int rmv_update(float* r_mean, float* r_var, float* r_count, const float* b_mean, const float* b_var, float n, int C, hipStream_t s) {
    if (!r_mean || !r_var || !r_count || !b_mean || !b_var || C <= 0 || C > 64) return HAB_ERR_ARG;
    rmv_update_kernel<<<1, 64, 0, s>>>(r_mean, r_var, r_count, b_mean, b_var, n, C);
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
        x[e] = a[c] + x[e] * b[c];
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
// GroupNorm (resnet.py:51-57,207-219; eps 1e-5) on NHWC.  One workgroup per frame.
// forward : stats per (frame, group) by per-thread Welford + Chan merge; y = (x-mu)*rstd*gamma + beta
//           [+ residual] [ReLU].  mu / rstd saved for backward.
// backward: dy' = dy * (relu_out > 0); per-channel sums S1 = sum dy', S2 = sum dy'*xhat are written per frame
//           (reduced over frames afterwards -> dbeta, dgamma); dx = rstd*(g - mean_g(g) - xhat*mean_g(g*xhat)),
//           g = dy'*gamma.
// Thread mapping: C4 = C/4 float4 columns; thread t owns column (t % C4), pixels t / C4 + k * (256 / C4).
// ------------------------------------------------------------------------------------------------------
struct WF { float n, mean, m2; };
__device__ inline void wf_add(WF& a, float x) {
    a.n += 1.f;
    const float d = x - a.mean;
    a.mean += d / a.n;
    a.m2 += d * (x - a.mean);
}
__device__ inline WF wf_merge(const WF& a, const WF& b) {
    if (b.n == 0.f) return a;
    if (a.n == 0.f) return b;
    WF r;
    r.n = a.n + b.n;
    const float d = b.mean - a.mean;
    r.mean = a.mean + d * (b.n / r.n);
    r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
    return r;
}

__global__ void __launch_bounds__(256) groupnorm_fwd_kernel(const GnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, C4 = C >> 2, HW = a.HW, G = a.groups, cpg = C / G;
    const int f = blockIdx.x, t = threadIdx.x;
    const int col = t % C4, pl = t / C4, np = 256 / C4;
    const float* x = a.x + (size_t)f * HW * C;
    // per-thread Welford per channel (4 channels)
    WF w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { w[k].n = 0.f; w[k].mean = 0.f; w[k].m2 = 0.f; }
    if (pl < np)
        for (int p = pl; p < HW; p += np) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)p * C + col * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) wf_add(w[k], v[k]);
        }
    // fold the 4 channels of this thread if they belong to one group (cpg >= 4), else keep per channel
    WF* red = reinterpret_cast<WF*>(sm);  // [256][4] worst case
#pragma unroll
    for (int k = 0; k < 4; ++k) red[t * 4 + k] = w[k];
    __syncthreads();
    float* mu_s = sm + 256 * 4 * 3;       // [G]
    float* rs_s = mu_s + G;               // [G]
    if (t < G) {
        WF acc; acc.n = 0.f; acc.mean = 0.f; acc.m2 = 0.f;
        const int c0 = t * cpg;
        for (int c = c0; c < c0 + cpg; ++c) {
            const int cc = c >> 2, k = c & 3;
            for (int q = 0; q < np; ++q) acc = wf_merge(acc, red[(q * C4 + cc) * 4 + k]);
        }
        const float var = acc.m2 / acc.n;
        mu_s[t] = acc.mean;
        rs_s[t] = rsqrtf(var + a.eps);
        a.mean[(size_t)f * G + t] = acc.mean;
        a.rstd[(size_t)f * G + t] = rs_s[t];
    }
    __syncthreads();
    if (pl >= np) return;
    float ga[4], be[4], mu[4], rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = col * 4 + k, g = c / cpg;
        ga[k] = a.gamma[c]; be[k] = a.beta[c]; mu[k] = mu_s[g]; rs[k] = rs_s[g];
    }
    float* y = a.y + (size_t)f * HW * C;
    const float* res = a.residual ? a.residual + (size_t)f * HW * C : nullptr;
    for (int p = pl; p < HW; p += np) {
        const size_t o = (size_t)p * C + col * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (v[k] - mu[k]) * rs[k] * ga[k] + be[k];
        if (res) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(res + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += r[k];
        }
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + o) = v;
    }
}

int groupnorm_forward(const GnArgs& a, hipStream_t s) {
    if (!a.x || !a.y || !a.gamma || !a.beta || !a.mean || !a.rstd || a.B <= 0 || a.C % 4 || a.C > 1024 || a.C % a.groups ||
        (256 % (a.C / 4) && a.C / 4 < 256))
        return HAB_ERR_ARG;
    if (a.C / 4 > 256) return HAB_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(256 * 4 * 3 + 2 * a.groups) * sizeof(float);
    groupnorm_fwd_kernel<<<a.B, 256, lds, s>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

__global__ void __launch_bounds__(256) groupnorm_bwd_kernel(const GnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, C4 = C >> 2, HW = a.HW, G = a.groups, cpg = C / G;
    const int f = blockIdx.x, t = threadIdx.x;
    const int col = t % C4, pl = t / C4, np = 256 / C4;
    const size_t base = (size_t)f * HW * C;
    const float* x = a.x + base;
    const float* dy = a.dy + base;
    const float* ro = a.relu_out ? a.relu_out + base : nullptr;
    float mu[4], rs[4], ga[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = col * 4 + k, g = c / cpg;
        mu[k] = a.mean[(size_t)f * G + g]; rs[k] = a.rstd[(size_t)f * G + g]; ga[k] = a.gamma[c];
    }
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < np)
        for (int p = pl; p < HW; p += np) {
            const size_t o = (size_t)p * C + col * 4;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
            f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
            if (ro) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(ro + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[k] = rv[k] > 0.f ? dv[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { s1[k] += dv[k]; s2[k] += dv[k] * (xv[k] - mu[k]) * rs[k]; }
        }
    float* r1 = sm;             // [256][4]
    float* r2 = sm + 1024;      // [256][4]
    float* c1 = sm + 2048;      // [C]  per-channel S1
    float* c2 = c1 + C;         // [C]
    float* g1 = c2 + C;         // [G]  mean_g(g), mean_g(g*xhat)
    float* g2 = g1 + G;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r1[t * 4 + k] = s1[k]; r2[t * 4 + k] = s2[k]; }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        const int cc = c >> 2, k = c & 3;
        float u = 0.f, v = 0.f;
        for (int q = 0; q < np; ++q) { u += r1[(q * C4 + cc) * 4 + k]; v += r2[(q * C4 + cc) * 4 + k]; }
        c1[c] = u; c2[c] = v;
        a.chan_sums[((size_t)f * 2 + 0) * C + c] = u;   // -> dbeta after the reduction over frames
        a.chan_sums[((size_t)f * 2 + 1) * C + c] = v;   // -> dgamma
    }
    __syncthreads();
    if (t < G) {
        float u = 0.f, v = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) { u += a.gamma[c] * c1[c]; v += a.gamma[c] * c2[c]; }
        const float inv_m = 1.0f / (float)(HW * cpg);
        g1[t] = u * inv_m; g2[t] = v * inv_m;
    }
    __syncthreads();
    if (pl >= np) return;
    float m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int g = (col * 4 + k) / cpg; m1[k] = g1[g]; m2[k] = g2[g]; }
    float* dx = a.dx + base;
    for (int p = pl; p < HW; p += np) {
        const size_t o = (size_t)p * C + col * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
        f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
        if (ro) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(ro + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) dv[k] = rv[k] > 0.f ? dv[k] : 0.f;
        }
        f32x4 out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mu[k]) * rs[k];
            out[k] = rs[k] * (dv[k] * ga[k] - m1[k] - xh * m2[k]);
        }
        *reinterpret_cast<f32x4*>(dx + o) = out;
    }
}

int groupnorm_backward(const GnBwdArgs& a, hipStream_t s) {
    if (!a.x || !a.dy || !a.dx || !a.gamma || !a.mean || !a.rstd || !a.chan_sums || a.B <= 0 || a.C % 4 || a.C % a.groups)
        return HAB_ERR_ARG;
    if (a.C / 4 > 256 || (256 % (a.C / 4))) return HAB_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(2048 + 2 * a.C + 2 * a.groups) * sizeof(float);
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
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long long t = e / C;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int f = (int)(t / Ho);
        float best = -INFINITY;
        int bi = 0;
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * 2 - 1 + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                const float v = x[(((size_t)f * H + h) * W + w) * C + c];
                if (v > best || v != v) { best = v; bi = kh * 3 + kw; }
            }
        }
        y[e] = best;
        idx[e] = (uint8_t)bi;
    }
}
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int B, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * H * W * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long long t = e / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int f = (int)(t / H);
        float s = 0.f;
        // windows (ho, wo) with ho*2-1 <= h <= ho*2+1
        for (int ho = (h) / 2; ho <= (h + 1) / 2; ++ho) {
            if (ho >= Ho) continue;
            const int kh = h - (ho * 2 - 1);
            if (kh < 0 || kh > 2) continue;
            for (int wo = (w) / 2; wo <= (w + 1) / 2; ++wo) {
                if (wo >= Wo) continue;
                const int kw = w - (wo * 2 - 1);
                if (kw < 0 || kw > 2) continue;
                const size_t o = (((size_t)f * Ho + ho) * Wo + wo) * C + c;
                if (idx[o] == kh * 3 + kw) s += dy[o];
            }
        }
        dx[e] = s;
    }
}
int maxpool_forward(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, hipStream_t s) {
    if (!x || !y || !idx || B <= 0) return HAB_ERR_ARG;
    const long long total = (long long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C;
    maxpool_fwd_kernel<<<(int)fmin(16384.0, (double)cdivl(total, 256)), 256, 0, s>>>(x, y, idx, B, H, W, C);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
int maxpool_backward(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, hipStream_t s) {
    if (!dy || !idx || !dx || B <= 0) return HAB_ERR_ARG;
    const long long total = (long long)B * H * W * C;
    maxpool_bwd_kernel<<<(int)fmin(16384.0, (double)cdivl(total, 256)), 256, 0, s>>>(dy, idx, dx, B, H, W, C);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// Goal + previous-action embeddings (resnet_policy.py:662-692,747-753) written into the RNN input:
//   out[f][col0 .. col0+32)    = W_t [rho, cos(-phi), sin(-phi)] + b_t
//   out[f][col0+32 .. col0+64) = E[ mask ? prev_action + 1 : 0 ]
// backward: dW_t, db_t, dE (deterministic: one workgroup per output row, fixed-order frame loop).
// ------------------------------------------------------------------------------------------------------
__global__ void embed_fwd_kernel(const EmbedArgs a) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (f >= a.B) return;
    const int r = a.rows ? a.rows[f] : f;
    float* o = a.out + (size_t)f * a.ld + a.col0;
    if (j < 32) {
        const float rho = a.goal[(size_t)r * 2], phi = a.goal[(size_t)r * 2 + 1];
        const float g0 = rho, g1 = cosf(-phi), g2 = sinf(-phi);
        o[j] = (a.w_t[j * 3] * g0 + a.w_t[j * 3 + 1] * g1) + a.w_t[j * 3 + 2] * g2 + a.b_t[j];
    } else {
        const int jj = j - 32;
        const int tok = a.masks[r] ? (int)a.prev_actions[r] + 1 : 0;
        o[j] = a.emb[(size_t)tok * 32 + jj];
    }
}
int embed_forward(const EmbedArgs& a, hipStream_t s) {
    if (!a.goal || !a.prev_actions || !a.masks || !a.out || a.B <= 0) return HAB_ERR_ARG;
    embed_fwd_kernel<<<cdiv(a.B, 4), 256, 0, s>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// grid = 4 (w_t rows by gate g in 0..2 -> 3 blocks of 32 threads for dW_t[:, g], 1 for db_t) + (A+1) embedding rows
__global__ void __launch_bounds__(64) embed_bwd_kernel(const EmbedBwdArgs a) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (j >= 32) return;
    float s = 0.f;
    if (b < 4) {
        for (int f = 0; f < a.B; ++f) {
            const int r = a.rows ? a.rows[f] : f;
            const float d = a.dout[(size_t)f * a.ld + a.col0 + j];
            float g = 1.f;
            if (b < 3) {
                const float rho = a.goal[(size_t)r * 2], phi = a.goal[(size_t)r * 2 + 1];
                g = b == 0 ? rho : (b == 1 ? cosf(-phi) : sinf(-phi));
            }
            s += d * g;
        }
        if (b < 3) a.dw_t[j * 3 + b] = s; else a.db_t[j] = s;
    } else {
        const int tok = b - 4;
        for (int f = 0; f < a.B; ++f) {
            const int r = a.rows ? a.rows[f] : f;
            const int tk = a.masks[r] ? (int)a.prev_actions[r] + 1 : 0;
            if (tk == tok) s += a.dout[(size_t)f * a.ld + a.col0 + 32 + j];
        }
        a.demb[(size_t)tok * 32 + j] = s;
    }
}
int embed_backward(const EmbedBwdArgs& a, hipStream_t s) {
    if (!a.dout || !a.dw_t || !a.db_t || !a.demb || a.B <= 0) return HAB_ERR_ARG;
    embed_bwd_kernel<<<4 + a.num_tokens, 64, 0, s>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
