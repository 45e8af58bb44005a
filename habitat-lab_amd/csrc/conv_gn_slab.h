// conv_gn_slab.h -- convolution + GroupNorm (+ residual, + ReLU) of a GroupNorm-ResNet block in ONE launch, for the small-batch passes
// (the 64-frame `act` of every rollout step, hab_policy_encode, small evaluate minibatches).
//
// Replaces, per layer: nn.Conv2d(bias=False) -> nn.GroupNorm -> [+ identity] -> [ReLU]
//   (habitat_baselines/rl/ddppo/policy/resnet.py:19-34 conv3x3 / conv1x1, :51-69 BasicBlock.forward, :129-152 Bottleneck, :207-219
//   downsample; resnet_policy.py:213-234 the compression layer) and what PPOTrainer._compute_actions_and_step_envs runs through it
//   once per environment step (rl/ppo/ppo_trainer.py:343-399).
//
// Why (profiles/r03_c3_phases.txt): a rollout step of the ResNet18 policy was 74 launches for 64 frames -- every convolution as an
// im2col contraction split over K to fill the chip, a split-K second pass, then a GroupNorm launch: 20 + 5 + 8 us per layer for
// 0.6 GFLOP.  The layers behind layer1 are small per frame (256 / 64 / 16 pixels) and GroupNorm statistics are per (frame, group of
// Cout / G consecutive channels), so a workgroup that owns ALL pixels of its frames for a SLAB of output channels made of whole groups
// can finish the layer alone:
//   * workgroup = (frame group, channel slab of NT x 32 channels); Mt = WM x MT x 32 pixel rows = fpw whole frames;
//   * 8 waves = WM pixel-tile groups x WK slices of the reduction (taps x input channels); a wave keeps its MT x NT 32 x 32 accumulators;
//   * NO LDS and NO barrier in the main loop: a wave reads its MFMA fragments straight from global / L2 in fragment order --
//     weights as three bf16 planes split ONCE per optimiser step (hab_split_weight_planes: the 129 steps of a rollout share them),
//     16 bytes per plane and lane; activations as 32 contiguous bytes of one NHWC pixel per lane (8 channels of a tap), split in
//     registers (exact 3-term split, igemm_bf3.h); D k-steps of loads are kept in flight in a register ring;
//   * the WK partial tiles meet in LDS and are summed in wave order (deterministic); GroupNorm: exact two-pass statistics per
//     (frame, group) by wave reductions over the folded tile, then y = (x - mean) * rstd * gamma + beta [+ residual] [ReLU] with one
//     16-byte store per thread.  The pre-normalisation output / mean / rstd are written only when the caller keeps them for a backward.
// Sign schedule as everywhere on the split path: every second workgroup accumulates the negated sum.
#pragma once
#include "bf3_split.h"

namespace hab {

typedef __bf16 cgs_bf16x8 __attribute__((ext_vector_type(8)));

struct CgsArgs {
    const float* x;            // [B][H][W][C] NHWC, C % 16 == 0
    const unsigned short* wp;  // [3 planes][Cout][K] bf16, K = KH*KW*C in (kh, kw, ci) order (the forward packing, split)
    const float* gamma; const float* beta;
    const float* residual;     // [B][Ho*Wo][Cout] or null, added before the ReLU
    float* y;                  // [B][Ho*Wo][Cout]
    float* raw;                // optional: convolution output before the normalisation (GroupNorm input, kept for backward)
    float* mean; float* rstd;  // optional: [B][groups]
    int B, H, W, C, Cout, KH, KW, stride, pad, Ho, Wo, HoWo;
    int K, KS;                 // reduction length, k-steps of 16
    int groups, gs;            // GroupNorm groups, channels per group
    int fpw;                   // frames per workgroup
    int nslab;                 // Cout / (NT * 32)
    int relu, sign_schedule;
    float eps;
};

template <int MT, int NT, int WM, int WK>
struct CgsCfg {
    static constexpr int Mt = WM * MT * 32, Nt = NT * 32, NQ = Nt / 4, RED_LD = Nt + 4;
    static constexpr int NQUADS = Mt * NQ, QPT = (NQUADS + 511) / 512;
    static constexpr size_t RED_FLOATS = (size_t)WK * Mt * RED_LD, QS_FLOATS = (size_t)NQUADS;
    __host__ __device__ static constexpr size_t lds_bytes(int npairs) { return (RED_FLOATS + QS_FLOATS + 2 * (size_t)npairs) * sizeof(float); }
    static_assert(WM * WK == 8, "eight waves");
};

template <int MT, int NT, int WM, int WK, int D>
__global__ void __launch_bounds__(512) conv_gn_slab_kernel(const CgsArgs a) {
    using Cfg = CgsCfg<MT, NT, WM, WK>;
    constexpr int Mt = Cfg::Mt, Nt = Cfg::Nt, NQ = Cfg::NQ, RED_LD = Cfg::RED_LD, QPT = Cfg::QPT;
    extern __shared__ __attribute__((aligned(16))) float cgs_sm[];
    float* red = cgs_sm;                         // [WK][Mt][RED_LD]
    float* qs = red + Cfg::RED_FLOATS;           // [Mt][NQ]
    float* mu_s = qs + Cfg::QS_FLOATS;           // [npairs]
    const int gpn = Nt / a.gs;                   // groups per slab
    const int npairs = a.fpw * gpn;
    float* rs_s = mu_s + npairs;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wk = wave % WK, wm = wave / WK;
    const int li = lane & 31, hi = lane >> 5;
    const int slab = blockIdx.x % a.nslab, fg = blockIdx.x / a.nslab;
    const int co0 = slab * Nt;
    const bool flip = a.sign_schedule && ((slab + fg) & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;

    // ---- this lane's pixel rows (operand B of the swapped MFMA: n = pixel) ----
    const float* xb[MT];
    int h0[MT], w0[MT];
    bool rv[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pl = (wm * MT + i) * 32 + li;
        const int fl = pl / a.HoWo, pix = pl - fl * a.HoWo;
        const int frame = fg * a.fpw + fl;
        rv[i] = (fl < a.fpw) & (frame < a.B);
        const int ho = pix / a.Wo, wo = pix - ho * a.Wo;
        h0[i] = ho * a.stride - a.pad;
        w0[i] = wo * a.stride - a.pad;
        xb[i] = a.x + (size_t)(rv[i] ? frame : 0) * a.H * a.W * a.C + 8 * hi;
    }
    // ---- this lane's weight rows (operand A: m = output channel), 8 reduction elements per k-step and plane ----
    const unsigned short* wrow[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wrow[j] = a.wp + (size_t)(co0 + j * 32 + li) * a.K + 8 * hi;
    const size_t wplane = (size_t)a.Cout * a.K;

    // k-steps of this wave: an even cut of KS over the WK slices
    const int s_begin = (int)(((long long)a.KS * wk) / WK), s_end = (int)(((long long)a.KS * (wk + 1)) / WK);

    struct Stage {
        u32x4 w[NT][3];
        f32x4 x[MT][2];
        unsigned ok;
    };
    Stage st[D];
    // position of the NEXT k-step to be loaded: tap (kh, kw) and first channel
    int l_s = s_begin, l_c, l_kh, l_kw;
    {
        const int kk = s_begin * 16, tap = kk / a.C;
        l_c = kk - tap * a.C;
        l_kh = tap / a.KW;
        l_kw = tap - l_kh * a.KW;
    }
    auto load = [&](Stage& g) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const unsigned short* p = wrow[j] + (size_t)l_s * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) g.w[j][pl] = *reinterpret_cast<const u32x4*>(p + pl * wplane);
        }
        g.ok = 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int hin = h0[i] + l_kh, win = w0[i] + l_kw;
            const bool ok = rv[i] & ((unsigned)hin < (unsigned)a.H) & ((unsigned)win < (unsigned)a.W);
            const float* p = xb[i] + (ok ? ((size_t)hin * a.W + win) * a.C + l_c : 0);
            g.x[i][0] = *reinterpret_cast<const f32x4*>(p);
            g.x[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
            g.ok |= ok ? (1u << i) : 0u;
        }
        ++l_s;
        l_c += 16;
        if (l_c >= a.C) {
            l_c = 0;
            if (++l_kw == a.KW) { l_kw = 0; ++l_kh; }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    auto compute = [&](const Stage& g) {
        cgs_bf16x8 af[MT][3], bw[NT][3];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bool ok = (g.ok >> i) & 1u;
            f32x4 v0 = g.x[i][0], v1 = g.x[i][1];
            if (!ok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
            unsigned p[3][4];
            bf3_split2(v0[0], v0[1], p[0][0], p[1][0], p[2][0]);
            bf3_split2(v0[2], v0[3], p[0][1], p[1][1], p[2][1]);
            bf3_split2(v1[0], v1[1], p[0][2], p[1][2], p[2][2]);
            bf3_split2(v1[2], v1[3], p[0][3], p[1][3], p[2][3]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[i][pl] = __builtin_bit_cast(cgs_bf16x8, u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x4 w = g.w[j][pl];
                bw[j][pl] = __builtin_bit_cast(cgs_bf16x8, u32x4{w[0] ^ sgn2, w[1] ^ sgn2, w[2] ^ sgn2, w[3] ^ sgn2});
            }
        constexpr int PX[6] = {2, 0, 1, 1, 0, 0}, PW[6] = {0, 2, 1, 0, 1, 0};  // smallest partial product first
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)  // operands swapped: D[m = output channel][n = pixel]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][PW[q]], af[i][PX[q]], acc[i][j], 0, 0, 0);
    };

    // ---- main loop: register ring of D k-steps ----
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (s_begin + d < s_end) load(st[d]);
    for (int s = s_begin; s < s_end; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (s + d < s_end) {
                compute(st[d]);
                if (s + d + D < s_end) load(st[d]);
            }
        }
    }

    // ---- the WK partial tiles -> LDS.  lane (pixel li): channels j*32 + 8 g + 4 hi .. +3 in acc[i][j][4 g .. 4 g + 3] ----
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float* mine = red + ((size_t)wk * Mt + (wm * MT + i) * 32 + li) * RED_LD;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(mine + j * 32 + 8 * g + 4 * hi) =
                    f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
    }
    __syncthreads();

    // ---- fold in wave order; thread owns QPT quads (pixel, 4 consecutive channels) ----
    const int qpg = a.gs >> 2;  // quads per group
    f32x4 v[QPT];
    int qpix[QPT], qcq[QPT], qpair[QPT];
    bool qok[QPT];
    size_t qout[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int q = t + u * 512;
        const bool in = (Cfg::NQUADS % 512 == 0) || q < Cfg::NQUADS;
        const int pixl = in ? q / NQ : 0, cq = in ? q - pixl * NQ : 0;
        f32x4 s = *reinterpret_cast<const f32x4*>(red + (size_t)pixl * RED_LD + cq * 4);
#pragma unroll
        for (int w = 1; w < WK; ++w) s += *reinterpret_cast<const f32x4*>(red + ((size_t)w * Mt + pixl) * RED_LD + cq * 4);
        if (flip) s = -s;
        const int fl = pixl / a.HoWo, pix = pixl - fl * a.HoWo, frame = fg * a.fpw + fl;
        qok[u] = in & (fl < a.fpw) & (frame < a.B);
        v[u] = s;
        qpix[u] = pixl; qcq[u] = cq;
        qpair[u] = min(fl, a.fpw - 1) * gpn + (cq * 4) / a.gs;
        qout[u] = ((size_t)frame * a.HoWo + pix) * a.Cout + co0 + cq * 4;
        if (in) qs[q] = qok[u] ? (s[0] + s[1]) + (s[2] + s[3]) : 0.f;
        if (a.raw && qok[u]) *reinterpret_cast<f32x4*>(a.raw + qout[u]) = s;
    }
    __syncthreads();
    const int nmem = a.HoWo * qpg;
    const float inv_n = 1.0f / (float)(a.HoWo * a.gs);
    // mean per (frame, group): wave w takes pairs w, w + 8, ...
    for (int pr = wave; pr < npairs; pr += 8) {
        const int fl = pr / gpn, gl = pr - fl * gpn;
        float sacc = 0.f;
        for (int m = lane; m < nmem; m += 64) {
            const int pp = m / qpg, cc = m - pp * qpg;
            sacc += qs[(size_t)(fl * a.HoWo + pp) * NQ + gl * qpg + cc];
        }
        sacc = wave_sum(sacc);
        if (lane == 0) mu_s[pr] = sacc * inv_n;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int q = t + u * 512;
        const bool in = (Cfg::NQUADS % 512 == 0) || q < Cfg::NQUADS;
        const float mu = mu_s[qpair[u]];
        const f32x4 d = v[u] - f32x4{mu, mu, mu, mu};
        if (in) qs[q] = qok[u] ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f;
    }
    __syncthreads();
    for (int pr = wave; pr < npairs; pr += 8) {
        const int fl = pr / gpn, gl = pr - fl * gpn;
        float sacc = 0.f;
        for (int m = lane; m < nmem; m += 64) {
            const int pp = m / qpg, cc = m - pp * qpg;
            sacc += qs[(size_t)(fl * a.HoWo + pp) * NQ + gl * qpg + cc];
        }
        sacc = wave_sum(sacc);
        if (lane == 0) {
            const float rs = rsqrtf(sacc * inv_n + a.eps);
            rs_s[pr] = rs;
            const int frame = fg * a.fpw + fl;
            if (a.mean && frame < a.B) {
                a.mean[(size_t)frame * a.groups + slab * gpn + gl] = mu_s[pr];
                a.rstd[(size_t)frame * a.groups + slab * gpn + gl] = rs;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        if (!qok[u]) continue;
        const int c = co0 + qcq[u] * 4;
        const float mu = mu_s[qpair[u]], rs = rs_s[qpair[u]];
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + c), be = *reinterpret_cast<const f32x4*>(a.beta + c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = rs * ga[k], sh = be[k] - mu * sc;  // the arithmetic of groupnorm_fwd_reg_kernel
            o[k] = v[u][k] * sc + sh;
        }
        if (a.residual) o += *reinterpret_cast<const f32x4*>(a.residual + qout[u]);
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.y + qout[u]) = o;
    }
    (void)qpix;
}

// fp32 [n] (n % 2 == 0) -> three bf16 planes [3][n]: plane p at out + p * n
__global__ void cgs_split_planes_kernel(const float* __restrict__ w, unsigned* __restrict__ out, long long npairs) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < npairs; e += (long long)gridDim.x * blockDim.x) {
        unsigned p0, p1, p2;
        bf3_split2(w[2 * e], w[2 * e + 1], p0, p1, p2);
        out[e] = p0; out[npairs + e] = p1; out[2 * npairs + e] = p2;
    }
}

inline int split_weight_planes(const float* w, long long n, unsigned short* planes, hipStream_t stream) {
    if (!w || !planes || n <= 0 || (n & 1)) return HAB_ERR_ARG;
    const long long np = n >> 1;
    cgs_split_planes_kernel<<<(int)(np + 255 < 256LL * 1024 ? (np + 255) / 256 : 1024), 256, 0, stream>>>(w, reinterpret_cast<unsigned*>(planes), np);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Geometry covered by the fused kernel (the engine reserves the weight planes for exactly these layers).
inline bool conv_gn_slab_shape(int C, int Cout, int HoWo, int groups) {
    if (C <= 0 || (C & 15) || Cout <= 0 || (Cout & 31) || groups <= 0 || Cout % groups) return false;
    const int gs = Cout / groups;
    if ((gs & 3) || gs > 128) return false;
    const int nt = gs <= 32 ? 32 : (gs <= 64 ? 64 : 128);
    if (nt % gs || Cout % nt) return false;
    if (HoWo <= 0 || HoWo > 256) return false;
    if (nt > 32 && HoWo > 32) return false;  // the wide slabs exist for the 32-row tile only
    return true;
}

template <int MT, int NT, int WM, int WK, int D>
inline int cgs_launch(CgsArgs& a, hipStream_t stream) {
    using Cfg = CgsCfg<MT, NT, WM, WK>;
    a.fpw = Cfg::Mt / a.HoWo;
    a.nslab = a.Cout / Cfg::Nt;
    const int npairs = a.fpw * (Cfg::Nt / a.gs);
    const size_t lds = Cfg::lds_bytes(npairs);
    if (lds > 160 * 1024) return 1;
    auto kern = conv_gn_slab_kernel<MT, NT, WM, WK, D>;
    // set once per process and instantiation (thread-safe static initialisation: engines of several inference workers call this concurrently)
    static const hipError_t attr_err =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_err != hipSuccess) return (int)attr_err;
    const int nfg = (a.B + a.fpw - 1) / a.fpw;
    kern<<<nfg * a.nslab, 512, lds, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// 1: geometry not covered (the caller runs the unfused path).
inline int conv_gn_slab(CgsArgs a, hipStream_t stream) {
    if (!a.x || !a.wp || !a.gamma || !a.beta || !a.y || a.B <= 0) return HAB_ERR_ARG;
    if ((a.mean == nullptr) != (a.rstd == nullptr)) return HAB_ERR_ARG;
    a.Ho = (a.H + 2 * a.pad - a.KH) / a.stride + 1;
    a.Wo = (a.W + 2 * a.pad - a.KW) / a.stride + 1;
    a.HoWo = a.Ho * a.Wo;
    if (a.Ho <= 0 || a.Wo <= 0 || !conv_gn_slab_shape(a.C, a.Cout, a.HoWo, a.groups)) return 1;
    if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.wp) | reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.gamma) |
         reinterpret_cast<uintptr_t>(a.beta) | reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(a.raw)) & 15)
        return 1;
    a.K = a.KH * a.KW * a.C;
    a.KS = a.K / 16;
    a.gs = a.Cout / a.groups;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    if (a.gs > 64) return cgs_launch<1, 4, 1, 8, 2>(a, stream);
    if (a.gs > 32) return cgs_launch<1, 2, 1, 8, 3>(a, stream);
    if (a.HoWo <= 32) return cgs_launch<1, 1, 1, 8, 4>(a, stream);
    if (a.HoWo <= 64) return cgs_launch<2, 1, 1, 8, 4>(a, stream);
    if (a.HoWo <= 128) return cgs_launch<2, 1, 2, 4, 4>(a, stream);
    return cgs_launch<2, 1, 4, 2, 4>(a, stream);
}

}  // namespace hab
