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
//   * the INPUT of those frames (<= 64 KB) is read once with whole-row 16-byte loads, split once (exact 3-term split, igemm_bf3.h) and
//     kept in LDS as three bf16 planes [pixel][C + 8] (pitch C/2 + 4 dwords: the 16-byte fragment reads of 16 consecutive pixels hit
//     all 64 banks); a filter tap is a row offset into that image, out-of-image taps read a zero row;
//   * 8 waves = WM pixel-tile groups x WK slices of the reduction (taps x input channels); a wave keeps its MT x NT 32 x 32
//     accumulators and runs its k-steps WITHOUT barriers: activations from LDS, weights from global / L2 in FRAGMENT ORDER -- three
//     bf16 planes split once per optimiser step (hab_split_weight_planes; the 129 steps of a rollout share them), stored so that the
//     64 lanes of an MFMA operand are 1 KB contiguous (the first version read [co][k] rows: 32 cache lines per load instruction, the
//     L1's one-tag-per-cycle rate made 18-32 us kernels out of 3 us of MFMA work); D k-steps of weight loads stay in flight;
//   * the WK partial tiles meet in LDS (over the dead input image) and are summed in wave order (deterministic); GroupNorm: exact
//     two-pass statistics per (frame, group) by wave reductions over the folded tile, then y = (x - mean) * rstd * gamma + beta
//     [+ residual] [ReLU], one 16-byte store per thread.  The pre-normalisation output / mean / rstd are written only when the caller
//     keeps them for a backward pass.
// Sign schedule as everywhere on the split path: every second workgroup accumulates the negated sum.
#pragma once
#include "bf3_split.h"

namespace hab {

typedef __bf16 cgs_bf16x8 __attribute__((ext_vector_type(8)));

struct CgsArgs {
    const float* x;            // [B][H][W][C] NHWC, C % 16 == 0
    const unsigned short* wq;  // fragment-ordered weight planes [3][Cout/32][K/16][64 lanes][8] bf16 (cgs_split_weights)
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
    int sub;                   // 1x1 convolution: the LDS image holds the (strided) pixels the outputs read, row = output pixel
    int prow;                  // pixel rows of the LDS image (the zero row is row `prow`)
    int region0_bytes;         // max(input image, partial tiles)
    int relu, sign_schedule;
    float eps;
    int ablate;                // development (HAB_CGS_ABLATE, tools/bench_conv_gn.py): 1 no input staging, 2 no MFMAs, 4 no weight loads, 8 no statistics
};

template <int MT, int NT, int WM, int WK>
struct CgsCfg {
    static constexpr int Mt = WM * MT * 32, Nt = NT * 32, NQ = Nt / 4, RED_LD = Nt + 4;
    static constexpr int NQUADS = Mt * NQ, QPT = (NQUADS + 511) / 512;
    static constexpr size_t RED_FLOATS = (size_t)WK * Mt * RED_LD, QS_FLOATS = (size_t)NQUADS;
    static_assert(WM * WK == 8, "eight waves");
};

template <int MT, int NT, int WM, int WK, int D>
__global__ void __launch_bounds__(512) conv_gn_slab_kernel(const CgsArgs a) {
    using Cfg = CgsCfg<MT, NT, WM, WK>;
    constexpr int Mt = Cfg::Mt, Nt = Cfg::Nt, NQ = Cfg::NQ, RED_LD = Cfg::RED_LD, QPT = Cfg::QPT;
    extern __shared__ __attribute__((aligned(16))) float cgs_sm[];
    unsigned short* xs = reinterpret_cast<unsigned short*>(cgs_sm);   // [3][prow + 1][C + 8] bf16, then reused as
    float* red = cgs_sm;                                               // [WK][Mt][RED_LD]
    float* qs = reinterpret_cast<float*>(reinterpret_cast<char*>(cgs_sm) + a.region0_bytes);  // [Mt][NQ]
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
    const int PITCH = a.C + 8;
    const int plane = (a.prow + 1) * PITCH;  // bf16 elements per plane of the LDS image

    // k-steps of this wave: an even cut of KS over the WK slices
    const int s_begin = (int)(((long long)a.KS * wk) / WK), s_end = (int)(((long long)a.KS * (wk + 1)) / WK);

    // ---- weight fragments: register ring of D k-steps, first loads in flight before the input image is staged ----
    struct WStage { u32x4 w[NT][3]; };
    WStage st[D];
    const size_t wplane = (size_t)(a.Cout >> 5) * a.KS * 512;
    const unsigned short* wbase = a.wq + (size_t)(slab * NT) * a.KS * 512 + lane * 8;
    auto wload = [&](WStage& g, int s) {
        if (a.ablate & 4) return;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const unsigned short* p = wbase + ((size_t)j * a.KS + s) * 512;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) g.w[j][pl] = *reinterpret_cast<const u32x4*>(p + pl * wplane);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (s_begin + d < s_end) wload(st[d], s_begin + d);

    // ---- epilogue role of this thread: QPT quads (pixel, 4 consecutive channels).  Their residual / gamma / beta loads are issued NOW:
    //      a dependent HBM round trip (~2 us) at the end of a ~10 us kernel otherwise ----
    const int qpg = a.gs >> 2;  // quads per group
    int qcq[QPT], qpair[QPT];
    bool qok[QPT], qin[QPT];
    size_t qout[QPT];
    f32x4 qres[QPT], qga[QPT], qbe[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int q = t + u * 512;
        qin[u] = (Cfg::NQUADS % 512 == 0) || q < Cfg::NQUADS;
        const int pixl = qin[u] ? q / NQ : 0, cq = qin[u] ? q - pixl * NQ : 0;
        const int fl = pixl / a.HoWo, pix = pixl - fl * a.HoWo, frame = fg * a.fpw + fl;
        qok[u] = qin[u] & (fl < a.fpw) & (frame < a.B);
        qcq[u] = cq;
        qpair[u] = min(fl, a.fpw - 1) * gpn + (cq * 4) / a.gs;
        qout[u] = ((size_t)frame * a.HoWo + pix) * a.Cout + co0 + cq * 4;
        qres[u] = (a.residual && qok[u]) ? *reinterpret_cast<const f32x4*>(a.residual + qout[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
        qga[u] = *reinterpret_cast<const f32x4*>(a.gamma + co0 + cq * 4);
        qbe[u] = *reinterpret_cast<const f32x4*>(a.beta + co0 + cq * 4);
    }

    // ---- stage the input image of this workgroup's frames: fp32 NHWC -> three bf16 planes in LDS ----
    {
        const int C4 = a.C >> 2, HW = a.H * a.W;
        const int rows_per_frame = a.sub ? a.HoWo : HW;
        const int units = a.prow * C4;
        constexpr int UB = 4;  // loads in flight per thread
        for (int u0 = t; u0 < ((a.ablate & 1) ? 0 : units); u0 += 512 * UB) {
            f32x4 v[UB];
            int dsto[UB];
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                const int u = u0 + j * 512;
                dsto[j] = -1;
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (u < units) {
                    const int row = u / C4, c4 = u - row * C4;
                    const int fl = row / rows_per_frame, rem = row - fl * rows_per_frame;
                    const int frame = fg * a.fpw + fl;
                    dsto[j] = row * PITCH + c4 * 4;
                    if (frame < a.B) {
                        int pin = rem;
                        if (a.sub) { const int ho = rem / a.Wo, wo = rem - ho * a.Wo; pin = ho * a.stride * a.W + wo * a.stride; }
                        v[j] = *reinterpret_cast<const f32x4*>(a.x + ((size_t)frame * HW + pin) * a.C + c4 * 4);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UB; ++j)
                if (dsto[j] >= 0) {
                    unsigned short* dst = xs + dsto[j];
                    unsigned a1, a2, a3, b1, b2, b3;
                    bf3_split2(v[j][0], v[j][1], a1, a2, a3);
                    bf3_split2(v[j][2], v[j][3], b1, b2, b3);
                    *reinterpret_cast<u32x2*>(dst) = u32x2{a1, b1};
                    *reinterpret_cast<u32x2*>(dst + plane) = u32x2{a2, b2};
                    *reinterpret_cast<u32x2*>(dst + 2 * plane) = u32x2{a3, b3};
                }
        }
        for (int e = t; e < 3 * PITCH; e += 512) {  // the zero row of every plane
            const int pl = e / PITCH;
            xs[(size_t)pl * plane + a.prow * PITCH + (e - pl * PITCH)] = 0;
        }
    }

    // ---- this lane's pixel rows (operand B of the swapped MFMA: n = pixel) ----
    int rbase[MT], h0[MT], w0[MT];
    bool rv[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pl = (wm * MT + i) * 32 + li;
        const int fl = pl / a.HoWo, pix = pl - fl * a.HoWo;
        rv[i] = (fl < a.fpw) & (fg * a.fpw + fl < a.B);
        const int ho = pix / a.Wo, wo = pix - ho * a.Wo;
        h0[i] = ho * a.stride - a.pad;
        w0[i] = wo * a.stride - a.pad;
        rbase[i] = a.sub ? pl : fl * a.H * a.W;
    }
    // position of the NEXT activation fragment to be read: tap (kh, kw) and first channel
    int l_c, l_kh, l_kw;
    {
        const int kk = s_begin * 16, tap = kk / a.C;
        l_c = kk - tap * a.C;
        l_kh = tap / a.KW;
        l_kw = tap - l_kh * a.KW;
    }
    struct AFrag { cgs_bf16x8 p[MT][3]; };
    auto aread = [&](AFrag& f) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int r;
            if (a.sub) {
                r = rv[i] ? rbase[i] : a.prow;
            } else {
                const int hin = h0[i] + l_kh, win = w0[i] + l_kw;
                const bool ok = rv[i] & ((unsigned)hin < (unsigned)a.H) & ((unsigned)win < (unsigned)a.W);
                r = ok ? rbase[i] + hin * a.W + win : a.prow;
            }
            const unsigned short* src = xs + r * PITCH + l_c + 8 * hi;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f.p[i][pl] = *reinterpret_cast<const cgs_bf16x8*>(src + pl * plane);
        }
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

    auto compute = [&](const AFrag& f, const WStage& g) {
        if (a.ablate & 2) return;
        cgs_bf16x8 bw[NT][3];
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][PW[q]], f.p[i][PX[q]], acc[i][j], 0, 0, 0);
    };

    __syncthreads();  // the input image is complete

    // ---- main loop: no barriers; the next step's activation fragments are read before this step's MFMAs ----
    AFrag cur, nxt;
    if (s_begin < s_end) aread(cur);
    for (int s = s_begin; s < s_end; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (s + d < s_end) {
                if (s + d + 1 < s_end) aread(nxt);
                compute(cur, st[d]);
                if (s + d + D < s_end) wload(st[d], s + d + D);
                cur = nxt;
            }
        }
    }
    __syncthreads();  // every wave is done with the input image: the partial tiles go over it

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
    f32x4 v[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int q = t + u * 512;
        const int pixl = qin[u] ? q / NQ : 0, cq = qcq[u];
        f32x4 s = *reinterpret_cast<const f32x4*>(red + (size_t)pixl * RED_LD + cq * 4);
#pragma unroll
        for (int w = 1; w < WK; ++w) s += *reinterpret_cast<const f32x4*>(red + ((size_t)w * Mt + pixl) * RED_LD + cq * 4);
        if (flip) s = -s;
        v[u] = s;
        if (qin[u]) qs[q] = qok[u] ? (s[0] + s[1]) + (s[2] + s[3]) : 0.f;
        if (a.raw && qok[u]) *reinterpret_cast<f32x4*>(a.raw + qout[u]) = s;
    }
    __syncthreads();
    const int nmem = (a.ablate & 8) ? 0 : a.HoWo * qpg;
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
        const float mu = mu_s[qpair[u]];
        const f32x4 d = v[u] - f32x4{mu, mu, mu, mu};
        if (qin[u]) qs[q] = qok[u] ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f;
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
        const float mu = mu_s[qpair[u]], rs = rs_s[qpair[u]];
        const f32x4 ga = qga[u], be = qbe[u];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = rs * ga[k], sh = be[k] - mu * sc;  // the arithmetic of groupnorm_fwd_reg_kernel
            o[k] = v[u][k] * sc + sh;
        }
        if (a.residual) o += qres[u];
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.y + qout[u]) = o;
    }
}

// Forward-packed weight wf [Cout][K] (K = KH*KW*C in (kh, kw, ci) order, Cout % 32 == 0, K % 16 == 0) -> the three bf16 planes of its
// exact split in MFMA FRAGMENT order: plane p, channel tile ct = co / 32, k-step s = k / 16 occupy 1 KB
//   out[((p * Cout/32 + ct) * K/16 + s) * 512 + lane * 8 + e] = plane_p(wf[ct * 32 + (lane & 31)][s * 16 + 8 * (lane >> 5) + e])
// so that the 64 lanes of a wave read one operand fragment as 1 KB of contiguous memory.
__global__ void cgs_split_weights_kernel(const float* __restrict__ wf, unsigned* __restrict__ out, int Cout, int K) {
    const int KS = K >> 4;
    const long long npairs = (long long)Cout * K / 2, plane_pairs = npairs;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (long long)gridDim.x * blockDim.x) {
        const int ep = (int)(q & 3), lane = (int)((q >> 2) & 63);
        const long long blk = q >> 8;  // (ct, s)
        const int s = (int)(blk % KS), ct = (int)(blk / KS);
        const int co = ct * 32 + (lane & 31), k = s * 16 + 8 * (lane >> 5) + 2 * ep;
        unsigned p0, p1, p2;
        bf3_split2(wf[(size_t)co * K + k], wf[(size_t)co * K + k + 1], p0, p1, p2);
        out[q] = p0; out[plane_pairs + q] = p1; out[2 * plane_pairs + q] = p2;
    }
}

inline int cgs_split_weights(const float* wf, int Cout, int K, unsigned short* planes, hipStream_t stream) {
    if (!wf || !planes || Cout <= 0 || K <= 0 || (Cout & 31) || (K & 15)) return HAB_ERR_ARG;
    const long long np = (long long)Cout * K / 2;
    cgs_split_weights_kernel<<<(int)(np + 255 < 256LL * 1024 ? (np + 255) / 256 : 1024), 256, 0, stream>>>(wf, reinterpret_cast<unsigned*>(planes), Cout, K);
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
constexpr int CGS_LDS_MAX = 160 * 1024;
// LDS bytes of a launch: (input image | partial tiles) + quad sums + statistics; 0 when the input image of one workgroup does not fit
template <int MT, int NT, int WM, int WK>
inline size_t cgs_lds(const CgsArgs& a, int* region0, int* prow_out) {
    using Cfg = CgsCfg<MT, NT, WM, WK>;
    const int fpw = Cfg::Mt / a.HoWo;
    const bool sub = a.KH == 1 && a.KW == 1;
    const long long prow = (long long)fpw * (sub ? a.HoWo : a.H * a.W);
    const long long image = (prow + 1) * (a.C + 8) * 2 * 3;
    long long r0 = (long long)(Cfg::RED_FLOATS * sizeof(float));
    if (image > r0) r0 = image;
    r0 = (r0 + 15) & ~15LL;
    const long long total = r0 + (long long)(Cfg::QS_FLOATS + 2 * (size_t)fpw * (Cfg::Nt / a.gs)) * (long long)sizeof(float);
    if (total > CGS_LDS_MAX) return 0;
    *region0 = (int)r0;
    *prow_out = (int)prow;
    return (size_t)total;
}

template <int MT, int NT, int WM, int WK, int D>
inline int cgs_launch(CgsArgs& a, hipStream_t stream) {
    using Cfg = CgsCfg<MT, NT, WM, WK>;
    a.fpw = Cfg::Mt / a.HoWo;
    a.nslab = a.Cout / Cfg::Nt;
    a.sub = (a.KH == 1 && a.KW == 1) ? 1 : 0;
    const size_t lds = cgs_lds<MT, NT, WM, WK>(a, &a.region0_bytes, &a.prow);
    if (lds == 0) return 1;
    auto kern = conv_gn_slab_kernel<MT, NT, WM, WK, D>;
    // set once per process and instantiation (thread-safe static initialisation: engines of several inference workers call this concurrently)
    static const hipError_t attr_err =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CGS_LDS_MAX);
    if (attr_err != hipSuccess) return (int)attr_err;
    const int nfg = (a.B + a.fpw - 1) / a.fpw;
    kern<<<nfg * a.nslab, 512, lds, stream>>>(a);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

inline int cgs_prepare(CgsArgs& a) {
    a.Ho = (a.H + 2 * a.pad - a.KH) / a.stride + 1;
    a.Wo = (a.W + 2 * a.pad - a.KW) / a.stride + 1;
    a.HoWo = a.Ho * a.Wo;
    if (a.Ho <= 0 || a.Wo <= 0 || !conv_gn_slab_shape(a.C, a.Cout, a.HoWo, a.groups)) return 1;
    if (a.KH == 1 && a.KW == 1 && a.pad != 0) return 1;
    a.K = a.KH * a.KW * a.C;
    a.KS = a.K / 16;
    a.gs = a.Cout / a.groups;
    return HAB_OK;
}

// does the geometry run on the fused kernel (shape covered AND the input image of a workgroup fits LDS)?
inline bool conv_gn_slab_covers(int C, int Cout, int H, int W, int KH, int KW, int stride, int pad, int groups) {
    CgsArgs a{};
    a.H = H; a.W = W; a.C = C; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.groups = groups;
    if (cgs_prepare(a) != HAB_OK) return false;
    int r0, pr;
    if (a.gs > 64) return cgs_lds<1, 4, 1, 8>(a, &r0, &pr) != 0;
    if (a.gs > 32) return cgs_lds<1, 2, 1, 8>(a, &r0, &pr) != 0;
    if (a.HoWo <= 32) return cgs_lds<1, 1, 1, 8>(a, &r0, &pr) != 0;
    if (a.HoWo <= 64) return cgs_lds<2, 1, 1, 8>(a, &r0, &pr) != 0;
    if (a.HoWo <= 128) return cgs_lds<2, 1, 2, 4>(a, &r0, &pr) != 0;
    return cgs_lds<2, 1, 4, 2>(a, &r0, &pr) != 0;
}

// 1: geometry not covered (the caller runs the unfused path).
inline int conv_gn_slab(CgsArgs a, hipStream_t stream) {
    if (!a.x || !a.wq || !a.gamma || !a.beta || !a.y || a.B <= 0) return HAB_ERR_ARG;
    if ((a.mean == nullptr) != (a.rstd == nullptr)) return HAB_ERR_ARG;
    const int rc = cgs_prepare(a);
    if (rc != HAB_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.wq) | reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.gamma) |
         reinterpret_cast<uintptr_t>(a.beta) | reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(a.raw)) & 15)
        return 1;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    static const int ablate = hab_env_int("HAB_CGS_ABLATE", 0);
    a.ablate = ablate;
    if (a.gs > 64) return cgs_launch<1, 4, 1, 8, 2>(a, stream);
    if (a.gs > 32) return cgs_launch<1, 2, 1, 8, 3>(a, stream);
    if (a.HoWo <= 32) return cgs_launch<1, 1, 1, 8, 4>(a, stream);
    if (a.HoWo <= 64) return cgs_launch<2, 1, 1, 8, 4>(a, stream);
    if (a.HoWo <= 128) return cgs_launch<2, 1, 2, 4, 4>(a, stream);
    return cgs_launch<2, 1, 4, 2, 4>(a, stream);
}

}  // namespace hab
