// conv1x1_gn_stream.h -- 1x1 convolution (bias-free, optionally strided) + GroupNorm (+ residual, + ReLU) in ONE launch for the frames
// conv_gn_slab.h cannot hold in LDS: the 32 x 32-pixel frames of the bottleneck ResNet's layer1 / layer2.0 at rollout batch sizes
// (habitat_baselines/rl/ddppo/policy/resnet.py:75-141 `Bottleneck`: conv1 / conv3 / downsample are 1x1, each followed by GroupNorm --
// 9 of ResNet50's 53 convolutions read a 1024-pixel frame), where the unfused form costs three to four launches per layer (contraction,
// split-K second pass, GroupNorm statistics + apply) of a rollout step that is launch-latency bound.
//
// A 1x1 convolution reads every input pixel exactly once and has no halo, so nothing is shared between waves and nothing needs a barrier:
//   * workgroup = (frame, slab of 32 output channels = whole GroupNorm groups), 8 waves; wave w owns output pixels
//     [32 MT w, 32 MT (w + 1)) as MT tiles of 32 -- the whole frame's outputs of the slab stay in the accumulators
//     (MT x 16 VGPRs per lane), which is what lets the GroupNorm statistics be taken without writing the convolution output;
//   * a tile's activations are read once, coalesced, into a wave-private LDS buffer and come back as MFMA fragments (the lane's own
//     pixel, 8 consecutive input channels), split in registers (exact 3-term bf16 split, bf3_split.h); the next tile's loads are in
//     flight under this tile's MFMAs;
//   * weights: the fragment-ordered bf16 planes conv_gn_slab.h uses (cgs_split_weights: 1 KB contiguous per fragment), one 16-byte
//     load per plane and k-step from L1 / L2, shared by the MT tiles; operands swapped so that a lane ends with 4 consecutive channels
//     of one pixel (16-byte stores);
//   * statistics: exact two-pass per (frame, group) on the accumulators -- per-lane partial sums, xor tree over the 32 pixels of a
//     half-wave, the 8 waves through LDS in a fixed order -- then y = (x - mean) * rstd * gamma + beta with the arithmetic of
//     groupnorm_fwd_reg_kernel, + residual, ReLU.
// Sign schedule as everywhere on the split path (every second workgroup accumulates the negated sum).
#pragma once
#include "bf3_split.h"

namespace hab {

typedef __bf16 c1g_bf16x8 __attribute__((ext_vector_type(8)));

struct C1gArgs {
    const float* x;            // [B][H][W][C] NHWC, C in {32, 64, 128, 256}
    const unsigned short* wq;  // [3][Cout/32][C/16][64 lanes][8] bf16 (cgs_split_weights of the [Cout][C] filter)
    const float* gamma; const float* beta;
    const float* residual;     // [B][Ho*Wo][Cout] or null, added before the ReLU
    float* y;                  // [B][Ho*Wo][Cout]
    float* raw;                // optional: convolution output before the normalisation
    float* mean; float* rstd;  // optional: [B][groups]
    int B, H, W, C, Cout, stride, Ho, Wo, HoWo, KS;
    int groups, gs;            // gs = Cout / groups in {2, 4, 8, 16, 32}
    int nslab;                 // Cout / 32
    int relu, sign_schedule;
    float eps;
};

// MT: tiles of 32 pixels per wave; the C input channels are walked in NCH chunks of CH (C = CH * NCH; CH in {32, 64, 128}).
// A tile's activations (32 pixels x CH floats, 4 .. 16 KB) go through a wave-private LDS buffer: read from memory with every lane of an
// instruction on consecutive 16-byte units (a pixel's channels are contiguous: whole 128-byte lines per instruction -- fragments read
// straight from memory take 32 bytes of every line per instruction and fetch each line four times), written as [pixel][CH + 4] floats,
// read back as MFMA fragments (8 consecutive channels of the lane's pixel: two conflict-free 16-byte reads), split in registers.  The
// loads of the next (tile, chunk) are in flight while the current one is multiplied; no barrier is involved (one wave, in-order LDS).
template <int MT, int CH, int NCH>
__global__ void __launch_bounds__(512) conv1x1_gn_stream_kernel(const C1gArgs a) {
    extern __shared__ __attribute__((aligned(16))) float c1g_sm[];
    __shared__ float red[8][16];
    __shared__ float mu_s[16], rs_s[16];
    constexpr int CP = CH + 4, QPP = CH / 4, NU = QPP / 2, KSC = CH / 16, KS = KSC * NCH;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int frame = blockIdx.x / a.nslab, slab = blockIdx.x - frame * a.nslab;
    const bool flip = a.sign_schedule && (blockIdx.x & 1);
    const unsigned sgn2 = flip ? 0x80008000u : 0u;
    const int gs = a.gs, gsh = 31 - __builtin_clz(gs), ng = 32 >> gsh;  // group size (a power of two), groups inside the slab
    float* xs = c1g_sm + (size_t)wave * 32 * CP;                               // this wave's tile [32][CP]
    int* offs = reinterpret_cast<int*>(c1g_sm + (size_t)8 * 32 * CP) + wave * MT * 32;  // input pixel (in floats / C) of its output pixels

    // this lane's pixels
    bool ok[MT];
    size_t yoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = (wave * MT + m) * 32 + li;
        ok[m] = p < a.HoWo;
        const int pc = ok[m] ? p : a.HoWo - 1;
        const int oh = pc / a.Wo, ow = pc - oh * a.Wo;
        if (hi == 0) offs[m * 32 + li] = (oh * a.stride) * a.W + ow * a.stride;
        yoff[m] = ((size_t)frame * a.HoWo + pc) * a.Cout + slab * 32 + 4 * hi;
    }
    const float* xf = a.x + (size_t)frame * a.H * a.W * a.C;
    const unsigned short* wbase = a.wq + ((size_t)slab * KS * 64 + lane) * 8;
    const size_t wplane = (size_t)a.nslab * KS * 512;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;

    constexpr int PX[6] = {2, 0, 1, 1, 0, 0}, PW_[6] = {0, 2, 1, 0, 1, 0};  // smallest partial product first
    f32x4 nx[NU];
    // unit u = lane + 64 j of a tile chunk: pixel u / QPP, channel quad u % QPP
    auto issue = [&](int m, int kc) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int u = lane + 64 * j;
            nx[j] = *reinterpret_cast<const f32x4*>(xf + (size_t)offs[m * 32 + u / QPP] * a.C + kc * CH + (u % QPP) * 4);
        }
    };
    issue(0, 0);
#pragma unroll
    for (int st = 0; st < MT * NCH; ++st) {
        const int m = st / NCH, kc = st % NCH;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int u = lane + 64 * j;
            *reinterpret_cast<f32x4*>(xs + (u / QPP) * CP + (u % QPP) * 4) = nx[j];
        }
        if (st + 1 < MT * NCH) issue((st + 1) / NCH, (st + 1) % NCH);
        u32x4 wr[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wr[pl] = *reinterpret_cast<const u32x4*>(wbase + pl * wplane + (size_t)(kc * KSC) * 512);
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            c1g_bf16x8 bw[3], af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bw[pl] = __builtin_bit_cast(c1g_bf16x8, u32x4{wr[pl][0] ^ sgn2, wr[pl][1] ^ sgn2, wr[pl][2] ^ sgn2, wr[pl][3] ^ sgn2});
            if (ks + 1 < KSC) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wr[pl] = *reinterpret_cast<const u32x4*>(wbase + pl * wplane + (size_t)(kc * KSC + ks + 1) * 512);
            }
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + li * CP + 16 * ks + 8 * hi);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + li * CP + 16 * ks + 8 * hi + 4);
            u32x4 p0, p1, p2;
            {
                unsigned a1, a2, a3, b1, b2, b3;
                bf3_split2(x0[0], x0[1], a1, a2, a3);
                bf3_split2(x0[2], x0[3], b1, b2, b3);
                p0[0] = a1; p0[1] = b1; p1[0] = a2; p1[1] = b2; p2[0] = a3; p2[1] = b3;
                bf3_split2(x1[0], x1[1], a1, a2, a3);
                bf3_split2(x1[2], x1[3], b1, b2, b3);
                p0[2] = a1; p0[3] = b1; p1[2] = a2; p1[3] = b2; p2[2] = a3; p2[3] = b3;
            }
            af[0] = __builtin_bit_cast(c1g_bf16x8, p0);
            af[1] = __builtin_bit_cast(c1g_bf16x8, p1);
            af[2] = __builtin_bit_cast(c1g_bf16x8, p2);
#pragma unroll
            for (int q = 0; q < 6; ++q) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[PW_[q]], af[PX[q]], acc[m], 0, 0, 0);
        }
    }
    if (flip) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[m][v] = -acc[m][v];
    }
    // lane (pixel li of tile m): channels 8 g + 4 hi .. + 3 of the slab in acc[m][4 g .. 4 g + 3]
    if (a.raw) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (ok[m]) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(a.raw + yoff[m] + 8 * g) = f32x4{acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
            }
    }

    // ---- GroupNorm statistics.  Lane entries: gs >= 4: entry g = the quad's 4 channels (group (8 g + 4 hi) / gs);
    //      gs == 2: entries 2 g, 2 g + 1 = the quad's two channel pairs (groups (8 g + 4 hi) / 2, + 1) ----
    const bool pairs = gs == 2;
    const int nent = pairs ? 8 : 4;
    float sj[8];
    auto wave_total = [&]() {  // fixed xor tree over the 32 pixels of each half-wave; lane li == 0 of a half leaves its entries in LDS
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nent) {
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) sj[j] += __shfl_xor(sj[j], off);
            }
        if (li == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nent) red[wave][hi * 8 + j] = sj[j];
        }
    };
    // group of entry j of half `h`
    auto entry_group = [&](int h, int j) { return pairs ? 4 * (j >> 1) + 2 * h + (j & 1) : (8 * j + 4 * h) >> gsh; };
    auto group_total = [&](int grp) {  // thread `grp`: its group's entries of every wave, in a fixed order
        float tot = 0.f;
        if (pairs) {  // one entry: channels 2 grp, 2 grp + 1 -> quad grp / 4 of half (grp / 2) & 1, pair grp & 1
            const int e = ((grp >> 1) & 1) * 8 + 2 * (grp >> 2) + (grp & 1);
            for (int w = 0; w < 8; ++w) tot += red[w][e];
        } else {      // gs / 4 entries: channel quads grp gs / 4 .. of the slab; quad cq sits in entry cq / 2 of half cq & 1
            const int nq = gs >> 2, q0 = grp * nq;
            for (int w = 0; w < 8; ++w)
                for (int i = 0; i < nq; ++i) tot += red[w][((q0 + i) & 1) * 8 + ((q0 + i) >> 1)];
        }
        return tot;
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sj[j] = 0.f;
        if (j < nent) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float s;
                if (pairs) s = acc[m][4 * (j >> 1) + 2 * (j & 1)] + acc[m][4 * (j >> 1) + 2 * (j & 1) + 1];
                else s = (acc[m][4 * (j & 3)] + acc[m][4 * (j & 3) + 1]) + (acc[m][4 * (j & 3) + 2] + acc[m][4 * (j & 3) + 3]);
                sj[j] += ok[m] ? s : 0.f;
            }
        }
    }
    wave_total();
    __syncthreads();
    const float inv_n = 1.0f / (float)(a.HoWo * gs);
    if (t < ng) mu_s[t] = group_total(t) * inv_n;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sj[j] = 0.f;
        if (j < nent) {
            const float mu = mu_s[entry_group(hi, j)];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float s;
                if (pairs) {
                    const float d0 = acc[m][4 * (j >> 1) + 2 * (j & 1)] - mu, d1 = acc[m][4 * (j >> 1) + 2 * (j & 1) + 1] - mu;
                    s = d0 * d0 + d1 * d1;
                } else {
                    const float d0 = acc[m][4 * (j & 3)] - mu, d1 = acc[m][4 * (j & 3) + 1] - mu, d2 = acc[m][4 * (j & 3) + 2] - mu,
                                d3 = acc[m][4 * (j & 3) + 3] - mu;
                    s = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                sj[j] += ok[m] ? s : 0.f;
            }
        }
    }
    wave_total();
    __syncthreads();
    if (t < ng) {
        const float rs = rsqrtf(group_total(t) * inv_n + a.eps);
        rs_s[t] = rs;
        if (a.mean) {
            a.mean[(size_t)frame * a.groups + slab * ng + t] = mu_s[t];
            a.rstd[(size_t)frame * a.groups + slab * ng + t] = rs;
        }
    }
    __syncthreads();

    // ---- normalise, residual, ReLU, store ----
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = 8 * g + 4 * hi;  // channel of the slab
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + slab * 32 + c0);
        const f32x4 be = *reinterpret_cast<const f32x4*>(a.beta + slab * 32 + c0);
        f32x4 sc, sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int grp = (c0 + k) >> gsh;
            const float mu = mu_s[grp], rs = rs_s[grp];
            sc[k] = rs * ga[k];  // the arithmetic of groupnorm_fwd_reg_kernel
            sh[k] = be[k] - mu * sc[k];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (!ok[m]) continue;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = acc[m][4 * g + k] * sc[k] + sh[k];
            if (a.residual) o += *reinterpret_cast<const f32x4*>(a.residual + yoff[m] + 8 * g);
            if (a.relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
            }
            *reinterpret_cast<f32x4*>(a.y + yoff[m] + 8 * g) = o;
        }
    }
}

inline bool conv1x1_gn_stream_covers(int C, int Cout, int H, int W, int KH, int KW, int stride, int pad, int groups) {
    if (KH != 1 || KW != 1 || pad != 0 || stride < 1 || (C != 32 && C != 64 && C != 128 && C != 256) || Cout <= 0 || (Cout & 31) || groups <= 0 ||
        Cout % groups)
        return false;
    const int gs = Cout / groups;
    if (gs != 2 && gs != 4 && gs != 8 && gs != 16 && gs != 32) return false;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (Ho < 1 || Wo < 1 || Ho * Wo > 1024) return false;
    // one workgroup streams the whole frame: beyond 1024 pixels x 64 channels (512 x 128, 256 x 256) the contraction spread over the
    // chip + the GroupNorm kernels is faster than one CU per (frame, slab) (tools/bench_conv_gn.py: 32.0 vs 29.6 us at 1024 x 128)
    return (long long)Ho * Wo * C <= 65536;
}

// 1: geometry not covered.
inline int conv1x1_gn_stream(C1gArgs a, hipStream_t stream) {
    if (!a.x || !a.wq || !a.gamma || !a.beta || !a.y || a.B <= 0) return HAB_ERR_ARG;
    if ((a.mean == nullptr) != (a.rstd == nullptr)) return HAB_ERR_ARG;
    if (!conv1x1_gn_stream_covers(a.C, a.Cout, a.H, a.W, 1, 1, a.stride, 0, a.groups)) return 1;
    if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.wq) | reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.gamma) |
         reinterpret_cast<uintptr_t>(a.beta) | reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(a.raw)) & 15)
        return 1;
    a.Ho = (a.H - 1) / a.stride + 1; a.Wo = (a.W - 1) / a.stride + 1; a.HoWo = a.Ho * a.Wo;
    a.KS = a.C / 16; a.gs = a.Cout / a.groups; a.nslab = a.Cout / 32;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    a.sign_schedule = sign_schedule;
    if ((long long)a.B * a.nslab > 0x7fffffffLL) return 1;
    const int grid = a.B * a.nslab;
    const int mt = a.HoWo <= 256 ? 1 : (a.HoWo <= 512 ? 2 : 4);
    const int ch = a.C >= 128 ? 128 : a.C;
    const size_t lds = (size_t)8 * 32 * (ch + 4) * sizeof(float) + (size_t)8 * mt * 32 * sizeof(int);  // <= 136 KB (+ 0.7 KB static)
#define C1G_LAUNCH(MT_, CH_, NCH_)                                                                                                   \
    {                                                                                                                                \
        static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_gn_stream_kernel<MT_, CH_, NCH_>), \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);              \
        if (attr_err != hipSuccess) return (int)attr_err;                                                                            \
        conv1x1_gn_stream_kernel<MT_, CH_, NCH_><<<grid, 512, lds, stream>>>(a);                                                     \
    }
#define C1G_C(MT_)                                                                        \
    switch (a.C) {                                                                        \
        case 32: C1G_LAUNCH(MT_, 32, 1) break;                                            \
        case 64: C1G_LAUNCH(MT_, 64, 1) break;                                            \
        case 128: C1G_LAUNCH(MT_, 128, 1) break;                                          \
        default: C1G_LAUNCH(MT_, 128, 2) break;                                           \
    }
    if (mt == 1) { C1G_C(1) } else if (mt == 2) { C1G_C(2) } else { C1G_C(4) }
#undef C1G_C
#undef C1G_LAUNCH
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
