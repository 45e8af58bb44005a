// wgrad3x3_patch.h -- weight gradient of a 3x3 / stride 1 / pad 1 convolution (the bulk of the GroupNorm-ResNet encoder,
// resnet.py:15-28) with the input patch RESIDENT in LDS (fp32 MFMA, gfx950).
//
//   dW[kh][kw][ci][co] = sum_{img,y,x} X[img, y+kh-1, x+kw-1, ci] * dY[img, y, x, co]
//
// The implicit-GEMM form (igemm.h) treats the nine taps as nine independent sets of rows and gathers X nine times; on gfx950
// every gather / staging instruction is time taken from the fp32 MFMA issue slot (tools/ubench/mfma_valu_overlap.hip).  Here a
// wave owns one (32 input channels) x (32 output channels) block of dW for ALL nine taps (9 x 16 accumulator registers):
//   * (in the shipped form the nine taps of a block are spread over three waves, one filter row each, sharing the patch)
//   * per chunk of 32 output pixels (32/W image rows) the workgroup DMA-loads (buffer_load ... lds, zero fill for rows outside
//     the image) the R+2 input rows of its 32-channel slice into a zero-haloed LDS patch [R+2][W+2][32] and each wave its
//     32-channel slice of dY [32 px][32];
//   * the dY fragment of an MFMA step is read once and used by all nine taps; the three horizontal taps share one 6-pixel
//     strip of X per lane (lane (ci, hi) holds 4 consecutive pixels per step, a shift by one pixel re-uses three of them);
//   * 144 MFMAs per wave per chunk against ~50 LDS reads and ~10 DMA instructions.
// Partial sums over the pixel ranges of the workgroups go through the split-K slabs of the implicit-GEMM path
// (slab[i = tap*C + ci][co]) and its fixed-order reduce + OIHW scatter, so results are deterministic.
#pragma once
#include "igemm_dma.h"

namespace hab {

struct Wgrad3x3Args {
    const float* x;    // [B][H][W][C]
    const float* dy;   // [B][H][W][Co]
    float* partial;    // [splits][9*C][Co]
    int B, H, W, C, Co;
    int chunks_per_block, total_chunks;  // chunk = 32 output pixels = 32/W rows of one image
    FastDiv dRowsPerImg;                  // chunks per image = H*W/32
};

// Workgroup = NWV x 3 waves: wave (ow, kh) owns output-channel block og*NWV + ow and filter row kh (3 taps, 48 accumulator
// registers); all waves share one 32-channel slice of X.  X patch and dY chunk are double-buffered: the DMA of chunk n+1 is
// issued right after the barrier that publishes chunk n and lands behind its MFMAs (one barrier per chunk).
template <int W, int NWV>
__global__ void __launch_bounds__(NWV * 192) wgrad3x3_patch_kernel(const Wgrad3x3Args a) {
    constexpr int R = 32 / W;                 // image rows per chunk
    constexpr int PW = W + 2, PR = R + 2;     // padded patch
    constexpr int XS = (PR * PW * 32 + 255) & ~255;  // floats per X buffer
    constexpr int YS = NWV * 1024;                   // floats per dY buffer
    constexpr int NWAVES = NWV * 3;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    float* Xs = smem;             // [2][PR][PW][32]
    float* Ys = smem + 2 * XS;    // [2][NWV][32 px][32 co]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int ow = wave / 3, kh = wave % 3;
    const int nob = a.Co / 32, ngrp = cdiv(nob, NWV);
    const int cb = blockIdx.x / ngrp, og = blockIdx.x % ngrp;
    const int ob = og * NWV + ow;              // this wave's output-channel block (may be >= nob in the last group)
    const bool wave_on = ob < nob;
    const int kz = blockIdx.z;
    const int c_begin = kz * a.chunks_per_block, c_end = min(a.total_chunks, c_begin + a.chunks_per_block);

    // zero both patches once: halo columns are never written again, interiors are overwritten by every chunk's DMA
    for (int i = t; i < 2 * XS; i += NWAVES * 64) Xs[i] = 0.f;
    __syncthreads();

    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[q][v] = 0.f;

    // DMA lane pattern: 8 pixels x 8 channel-quads per instruction
    const uint32_t xlane = (uint32_t)((lane >> 3) * a.C + (lane & 7) * 4) * 4u;
    const uint32_t ylane = (uint32_t)((lane >> 3) * a.Co + (lane & 7) * 4) * 4u;
    const size_t img_x = (size_t)a.H * W * a.C, img_y = (size_t)a.H * W * a.Co;
    constexpr int XI = PR * (W / 8);           // X DMA instructions per chunk
    constexpr int YI = NWV * 4;                // dY DMA instructions per chunk
    // fragment read offsets: MFMA step (c, s) consumes output pixel k = 8c + 4hi + s of the chunk -> (row k / W, column k % W)
    int xoff[4], yoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = 8 * c + 4 * hi;
        xoff[c] = ((k / W + kh) * PW + (k % W)) * 32 + li;   // tap (kh, kw), shift s: + (kw + s) * 32
        yoff[c] = ow * 1024 + k * 32 + li;                   // + s * 32
    }

    auto issue = [&](int ch, int buf) {
        int img, rc;
        a.dRowsPerImg.divmod(ch, img, rc);
        const int y0 = rc * R;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)img * img_x, 0,
                                                                            (int)(img_x * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + (size_t)img * img_y, 0,
                                                                            (int)(img_y * 4), 0x00020000);
        for (int ins = wave; ins < XI + YI; ins += NWAVES) {
            if (ins < XI) {  // X rows y0-1 .. y0+R : instruction (pr, q) = patch row pr, pixels 8q..8q+7
                const int pr = ins / (W / 8), q = ins % (W / 8);
                const int yy = y0 - 1 + pr;
                const bool ok = (unsigned)yy < (unsigned)a.H;  // rows outside the image: zeros through the range check
                const uint32_t soff = ok ? (uint32_t)((yy * W + 8 * q) * a.C + cb * 32) * 4u : 0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(Xs + buf * XS + (pr * PW + 1 + 8 * q) * 32),
                                                         16, (int)(ok ? xlane : DMA_OOB), (int)soff, 0, 0);
            } else {         // dY: output-channel block (og*NWV + o), pixels 8q..8q+7 of the chunk
                const int o = (ins - XI) >> 2, q = (ins - XI) & 3;
                const bool ok = og * NWV + o < nob;
                const uint32_t soff = ok ? (uint32_t)((y0 * W + 8 * q) * a.Co + (og * NWV + o) * 32) * 4u : 0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (__attribute__((address_space(3))) void*)(Ys + buf * YS + o * 1024 + q * 256), 16,
                                                         (int)(ok ? ylane : DMA_OOB), (int)soff, 0, 0);
            }
        }
    };
    auto compute = [&](const float* xs_, const float* ys_) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float bf[4], xs[6];
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[s] = ys_[yoff[c] + s * 32];
#pragma unroll
            for (int j = 0; j < 6; ++j) xs[j] = xs_[xoff[c] + j * 32];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[kw + s], bf[s], acc[kw], 0, 0, 0);
        }
    };

    if (c_begin < c_end) issue(c_begin, 0);
    for (int ch = c_begin; ch < c_end; ch += 2) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's share of chunk ch has landed
        __syncthreads();
        if (ch + 1 < c_end) issue(ch + 1, 1);
        if (wave_on) compute(Xs, Ys);
        if (ch + 1 < c_end) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (ch + 2 < c_end) issue(ch + 2, 0);
            if (wave_on) compute(Xs + XS, Ys + YS);
        }
    }
    if (!wave_on) return;
    // slab[kz][i = (kh*3 + kw)*C + cb*32 + ci][co = ob*32 + li]
    const size_t MN = (size_t)9 * a.C * a.Co;
    float* out = a.partial + (size_t)kz * MN;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = (v & 3) + 8 * (v >> 2) + 4 * hi;
            out[(size_t)((kh * 3 + kw) * a.C + cb * 32 + ci) * a.Co + ob * 32 + li] = acc[kw][v];
        }
}

inline bool wgrad3x3_patch_ok(const ConvWgradProb& p) {
    const ConvGeom& g = p.g;
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && (g.W == 32 || g.W == 16) && (g.H * g.W) % 32 == 0 && g.C % 32 == 0 &&
           g.Cout % 32 == 0 && p.Creal == g.C && (size_t)g.H * g.W * std::max(g.C, g.Cout) * 4 < 0x7fffffffull;
}

// ws must hold splits * 9*C*Co floats; the final sum + OIHW scatter (+ nothing else: resnet convs have no bias) is the
// implicit-GEMM path's split-K reduce.
inline int wgrad3x3_patch(const ConvWgradProb& p, float* ws, size_t ws_floats, hipStream_t stream) {
    const ConvGeom& g = p.g;
    Wgrad3x3Args a;
    a.x = p.x; a.dy = p.dy; a.partial = ws; a.B = g.B; a.H = g.H; a.W = g.W; a.C = g.C; a.Co = g.Cout;
    a.total_chunks = g.B * g.H * g.W / 32;
    a.dRowsPerImg = FastDiv(g.H * g.W / 32);
    const int nob = g.Cout / 32, ncb = g.C / 32;
    const int nwv = nob >= 4 ? 4 : nob;
    const int groups = ncb * cdiv(nob, nwv);
    const size_t MN = (size_t)9 * g.C * g.Cout;
    // ~12 waves per CU on 256 CUs
    int splits = cdiv(3072, groups * nwv * 3);
    if (splits > a.total_chunks) splits = a.total_chunks;
    while (splits > 1 && (size_t)splits * MN > ws_floats) splits >>= 1;
    if ((size_t)splits * MN > ws_floats || splits < 1) return HAB_ERR_ARG;
    a.chunks_per_block = cdiv(a.total_chunks, splits);
    splits = cdiv(a.total_chunks, a.chunks_per_block);
    const int R = 32 / g.W;
    const size_t lds = 2 * ((((size_t)(R + 2) * (g.W + 2) * 32 + 255) & ~(size_t)255) + (size_t)nwv * 1024) * sizeof(float);
    dim3 grid(groups, 1, splits);
#define HAB_W3_LAUNCH(WW, NN) wgrad3x3_patch_kernel<WW, NN><<<grid, NN * 192, lds, stream>>>(a)
    if (g.W == 32) {
        if (nwv == 1) HAB_W3_LAUNCH(32, 1); else if (nwv == 2) HAB_W3_LAUNCH(32, 2); else if (nwv == 3) HAB_W3_LAUNCH(32, 3); else HAB_W3_LAUNCH(32, 4);
    } else {
        if (nwv == 1) HAB_W3_LAUNCH(16, 1); else if (nwv == 2) HAB_W3_LAUNCH(16, 2); else if (nwv == 3) HAB_W3_LAUNCH(16, 3); else HAB_W3_LAUNCH(16, 4);
    }
#undef HAB_W3_LAUNCH
    HAB_LAUNCH_CHECK();
    igemm_splitk_reduce<ConvWgradProb>(p, ws, splits, stream);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
