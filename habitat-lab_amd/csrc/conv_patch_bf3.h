// conv_patch_bf3.h -- stride-1 3x3 convolutions (forward and data gradient) with the INPUT PATCH resident in LDS, on the bf16 matrix
// pipe with fp32-equivalent arithmetic (split scheme, sign schedule: igemm_bf3.h).
//
// Why.  In the im2col form (igemm_bf3.h) every activation is gathered, split into its three bf16 terms (4.5 VALU) and written to LDS
// once per filter tap that touches it: 9x for a 3x3 filter -- and for the 32-channel layers (ResNet layer1, SimpleCNN conv3: N = 32)
// each split value then feeds only 32 output columns, so those kernels are bound by the split's VALU issue at ~85 TFLOP/s-eq while the
// matrix pipe idles (NOTEBOOK.md 4).  Here a workgroup owns TH output rows x the full width of ONE image: it splits the (TH + 2) x
// (Wo + 2) x 32-channel input patch ONCE, keeps it in LDS as three bf16 planes, and the nine taps are nine shifted views of the
// same image: an A fragment is one ds_read_b128 per plane at pixel (ty + a, tx + b).  Per 32-channel chunk the VALU work drops from
// 9 x (BM x 32) to (TH+2)(Wo+2) x 32 splits; only the (small) weight tile is staged per filter row / tap.
//
// Lane <-> pixel: an MFMA tile's 32 rows are 32 / TW consecutive output rows x TW columns (TW = 8, 16, 32 >= Wo; columns >= Wo
// idle).  Pixel pitch in LDS 80 B: the 16 lanes of a ds_read_b128 service group are consecutive pixels (or two runs of them) and
// 5 * pixel mod 16 is a bijection -> conflict-free for TW = 32, <= 2-way at the row seam for TW < 32.
//
// Forward (ConvFwdProb) and data gradient (ConvDgradProb, stride 1: the same convolution with the taps flipped and padding
// K - 1 - pad) share the kernel; weights and epilogues come from the problem functors, so bias / ReLU / residual-add / ReLU-mask
// epilogues and the output row numbering are those of problems.h.
#pragma once
#include <atomic>
#include "igemm_bf3.h"

namespace hab {

constexpr int CPB_CC = 32;               // channels per chunk
constexpr int CPB_PIX = CPB_CC + 8;      // patch pixel pitch (bf16 elements): 80 B

struct PatchGeom {
    const float* in;        // NHWC input of the contraction: x (forward) or dY (data gradient)
    int Hi, Wi, Ci;         // its spatial size and channels
    int Ho, Wo;             // output pixel grid per image; output row m = (img * Ho + ho) * Wo + wo
    int KH, KW, off_h, off_w;  // output pixel (r, c), tap (a, b) reads input pixel (r + a + off_h, c + b + off_w)
    int flip;               // tap (a, b) is the problem's reduction block ((KH-1-a) * KW + (KW-1-b)) instead of (a * KW + b)
    int TH, tiles_per_img, PW, PP;  // output rows per tile, tiles per image, patch width / pixels
    FastDiv dPW;
    int nt_m, nt_n;
};

template <class P, int TW, int WM, int WN, int TN, int BT>
struct ConvPatchCfg {
    static constexpr int NT = WM * WN * 64;
    static constexpr int BN = WN * TN * 32;
    static constexpr int ROWS_PER_MT = 32 / TW;        // output rows of one 32-row MFMA tile
    static constexpr int TH = WM * ROWS_PER_MT;        // output rows per workgroup tile
    static constexpr int BP = BT * CPB_CC + 8;         // B row pitch (bf16 elements)
    static constexpr int B_PLANE = BN * BP;
    static constexpr int MAX_PP = (TH + 2) * (TW + 2);  // 3x3: patch pixels <= this
    static constexpr size_t LDS_BYTES = (size_t)(3 * MAX_PP * CPB_PIX + 3 * B_PLANE) * 2;
};

// both conv problems keep their weights as rows of 9 * Ci elements (Wf[co][(kh,kw,ci)], Wd[ci][(kh,kw,co)])
inline size_t cpb_w_elems(const ConvFwdProb& p) { return (size_t)p.N * p.K; }
inline size_t cpb_w_elems(const ConvDgradProb& p) { return (size_t)p.N * p.Kfull; }

// planes[(s * 3 + pl) * n + e] = bf16 term pl of (s ? -w[e] : w[e]): the weights split ONCE per call instead of once per workgroup
__global__ void cpb_split_weights(const float* __restrict__ w, size_t n, unsigned short* __restrict__ planes) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    unsigned h1, h2, h3;
    bf3_split(w[e], h1, h2, h3);
    planes[e] = (unsigned short)h1; planes[n + e] = (unsigned short)h2; planes[2 * n + e] = (unsigned short)h3;
    bf3_split(-w[e], h1, h2, h3);
    planes[3 * n + e] = (unsigned short)h1; planes[4 * n + e] = (unsigned short)h2; planes[5 * n + e] = (unsigned short)h3;
}

// PRE: weights arrive as bf16 planes (cpb_split_weights) -- B staging is a copy.  Persistent workgroups: blockIdx -> tiles b, b + G, ...
// (G a multiple of 8: a workgroup stays on its XCD's run of tiles); the NEXT tile's input patch is gathered into registers while
// the current tile's MFMAs run.
constexpr int CPB_DRAW_SLOTS = 64;
__device__ unsigned cpb_draw_pool[CPB_DRAW_SLOTS * 16];

template <class P, int TW, int WM, int WN, int TN, int BT, bool PRE>
__global__ void __launch_bounds__(WM* WN * 64) conv_patch_bf3_kernel(const P p, const PatchGeom gq, const int sign_schedule,
                                                                      const unsigned short* __restrict__ wplanes, const size_t wn_elems,
                                                                      unsigned* __restrict__ dyn_ctr) {
    using Cfg = ConvPatchCfg<P, TW, WM, WN, TN, BT>;
    constexpr int NT = Cfg::NT, BN = Cfg::BN, BP = Cfg::BP, TH = Cfg::TH;
    constexpr int PU = (Cfg::MAX_PP * (CPB_CC / 4) + NT - 1) / NT;  // patch gather units per thread and chunk
    static_assert(EpiV4<P>::value, "transposed-accumulator epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int PPL = gq.PP * CPB_PIX;                   // one patch plane (bf16 elements)
    unsigned short* Pa = smem16;                       // [3][PP][CPB_PIX]
    unsigned short* Bs = smem16 + 3 * Cfg::MAX_PP * CPB_PIX;  // [3][BN][BP]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;
    const int ntiles = gq.nt_m * gq.nt_n;
    auto tile_of = [&](int vb) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7, idx = vb >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    // this lane's output pixel inside a tile and its patch pixel for tap (0, 0)
    const int ty = wm * Cfg::ROWS_PER_MT + li / TW, tx = li % TW;
    const int txc = tx < gq.Wo ? tx : gq.Wo - 1;       // idle columns read a valid pixel
    const int pix0 = ty * gq.PW + txc;
    const int patch_units = gq.PP * (CPB_CC / 4);
    const int nchunks = gq.Ci / CPB_CC;
    const int nstage = 9 / BT;                          // B stages per chunk (BT = 3: one per filter row; BT = 1: one per tap)
    const typename P::KCtx kc = p.k_ctx(0, p.K);

    // ---- patch gather (registers) / split + store ----
    // what a thread gathers does not depend on the tile: unit j = patch pixel (pr, pc), channel quad cq -> row offset and element
    // offset relative to the tile's first input row are computed once; per tile only the row test and one add remain
    f32x4 pv[PU];
    int prow[PU], poff[PU];
#pragma unroll
    for (int j = 0; j < PU; ++j) {
        const int u = t + NT * j;
        const int pix = u >> 3, cq = u & 7;
        int pr, pc;
        gq.dPW.divmod(pix, pr, pc);
        const int w_ = pc + gq.off_w;
        const bool colok = (u < patch_units) & ((unsigned)w_ < (unsigned)gq.Wi);
        prow[j] = colok ? pr + gq.off_h : HAB_FAR;
        poff[j] = ((pr + gq.off_h) * gq.Wi + w_) * gq.Ci + cq * 4;
    }
    auto fetch_patch = [&](int tile_m, int chunk) {
        const int img = tile_m / gq.tiles_per_img, ho0 = (tile_m - img * gq.tiles_per_img) * TH;
        const float* base = gq.in + ((size_t)img * gq.Hi + ho0) * gq.Wi * gq.Ci + chunk * CPB_CC;
#pragma unroll
        for (int j = 0; j < PU; ++j) {
            const bool ok = (unsigned)(ho0 + prow[j]) < (unsigned)gq.Hi;
            pv[j] = ok ? ld4(base + poff[j]) : zero4();
        }
    };
    auto stage_patch = [&]() {
#pragma unroll
        for (int j = 0; j < PU; ++j) {
            const int u = t + NT * j;
            if (u < patch_units) {
                unsigned short* dst = Pa + (u >> 3) * CPB_PIX + (u & 7) * 4;
                bf3_store4(pv[j], dst, dst + PPL, dst + 2 * PPL);
            }
        }
    };
    // ---- weight tile: BT taps x 32 channels x BN rows ----
    constexpr int B_UNITS_TOTAL = BN * BT * (CPB_CC / 4);
    constexpr int B_UNITS = (B_UNITS_TOTAL + NT - 1) / NT;
    typename P::BRaw braw[PRE ? 1 : B_UNITS];
    u32x2 bpl[PRE ? B_UNITS : 1][3];
    // PRE: both problems keep W as rows of 9 * Ci elements, reduction block blk at blk * Ci: the element offset of unit j at stage st is
    // brow[j] + (btap[j] + st * bstep) * Ci + chunk * 32  (+ n0 * 9 * Ci), valid while n0 + bn[j] < N
    int brow[PRE ? B_UNITS : 1], btap[PRE ? B_UNITS : 1], bn[PRE ? B_UNITS : 1];
    const int bstep = gq.flip ? -BT : BT;
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) {
            const int u = t + NT * j;
            const int n = u / (BT * 8), rem = u - n * (BT * 8), bt = rem >> 3, cq = rem & 7;
            bn[j] = (B_UNITS_TOTAL % NT == 0 || u < B_UNITS_TOTAL) ? n : (1 << 28);
            brow[j] = n * 9 * gq.Ci + cq * 4;
            btap[j] = gq.flip ? 8 - bt : bt;
        }
    }
    auto fetch_b = [&](int n0, int chunk, int st, bool neg) {
        if constexpr (PRE) {
            const unsigned short* src0 = wplanes + (neg ? 3 * wn_elems : 0) + (size_t)n0 * 9 * gq.Ci + chunk * CPB_CC;
#pragma unroll
            for (int j = 0; j < B_UNITS; ++j) {
                const bool ok = n0 + bn[j] < p.N;
                const unsigned short* src = src0 + (ok ? brow[j] + (btap[j] + st * bstep) * gq.Ci : 0);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    bpl[j][pl] = *reinterpret_cast<const u32x2*>(src + pl * wn_elems);
                    if (!ok) { bpl[j][pl][0] = 0; bpl[j][pl][1] = 0; }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < B_UNITS; ++j) {
                const int u = t + NT * j;
                if (B_UNITS_TOTAL % NT == 0 || u < B_UNITS_TOTAL) {
                    const int n = u / (BT * 8), rem = u - n * (BT * 8), bt = rem >> 3, cq = rem & 7;
                    const int tapi = st * BT + bt;              // tap index a * 3 + b
                    const int blk = gq.flip ? 8 - tapi : tapi;
                    const int k = blk * gq.Ci + chunk * CPB_CC + cq * 4;
                    braw[j] = p.b_fetch(p.b_ctx(n0 + n), kc, p.b_key(kc, k, p.K));
                }
            }
        }
    };
    auto stage_b = [&](unsigned sgn) {
#pragma unroll
        for (int j = 0; j < B_UNITS; ++j) {
            const int u = t + NT * j;
            if (B_UNITS_TOTAL % NT == 0 || u < B_UNITS_TOTAL) {
                const int n = u / (BT * 8), rem = u - n * (BT * 8);
                unsigned short* dst = Bs + n * BP + rem * 4;
                if constexpr (PRE) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2*>(dst + pl * Cfg::B_PLANE) = bpl[j][pl];
                } else {
                    const f32x4 v = p.b_cvt(braw[j]);
                    f32x4 vs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vs[e] = __uint_as_float(__float_as_uint(v[e]) ^ sgn);
                    bf3_store4(vs, dst, dst + Cfg::B_PLANE, dst + 2 * Cfg::B_PLANE);
                }
            }
        }
    };

    // Tile order.  Static: workgroup b walks b, b + G, ... .  Dynamic (dyn_ctr, launches with more tiles than workgroups): the workgroups of
    // an XCD draw the next tile of THEIR XCD's run from a counter -- one that starts late or shares its CU (the persistent recurrence of
    // the second stream holds 32 CUs for the length of a time chunk: conv3 forward 127 us alone, 194 us beside it with the static order)
    // simply takes fewer tiles instead of holding the kernel's tail.  A tile's arithmetic does not depend on who computes it.
    __shared__ int s_draw[2];
    const bool dyn = dyn_ctr != nullptr;
    const int xcd = blockIdx.x & 7;
    const int run_len = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);
    int vb = blockIdx.x, vb_next = blockIdx.x + gridDim.x, it = 0;
    if (dyn) {
        if (t == 0) { s_draw[0] = (int)atomicAdd(dyn_ctr + xcd, 1u); s_draw[1] = (int)atomicAdd(dyn_ctr + xcd, 1u); }
        __syncthreads();
        const int i0 = s_draw[0], i1 = s_draw[1];
        __syncthreads();
        vb = i0 < run_len ? i0 * 8 + xcd : ntiles;
        vb_next = i1 < run_len ? i1 * 8 + xcd : ntiles;
    }
    // (dynamic order: the last workgroup to leave zeroes the counters for the next launch that is handed this slot)
    auto leave = [&]() {
        if (dyn && t == 0 && atomicAdd(dyn_ctr + 8, 1u) == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) dyn_ctr[i] = 0u;
        }
    };
    if (vb >= ntiles) { leave(); return; }
    fetch_patch(tile_of(vb) / gq.nt_n, 0);
    for (;;) {
        const int tile = tile_of(vb);
        const int tile_n = tile % gq.nt_n, tile_m = tile / gq.nt_n;
        const int n0 = tile_n * BN;
        const bool flip_all = sign_schedule && (((tile_m + tile_n) & 1) != 0);  // sign schedule (igemm_bf3.h)
        const unsigned sgn = flip_all ? 0x80000000u : 0u;
        const bool more = vb_next < ntiles;
        // the draw for the tile after next: written now, read behind this tile's barriers (two slots: the other one is read at the
        // end of the PREVIOUS tile, a full tile of barriers ago)
        if (dyn && more && t == 0) s_draw[it & 1] = (int)atomicAdd(dyn_ctr + xcd, 1u);

        f32x16 acc[TN], acc2[TN], tot[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[j][v] = 0.0f; acc2[j][v] = 0.0f; tot[j][v] = 0.0f; }

        for (int chunk = 0; chunk < nchunks; ++chunk) {
            fetch_b(n0, chunk, 0, flip_all);
            stage_patch();  // registers hold this (tile, chunk); the previous users of the LDS patch passed their last barrier
            // gather of the next patch: next chunk of this tile, or chunk 0 of the next tile -- in flight across the MFMAs below
            if (chunk + 1 < nchunks) fetch_patch(tile_m, chunk + 1);
            else if (more) fetch_patch(tile_of(vb_next) / gq.nt_n, 0);
            for (int st = 0; st < nstage; ++st) {
                stage_b(sgn);
                __syncthreads();
                if (st + 1 < nstage) fetch_b(n0, chunk, st + 1, flip_all);
                // BT * 2 fragment groups (tap, 16-channel k-group); the reads of group g + 1 are issued before the MFMAs of group g
                // (two register sets), so a wave does not sit out the LDS latency in front of every six MFMAs
                constexpr int NG = BT * (CPB_CC / 16);
                bf16x8 af[2][3], bf[2][TN][3];
                auto load_group = [&](int g, int buf) {
                    const int bt = g / (CPB_CC / 16), cg = g % (CPB_CC / 16);
                    const int tapi = st * BT + bt;
                    const int a = tapi / 3, b = tapi - a * 3;  // 3x3 (checked by the dispatcher)
                    const unsigned short* arow = Pa + (pix0 + a * gq.PW + b) * CPB_PIX + hi * 8 + cg * 16;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) af[buf][pl] = *reinterpret_cast<const bf16x8*>(arow + pl * PPL);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned short* src = Bs + ((wn * TN + j) * 32 + li) * BP + bt * CPB_CC + cg * 16 + hi * 8;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) bf[buf][j][pl] = *reinterpret_cast<const bf16x8*>(src + pl * Cfg::B_PLANE);
                    }
                };
                // Two accumulators, alternating MFMA by MFMA (groups g, g + 1 side by side).  Every MFMA rounds its accumulator to fp32;
                // six back-to-back updates of ONE accumulator per k-group measured 2.7x the rounding noise of the im2col kernel,
                // which interleaves two output tiles -- so this kernel interleaves two k-groups.  Summed before the epilogue.
                static_assert(NG % 2 == 0, "pairs of k-groups");
#pragma unroll
                for (int g = 0; g < NG; g += 2) {
                    load_group(g, 0);
                    load_group(g + 1, 1);
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0][j][PB[q]], af[0][PA[q]], acc[j], 0, 0, 0);
                            acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1][j][PB[q]], af[1][PA[q]], acc2[j], 0, 0, 0);
                        }
                }
                __syncthreads();
            }
            // per 32-channel chunk the two MFMA chains (54 instructions each) are folded into the running total with one fp32 add
            // each: the chains stay short whatever Cin is
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    tot[j][v] += acc[j][v] + acc2[j][v];
                    acc[j][v] = 0.0f; acc2[j][v] = 0.0f;
                }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][v] = flip_all ? -tot[j][v] : tot[j][v];
        // ---- epilogue of the problem (transposed accumulator: lane = output pixel, register quad = 4 consecutive channels) ----
        {
            const int img = tile_m / gq.tiles_per_img, ho = (tile_m - img * gq.tiles_per_img) * TH + ty;
            const int m = (tx < gq.Wo && ho < gq.Ho) ? (img * gq.Ho + ho) * gq.Wo + tx : p.M;
            const typename P::EpiRow erow = p.epi_row(m);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                typename P::EpiCol4 ecol[4];
                typename P::EpiAux4 aux[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    ecol[g] = p.epi_col4(n0 + (wn * TN + j) * 32 + 8 * g + 4 * hi);
                    aux[g] = p.epi_fetch4(erow, ecol[g]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    v[0] = acc[j][4 * g]; v[1] = acc[j][4 * g + 1]; v[2] = acc[j][4 * g + 2]; v[3] = acc[j][4 * g + 3];
                    p.epi_store4(erow, ecol[g], aux[g], v);
                }
            }
        }
        if (!more) { leave(); break; }
        vb = vb_next;
        if (dyn) { const int i2 = s_draw[it & 1]; vb_next = i2 < run_len ? i2 * 8 + xcd : ntiles; }
        else vb_next = vb + gridDim.x;
        ++it;
    }
}

template <class P, int TW, int WM, int WN, int TN, int BT>
inline int conv_patch_bf3_launch(const P& p, PatchGeom gq, float* ws, size_t ws_floats, hipStream_t stream) {
    using Cfg = ConvPatchCfg<P, TW, WM, WN, TN, BT>;
    gq.TH = Cfg::TH;
    gq.tiles_per_img = cdiv(gq.Ho, Cfg::TH);
    gq.PW = gq.Wo + 2;
    gq.PP = (Cfg::TH + 2) * gq.PW;
    gq.dPW = FastDiv(gq.PW);
    if (gq.PP > Cfg::MAX_PP) return 1;
    const int B = p.M / (gq.Ho * gq.Wo);
    gq.nt_m = B * gq.tiles_per_img;
    gq.nt_n = cdiv(p.N, Cfg::BN);
    const int ntiles = gq.nt_m * gq.nt_n;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    constexpr int wgs_per_cu = 2;  // measured (round 2): two persistent workgroups per CU
    const int occ = (int)(160 * 1024 / Cfg::LDS_BYTES) < wgs_per_cu ? (int)(160 * 1024 / Cfg::LDS_BYTES) : wgs_per_cu;
    int grid = 256 * (occ > 0 ? occ : 1);
    if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
    // weights as bf16 planes once per call when the launch is large enough to pay for the extra kernel
    constexpr int pre_min = 131072;
    const size_t wn_elems = cpb_w_elems(p);
    const bool pre = ws && ((reinterpret_cast<uintptr_t>(ws) & 15) == 0) && ws_floats * 4 >= wn_elems * 12 && p.M >= pre_min && (wn_elems % 4 == 0);
    hipError_t e = hipSuccess;
    // dynamic tile order for persistent launches: 8 per-XCD counters + a leave counter from a small pool of device slots (zero at load, zeroed
    // again by the last workgroup of the launch that used them -- no memset on the stream); launches in flight at once get different slots
    static const bool dyn_on = !hab_env_flag("HAB_NO_DYN_TILES");
    unsigned* ctr = nullptr;
    if (dyn_on && grid < ntiles) {
        static std::atomic<unsigned> next_slot{0};
        static unsigned* pool = nullptr;
        static const hipError_t sym_err = hipGetSymbolAddress(reinterpret_cast<void**>(&pool), HIP_SYMBOL(cpb_draw_pool));
        if (sym_err != hipSuccess || !pool) return HAB_ERR_ARG;
        ctr = pool + (size_t)(next_slot.fetch_add(1, std::memory_order_relaxed) % CPB_DRAW_SLOTS) * 16;
    }
    if (pre) {
        auto kern = conv_patch_bf3_kernel<P, TW, WM, WN, TN, BT, true>;
        // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
        static const hipError_t attr_err = (Cfg::LDS_BYTES > 64 * 1024) ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) : hipSuccess;
        if (attr_err != hipSuccess) return (int)attr_err;
        unsigned short* planes = reinterpret_cast<unsigned short*>(ws);
        cpb_split_weights<<<(unsigned)((wn_elems + 255) / 256), 256, 0, stream>>>(p.w, wn_elems, planes);
        HAB_LAUNCH_CHECK();
        kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, gq, sign_schedule, planes, wn_elems, ctr);
    } else {
        auto kern = conv_patch_bf3_kernel<P, TW, WM, WN, TN, BT, false>;
        // once per process and instantiation; thread-safe static initialisation (engines of several inference-worker threads launch concurrently)
        static const hipError_t attr_err = (Cfg::LDS_BYTES > 64 * 1024) ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) : hipSuccess;
        if (attr_err != hipSuccess) return (int)attr_err;
        kern<<<grid, Cfg::NT, Cfg::LDS_BYTES, stream>>>(p, gq, sign_schedule, nullptr, 0, ctr);
    }
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Picks a configuration for the output width / channel count; returns 1 when the shape is not covered (caller falls back).
template <class P>
inline int conv_patch_bf3_dispatch(const P& p, const PatchGeom& gq, float* ws, size_t ws_floats, hipStream_t stream) {
    if (gq.KH != 3 || gq.KW != 3 || gq.Ci % CPB_CC != 0 || (p.N & 31) || gq.Wo > 32 || gq.Wo < 5 || p.M % (gq.Ho * gq.Wo) != 0) return 1;
    // Measured (tools/bench_layers.py, 1024 frames): the patch form wins where the im2col form is bound by the A split, i.e. N = 32
    // (ResNet layer1 32x32: 92 -> 135 TFLOP/s-eq forward, 88 -> 136 data gradient; SimpleCNN conv3 forward 91 -> 119); at N >= 64
    // (layer2 / layer3, conv3's data gradient) it ties or loses against igemm_bf3's larger N tiles, so those stay there.
    if (p.N != 32) return 1;
    if (gq.Wo > 16) return conv_patch_bf3_launch<P, 32, 4, 1, 1, 3>(p, gq, ws, ws_floats, stream);
    if (gq.Wo > 8) return conv_patch_bf3_launch<P, 16, 4, 1, 1, 3>(p, gq, ws, ws_floats, stream);
    return 1;
}

}  // namespace hab
