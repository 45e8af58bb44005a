// rnn.hip -- sequence-packed GRU / LSTM state encoder (K9-K11) for gfx950.
//
// Reference: rl/models/rnn_state_encoder.py:187-277,301-350 wraps the minibatch into a
// PackedSequence (one sequence per episode fragment, sorted by length) and calls cuDNN/ATen.
// Here the same packing drives hand-written kernels:
//   1. the input projections of ALL T*n frames are one MFMA contraction (igemm, frame order),
//   2. the recurrence walks the packed steps s = 0..max_len-1; step s touches the first
//      num_seqs_at_step[s] fragments.  One launch per step: a 16-row x 16-unit tile per workgroup,
//      its 4 waves split the K=H reduction and meet in LDS, then the gate math is fused in the
//      epilogue.  Operands go global->VGPR as 16-byte loads in MFMA fragment order
//      (v_mfma_f32_16x16x4_f32; W_hh is 3-4 MB and L2-resident) -- no LDS staging, this phase is
//      latency-bound, not bandwidth-bound.
//   3. outputs are written straight in frame order (the inverse permutation of
//      build_rnn_out_from_seq is free), the hidden state entering every step is kept for BPTT.
// BPTT runs the steps in reverse (one launch per step: transposed mat-vec of step s+1 fused with the gate gradients of step s) and
// finishes with three contractions over all frames (dW_ih, dW_hh, dX) and two column sums.
#include "ops.h"
#include "rnn_gates.h"
#include "rnn_persist.h"
#include "../../include/habitat_amd.h"

namespace hab {

constexpr int RNN_GRU = 0, RNN_LSTM = 1;

// hinit[q] = masks[frame frag_start[q]] ? h0[env frag_env[q]] : 0       (build_rnn_inputs :232-239)
__global__ void rnn_frag_init_kernel(const float* __restrict__ h0, const int* __restrict__ env_rows, int env_stride,
                                     const uint8_t* __restrict__ masks, const int* __restrict__ mask_rows,
                                     const int* __restrict__ frag_env, const int* __restrict__ frag_start, int F, int H,
                                     float* __restrict__ hinit, const int* __restrict__ env_first) {
    const int q = blockIdx.x;
    if (q >= F) return;
    const int f = frag_start ? frag_start[q] : q;
    int e = frag_env ? frag_env[q] : q;
    if (env_first) e = env_first[e];  // VER minibatch: the frame that holds batch environment e's stored hidden state
    const bool keep = masks[mask_rows ? mask_rows[f] : f] != 0;
    const float* src = h0 + (size_t)(env_rows ? env_rows[e] : e) * env_stride;
    for (int u = threadIdx.x; u < H; u += blockDim.x) hinit[(size_t)q * H + u] = keep ? src[u] : 0.f;
}

int rnn_frag_init(const float* h0, const int* env_rows, int env_stride, const uint8_t* masks, const int* mask_rows,
                  const int* frag_env, const int* frag_start, int F, int H, float* hinit, hipStream_t stream, const int* env_first) {
    if (!h0 || !masks || !hinit || F <= 0 || H <= 0) return HAB_ERR_ARG;
    rnn_frag_init_kernel<<<F, 128, 0, stream>>>(h0, env_rows, env_stride, masks, mask_rows, frag_env, frag_start, F, H, hinit, env_first);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

struct StepArgs {
    int R, H;
    const float* hp_base; const int* hp_idx; int hp_stride;   // hidden entering the step, row q
    const float* cp_base; const int* cp_idx; int cp_stride;   // LSTM cell entering the step
    const uint8_t* row_mask;                                   // optional [R]: state entering row q is zeroed where 0 (episode start)
    const int* out_idx;                                        // row q -> frame (null = q)
    const float* gi;                                           // [frames][G*H] input projection incl. b_ih
    const float* w_hh; const float* b_hh;                      // [G*H][H]
    float* gates; float* hn; float* hprev; float* cprev; float* c; // saved for BPTT (may be null in inference)
    float* out; int out_stride;                                // h' per frame
    float* c_out; int c_out_stride;                            // LSTM c' (inference path); training uses `c`
    // Inference form with the INPUT PROJECTION inside the step (rollout `act`: 64 rows): x row q, W_ih rows padded to a multiple of 16
    // columns (x_dim, zero beyond in_dim), b_ih.  `gi` is then unused: the separate M = 64 contraction + its split-K second pass (two
    // launches per layer and environment step) disappear.
    const float* x_base = nullptr; int x_stride = 0; const float* w_ih = nullptr; int w_ih_ld = 0; int x_dim = 0; const float* b_ih = nullptr;
    const int* x_idx = nullptr;                                // row q -> row of x_base (null = q): the layer below's output frames
};
// The steps that one launch of a layer wavefront runs side by side (blockIdx.z = layer): layer l works on packed step w - l while
// layer l - 1 works on step w -- it only needs the layer below's output of ITS step, which the previous launch wrote.
constexpr int RNN_MAX_WAVE_LAYERS = 4;
struct StepArgsN {
    StepArgs l[RNN_MAX_WAVE_LAYERS];
};

// One workgroup = 16 rows x 16 hidden units x G gates; NW waves split K = H.  The recurrence is latency-bound (a step is a
// few MFLOP): each wave issues its loads in batches of U K-chunks before the dependent MFMAs, so a step costs ~1-2 L2 round
// trips instead of one per chunk.
template <int G, int NW>
__device__ __forceinline__ void rnn_step_body(const StepArgs& a) {
    // (four accumulators in the fused-projection form of BOTH cell types: LSTM i, f, g, o with x- and h-parts summed; GRU r, z summed,
    //  the n gate's h-part and x-part apart -- n = tanh(gi_n + r * gh_n))
    constexpr int GA = 4;
    __shared__ float red[NW][GA][256];
    constexpr int U = 4;
    const bool fused_x = a.x_base != nullptr;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * 16, u0 = blockIdx.y * 16;
    const int i = lane & 15, kg = lane >> 4;
    const int H = a.H;
    if (row0 >= a.R) return;  // (a wavefront launch is sized for its widest layer)
    const int q = min(row0 + i, a.R - 1);
    const float* hrow = a.hp_base + (size_t)(a.hp_idx ? a.hp_idx[q] : q) * a.hp_stride;
    const float keep = (a.row_mask && !a.row_mask[q]) ? 0.f : 1.f;  // h * mask (rnn_state_encoder.py:308-311), applied to the operand
    const int kq = H / NW;  // per-wave K range (multiple of 16)
    const int kb = wave * kq;
    // Gate-phase operands of this thread's (row, unit) element -- frame index, input projection, previous state, bias: they do not
    // depend on the mat-vec, so their two dependent L2 round trips (index, then data) run underneath it instead of after it.
    const int r_ = t >> 4, u_ = t & 15, qq_ = row0 + r_;
    const bool gate_thread = (t < 256) & (qq_ < a.R);
    int f_pre = 0;
    float hp_pre = 0.f, cp_pre = 0.f, gi_pre[G], bhh_pre[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { gi_pre[g] = 0.f; bhh_pre[g] = 0.f; }
    if (gate_thread) {
        f_pre = a.out_idx ? a.out_idx[qq_] : qq_;
        const bool kept_ = !(a.row_mask && !a.row_mask[qq_]);
        const size_t prow_ = (size_t)(a.hp_idx ? a.hp_idx[qq_] : qq_);
        hp_pre = kept_ ? a.hp_base[prow_ * a.hp_stride + u0 + u_] : 0.f;
        if constexpr (G == 4) cp_pre = kept_ ? a.cp_base[(size_t)(a.cp_idx ? a.cp_idx[qq_] : qq_) * a.cp_stride + u0 + u_] : 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            gi_pre[g] = fused_x ? a.b_ih[g * H + u0 + u_] : a.gi[(size_t)f_pre * G * H + g * H + u0 + u_];
            bhh_pre[g] = a.b_hh[g * H + u0 + u_];
        }
    }
    f32x4 acc[GA];
#pragma unroll
    for (int g = 0; g < GA; ++g) { acc[g][0] = 0.f; acc[g][1] = 0.f; acc[g][2] = 0.f; acc[g][3] = 0.f; }
    const float* wrow[G];
#pragma unroll
    for (int g = 0; g < G; ++g) wrow[g] = a.w_hh + (size_t)(g * H + u0 + i) * H;
    int c = 0;
    for (; c + 16 * U <= kq; c += 16 * U) {
        f32x4 av[U], bv[U][G];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int k = kb + c + 16 * j + 4 * kg;
            av[j] = *reinterpret_cast<const f32x4*>(hrow + k) * keep;
#pragma unroll
            for (int g = 0; g < G; ++g) bv[j][g] = *reinterpret_cast<const f32x4*>(wrow[g] + k);
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][s], bv[j][g][s], acc[g], 0, 0, 0);
    }
    for (; c < kq; c += 16) {
        const int k = kb + c + 4 * kg;
        const f32x4 av = *reinterpret_cast<const f32x4*>(hrow + k) * keep;
        f32x4 bv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) bv[g] = *reinterpret_cast<const f32x4*>(wrow[g] + k);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[g][s], acc[g], 0, 0, 0);
    }
    if (fused_x) {  // + W_ih x: K-chunks of 16 dealt round-robin over the waves, two chunks of loads in flight
        const float* xrow = a.x_base + (size_t)(a.x_idx ? a.x_idx[q] : q) * a.x_stride;
        const float* wi[G];
#pragma unroll
        for (int g = 0; g < G; ++g) wi[g] = a.w_ih + (size_t)(g * H + u0 + i) * a.w_ih_ld;
        const int nch = a.x_dim >> 4;
        for (int c0 = wave; c0 < nch; c0 += 2 * NW) {
            f32x4 xv[2], wv[2][G];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cc = c0 + j * NW;
                const int k = (cc < nch ? cc : c0) * 16 + 4 * kg;
                xv[j] = *reinterpret_cast<const f32x4*>(xrow + k);
#pragma unroll
                for (int g = 0; g < G; ++g) wv[j][g] = *reinterpret_cast<const f32x4*>(wi[g] + k);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (c0 + j * NW >= nch) continue;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int slot = (G == 3 && g == 2) ? 3 : g;  // GRU: the n gate's input part stays apart
                        acc[slot] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[j][s], wv[j][g][s], acc[slot], 0, 0, 0);
                    }
            }
        }
    }
    // D layout (16x16): col = lane & 15 (unit), row = (lane >> 4) * 4 + v
#pragma unroll
    for (int g = 0; g < GA; ++g)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (g < G || fused_x) red[wave][g][(kg * 4 + v) * 16 + i] = acc[g][v];
    __syncthreads();
    if (t >= 256) return;
    const int r = t >> 4, u = t & 15;  // one (row, unit) per thread
    const int qq = row0 + r;
    if (qq >= a.R) return;
    float gh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4) sum += (red[w][g][t] + red[w + 1][g][t]) + (red[w + 2][g][t] + red[w + 3][g][t]);
        gh[g] = sum + bhh_pre[g];
    }
    if (G == 3 && fused_x) {  // gi_n = W_in x + b_in (gi_pre holds the bias; for r, z -- and every LSTM gate -- the x-part is inside gh)
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4) sum += (red[w][3][t] + red[w + 1][3][t]) + (red[w + 2][3][t] + red[w + 3][3][t]);
        gi_pre[2] += sum;
    }
    const int f = f_pre;
    const int uu = u0 + u;
    const float hp = hp_pre;
    if constexpr (G == 3) {
        float rg, zg, ng;
        const float hnew = gru_cell_fwd(gi_pre[0], gi_pre[1], gi_pre[2], gh[0], gh[1], gh[2], hp, rg, zg, ng);
        a.out[(size_t)f * a.out_stride + uu] = hnew;
        if (a.gates) {
            float* gs = a.gates + (size_t)f * 3 * H;
            gs[uu] = rg; gs[H + uu] = zg; gs[2 * H + uu] = ng;
            a.hn[(size_t)f * H + uu] = gh[2];
            a.hprev[(size_t)f * H + uu] = hp;
        }
    } else {
        const float cp = cp_pre;
        float ig, fg, gg, og, cn;
        const float hnew = lstm_cell_fwd(gi_pre[0] + gh[0], gi_pre[1] + gh[1], gi_pre[2] + gh[2], gi_pre[3] + gh[3], cp, ig, fg, gg, og, cn);
        a.out[(size_t)f * a.out_stride + uu] = hnew;
        if (a.c_out) a.c_out[(size_t)f * a.c_out_stride + uu] = cn;
        if (a.gates) {
            float* gs = a.gates + (size_t)f * 4 * H;
            gs[uu] = ig; gs[H + uu] = fg; gs[2 * H + uu] = gg; gs[3 * H + uu] = og;
            a.hprev[(size_t)f * H + uu] = hp;
            a.cprev[(size_t)f * H + uu] = cp;
            a.c[(size_t)f * H + uu] = cn;
        }
    }
}

template <int G, int NW>
__global__ void __launch_bounds__(64 * NW) rnn_step_kernel(const StepArgs a) { rnn_step_body<G, NW>(a); }
template <int G, int NW>
__global__ void __launch_bounds__(64 * NW) rnn_step_wave_kernel(const StepArgsN a) { rnn_step_body<G, NW>(a.l[blockIdx.z]); }

static int launch_step(int rnn_type, const StepArgs& a, hipStream_t stream) {
    if (a.R <= 0) return HAB_OK;
    if (a.H % 64) return HAB_ERR_UNSUPPORTED;
    dim3 grid(cdiv(a.R, 16), a.H / 16);
    const bool wide = a.H % 128 == 0;  // 8 waves need H / 8 to be a multiple of the 16-wide K chunk
    if (rnn_type == RNN_GRU) {
        if (wide) rnn_step_kernel<3, 8><<<grid, 512, 0, stream>>>(a);
        else rnn_step_kernel<3, 4><<<grid, 256, 0, stream>>>(a);
    } else {
        if (wide) rnn_step_kernel<4, 8><<<grid, 512, 0, stream>>>(a);
        else rnn_step_kernel<4, 4><<<grid, 256, 0, stream>>>(a);
    }
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

int rnn_seq_layer_forward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx,
                          const float* hinit, const float* cinit, const PackInfo& pk, float* ws, size_t ws_floats,
                          hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    HAB_TRY(linear_fwd(x, ldx, lp.w_ih, lp.in_dim, lp.b_ih, wk.gi, G * H, pk.P, G * H, lp.in_dim, 0, 0, ws, ws_floats, stream));
    for (int s = 0; s < pk.max_len; ++s) {
        StepArgs a;
        a.R = pk.num_seqs_at_step[s]; a.H = H;
        if (s == 0) {
            a.hp_base = hinit; a.hp_idx = nullptr; a.hp_stride = H;
            a.cp_base = cinit; a.cp_idx = nullptr; a.cp_stride = H;
        } else {
            a.hp_base = wk.out; a.hp_idx = pk.select_inds + pk.step_offsets[s - 1]; a.hp_stride = H;
            a.cp_base = wk.c; a.cp_idx = a.hp_idx; a.cp_stride = H;
        }
        a.row_mask = nullptr;
        a.out_idx = pk.select_inds + pk.step_offsets[s];
        a.gi = wk.gi; a.w_hh = lp.w_hh; a.b_hh = lp.b_hh;
        a.gates = wk.gates; a.hn = wk.hn; a.hprev = wk.hprev; a.cprev = wk.cprev; a.c = wk.c;
        a.out = wk.out; a.out_stride = H; a.c_out = nullptr; a.c_out_stride = 0;
        HAB_TRY(launch_step(rnn_type, a, stream));
    }
    return HAB_OK;
}

// The L layers of the packed recurrence as a WAVEFRONT: launch w runs step w - l of layer l for every layer at once (grid z = layer),
// max_len + L - 1 launches instead of L * max_len -- the chain of dependent ~9 us launches is what the recurrent encoder costs.
// Layer 0 reads its input projection from the one large contraction over all frames, as in rnn_seq_layer_forward; a layer above cannot
// (its input is finished step by step), so its projection runs inside the step kernel (the rollout's fused form) on the rows the layer
// below wrote in the previous launch.  Saves for BPTT as in the layer-by-layer form; wk[l].gi of the upper layers is not written.
// Returns 1 when the form does not apply (the caller runs layer by layer).
int rnn_seq_wave_forward(int rnn_type, int H, int L, const RnnLayerParams* lp, const RnnWork* wk, const float* x, int ldx,
                         const float* const* hinit, const float* const* cinit, const PackInfo& pk, float* ws, size_t ws_floats,
                         hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    if (L < 2 || L > RNN_MAX_WAVE_LAYERS || (H % 64) || pk.max_len < 1) return 1;
    for (int l = 1; l < L; ++l)
        if (!lp[l].w_ih_pad || lp[l].w_ih_ld != H || (reinterpret_cast<uintptr_t>(lp[l].w_ih_pad) & 15)) return 1;
    HAB_TRY(linear_fwd(x, ldx, lp[0].w_ih, lp[0].in_dim, lp[0].b_ih, wk[0].gi, G * H, pk.P, G * H, lp[0].in_dim, 0, 0, ws, ws_floats, stream));
    const bool wide = H % 128 == 0;
    for (int w = 0; w < pk.max_len + L - 1; ++w) {
        StepArgsN n{};
        int rmax = 0;
        for (int l = 0; l < L; ++l) {
            StepArgs& a = n.l[l];
            const int s = w - l;
            a.H = H;
            a.R = (s >= 0 && s < pk.max_len) ? pk.num_seqs_at_step[s] : 0;
            if (a.R <= 0) { a.R = 0; continue; }
            rmax = std::max(rmax, a.R);
            if (s == 0) {
                a.hp_base = hinit[l]; a.hp_idx = nullptr; a.hp_stride = H;
                a.cp_base = cinit[l]; a.cp_idx = nullptr; a.cp_stride = H;
            } else {
                a.hp_base = wk[l].out; a.hp_idx = pk.select_inds + pk.step_offsets[s - 1]; a.hp_stride = H;
                a.cp_base = wk[l].c; a.cp_idx = a.hp_idx; a.cp_stride = H;
            }
            a.row_mask = nullptr;
            a.out_idx = pk.select_inds + pk.step_offsets[s];
            a.gi = wk[l].gi; a.w_hh = lp[l].w_hh; a.b_hh = lp[l].b_hh;
            a.gates = wk[l].gates; a.hn = wk[l].hn; a.hprev = wk[l].hprev; a.cprev = wk[l].cprev; a.c = wk[l].c;
            a.out = wk[l].out; a.out_stride = H; a.c_out = nullptr; a.c_out_stride = 0;
            if (l > 0) {
                a.x_base = wk[l - 1].out; a.x_stride = H; a.x_idx = a.out_idx; a.w_ih = lp[l].w_ih_pad; a.w_ih_ld = H; a.x_dim = H;
                a.b_ih = lp[l].b_ih;
            }
        }
        if (rmax == 0) continue;
        const dim3 grid(cdiv(rmax, 16), H / 16, L);
        if (rnn_type == RNN_GRU) {
            if (wide) rnn_step_wave_kernel<3, 8><<<grid, 512, 0, stream>>>(n);
            else rnn_step_wave_kernel<3, 4><<<grid, 256, 0, stream>>>(n);
        } else {
            if (wide) rnn_step_wave_kernel<4, 8><<<grid, 512, 0, stream>>>(n);
            else rnn_step_wave_kernel<4, 4><<<grid, 256, 0, stream>>>(n);
        }
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

// Inference step (rollout `act`, rnn_state_encoder.py:301-316): n rows, masked state supplied dense.
int rnn_step_layer_forward(int rnn_type, int H, const RnnLayerParams& lp, const float* x, int ldx, const float* h_in, int h_in_stride,
                           const float* c_in, int c_in_stride, const uint8_t* masks, int n, float* gi_scratch, float* h_out,
                           int h_out_stride, float* c_out, int c_out_stride, float* ws, size_t ws_floats, hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    // input projection inside the step kernel when the engine supplies W_ih with padded, aligned rows: x rows must be readable (and
    // zero where W_ih's padding is, or at least finite) up to w_ih_ld
    const bool fuse = lp.w_ih_pad && lp.w_ih_ld > 0 && (lp.w_ih_ld & 15) == 0 && ldx >= lp.w_ih_ld && (ldx & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(lp.w_ih_pad)) & 15) == 0;
    if (!fuse) HAB_TRY(linear_fwd(x, ldx, lp.w_ih, lp.in_dim, lp.b_ih, gi_scratch, G * H, n, G * H, lp.in_dim, 0, 0, ws, ws_floats, stream));
    StepArgs a;
    if (fuse) { a.x_base = x; a.x_stride = ldx; a.w_ih = lp.w_ih_pad; a.w_ih_ld = lp.w_ih_ld; a.x_dim = lp.w_ih_ld; a.b_ih = lp.b_ih; }
    a.R = n; a.H = H;
    a.hp_base = h_in; a.hp_idx = nullptr; a.hp_stride = h_in_stride;
    a.cp_base = c_in; a.cp_idx = nullptr; a.cp_stride = c_in_stride;
    a.row_mask = masks;  // the episode-start mask is applied to the state operand inside the kernel
    a.out_idx = nullptr; a.gi = gi_scratch; a.w_hh = lp.w_hh; a.b_hh = lp.b_hh;
    a.gates = nullptr; a.hn = nullptr; a.hprev = nullptr; a.cprev = nullptr; a.c = nullptr;
    a.out = h_out; a.out_stride = h_out_stride; a.c_out = c_out; a.c_out_stride = c_out_stride;
    return launch_step(rnn_type, a, stream);
}

// ------------------------------------------- BPTT ---------------------------------------------
// One launch per packed step s (walking backwards): workgroup (row tile, 16-unit tile)
//   A. carry tile = dgh[frames of step s+1] * W_hh   (rows < R_next; K = G*H split over NW waves, reduced in LDS)
//      -- the transposed mat-vec of step s+1, which only this step's gate math consumes;
//   B. gate math of step s for the same (row, unit) elements: dh = dout + carry + direct term, then the pre-activation
//      gradients dgi / dgh of all G gates of that unit, and the direct term / cell carry for step s-1.
// Row q is the same fragment at every step (fragments are sorted by length), so a (row, unit) element is owned by the same
// thread of the same workgroup in consecutive launches: the direct/cell carries are private read-modify-write.
struct BwdStepArgs {
    int R, H, K, R_next;
    const int* idx;        // row q -> frame at step s
    const int* idx_next;   // row q -> frame at step s+1
    const uint8_t* next_keep;  // time-major form: [R_next] episode-start mask of the frames at step s+1 (0: that frame starts from the zero
                               // state, nothing flows back into row q); null in the packed form, where such a row is simply absent
    const float* dout;     // [frames][H] gradient wrt the layer output
    const float* w_hh_t;   // [H][K]
    float* dh_direct;      // [F][H] in: direct part of dh from step s+1 (GRU dh*z); out: the same for step s-1
    float* dc_carry;       // LSTM [F][H] in/out
    const float* gates; const float* hn; const float* hprev; const float* cprev; const float* c;
    float* dgi; float* dgh;  // [frames][K]; LSTM: dgh == dgi
    // Layer wavefront (rnn_seq_wave_backward): the gradient wrt this layer's output is the data gradient of the layer ABOVE,
    // dgi_up[frame] * W_ih_up, computed here for the rows of this step (the layer above finished this step in the previous launch)
    // instead of by one contraction over all frames after the layer above has walked its whole chain.  `dout` may then be null.
    const float* up_dgi = nullptr;     // [frames][K]
    const float* up_w_ih_t = nullptr;  // [H][K]: W_ih of the layer above, transposed
};
struct BwdStepArgsN {
    BwdStepArgs l[RNN_MAX_WAVE_LAYERS];
};

// UP: the launch may carry the layer-above term (layer wavefront); the single-layer forms do not allocate its reduction buffer -- LDS is what
// decides whether a step workgroup fits on a CU beside a 148 KB workgroup of the encoder's dense / strip kernels (second-stream recurrence).
template <int G, int NW, bool UP>
__device__ __forceinline__ void rnn_bwd_step_body(const BwdStepArgs& a) {
    __shared__ float red[NW][256];
    __shared__ float red_up[UP ? NW : 1][UP ? 256 : 1];
    constexpr int U = 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * 16, u0 = blockIdx.y * 16;
    const int H = a.H;
    if (row0 >= a.R) return;  // (a wavefront launch is sized for its widest layer)
    const bool has_carry = row0 < a.R_next;  // workgroup-uniform
    const bool has_up = UP && a.up_dgi != nullptr;
    // gate-phase operands of this thread's (row, unit) element, fetched underneath the carry mat-vec (see rnn_step_kernel)
    const int r_ = t >> 4, uu_ = u0 + (t & 15), q_ = row0 + r_;
    const bool gate_thread = (t < 256) & (q_ < a.R);
    int f_pre = 0;
    float dout_pre = 0.f, dhd_pre = 0.f, dcc_pre = 0.f, gs_pre[G], hp_pre = 0.f, hn_pre = 0.f, cn_pre = 0.f, cp_pre = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) gs_pre[g] = 0.f;
    if (gate_thread) {
        f_pre = a.idx[q_];
        const size_t qo_ = (size_t)q_ * H + uu_, fo_ = (size_t)f_pre * H + uu_;
        dout_pre = a.dout ? a.dout[fo_] : 0.f;
        if (q_ < a.R_next && (!a.next_keep || a.next_keep[q_])) {
            if constexpr (G == 3) dhd_pre = a.dh_direct[qo_];
            else dcc_pre = a.dc_carry[qo_];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) gs_pre[g] = a.gates[(size_t)f_pre * G * H + g * H + uu_];
        if constexpr (G == 3) { hp_pre = a.hprev[fo_]; hn_pre = a.hn[fo_]; }
        else { cn_pre = a.c[fo_]; cp_pre = a.cprev[fo_]; }
    }
    // tile[16 rows][16 units] = A[rows][K] * B[units][K]^T, K split over the waves, partial tiles into `out`
    auto matvec = [&](const float* arow, const float* brow, float (*out)[256]) {
        const int i = lane & 15, kg = lane >> 4;
        const int kq = a.K / NW, kb = wave * kq;
        f32x4 acc;
        acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f; acc[3] = 0.f;
        int c = 0;
        for (; c + 16 * U <= kq; c += 16 * U) {
            f32x4 av[U], bv[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int k = kb + c + 16 * j + 4 * kg;
                av[j] = *reinterpret_cast<const f32x4*>(arow + k);
                bv[j] = *reinterpret_cast<const f32x4*>(brow + k);
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][s], bv[j][s], acc, 0, 0, 0);
        }
        for (; c < kq; c += 16) {
            const int k = kb + c + 4 * kg;
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + k);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(brow + k);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) out[wave][(kg * 4 + v) * 16 + i] = acc[v];
    };
    if constexpr (UP) {
        if (has_up) {  // gradient wrt this layer's output at the frames of THIS step
            const int i = lane & 15;
            const int q = min(row0 + i, a.R - 1);
            matvec(a.up_dgi + (size_t)a.idx[q] * a.K, a.up_w_ih_t + (size_t)(u0 + i) * a.K, red_up);
        }
    }
    if (has_carry) {
        const int i = lane & 15;
        const int q = min(row0 + i, a.R_next - 1);
        matvec(a.dgh + (size_t)a.idx_next[q] * a.K, a.w_hh_t + (size_t)(u0 + i) * a.K, red);
    }
    if (has_up || has_carry) __syncthreads();
    if (t >= 256) return;
    const int r = t >> 4, uu = u0 + (t & 15), q = row0 + r;
    if (q >= a.R) return;
    const int f = f_pre;
    const size_t qo = (size_t)q * H + uu;
    float dh = dout_pre;
    if constexpr (UP) {
        if (has_up) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w += 4) sum += (red_up[w][t] + red_up[w + 1][t]) + (red_up[w + 2][t] + red_up[w + 3][t]);
            dh += sum;
        }
    }
    const bool carried = (q < a.R_next) && (!a.next_keep || a.next_keep[q]);
    if (carried) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4) sum += (red[w][t] + red[w + 1][t]) + (red[w + 2][t] + red[w + 3][t]);
        dh += sum;
        if constexpr (G == 3) dh += dhd_pre;
    }
    if constexpr (G == 3) {
        float dr_pre, dz_pre, dn_pre, dhn_pre, direct;
        gru_cell_bwd(dh, gs_pre[0], gs_pre[1], gs_pre[2], hp_pre, hn_pre, dr_pre, dz_pre, dn_pre, dhn_pre, direct);
        float* gi = a.dgi + (size_t)f * 3 * H;
        float* gh = a.dgh + (size_t)f * 3 * H;
        gi[uu] = dr_pre; gi[H + uu] = dz_pre; gi[2 * H + uu] = dn_pre;
        gh[uu] = dr_pre; gh[H + uu] = dz_pre; gh[2 * H + uu] = dhn_pre;
        a.dh_direct[qo] = direct;
    } else {
        float di_pre, df_pre, dg_pre, do_pre, cc;
        lstm_cell_bwd(dh, carried, dcc_pre, gs_pre[0], gs_pre[1], gs_pre[2], gs_pre[3], cn_pre, cp_pre, di_pre, df_pre, dg_pre, do_pre, cc);
        float* gi = a.dgi + (size_t)f * 4 * H;
        gi[uu] = di_pre;
        gi[H + uu] = df_pre;
        gi[2 * H + uu] = dg_pre;
        gi[3 * H + uu] = do_pre;
        a.dc_carry[qo] = cc;
    }
}

template <int G, int NW>
__global__ void __launch_bounds__(64 * NW) rnn_bwd_step_kernel(const BwdStepArgs a) { rnn_bwd_step_body<G, NW, false>(a); }
template <int G, int NW>
__global__ void __launch_bounds__(64 * NW) rnn_bwd_step_wave_kernel(const BwdStepArgsN a) { rnn_bwd_step_body<G, NW, true>(a.l[blockIdx.z]); }

int rnn_seq_layer_backward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx,
                           const float* dout, float* dx, int lddx, const float* dx_mask, int ldmask, int mask_cols,
                           const PackInfo& pk, float* scratch /* 3*F*H floats */, float* ws, size_t ws_floats, hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    if (H % 64) return HAB_ERR_UNSUPPORTED;
    float* dh_direct = scratch;
    float* dc_carry = scratch + (size_t)pk.F * H;
    float* dgh = (rnn_type == RNN_GRU) ? wk.dgh : wk.dgi;
    const int K = G * H;
    const bool wide = K % 128 == 0;  // 8 waves need K / 8 to be a multiple of the 16-wide K chunk
    for (int s = pk.max_len - 1; s >= 0; --s) {
        BwdStepArgs g;
        g.R = pk.num_seqs_at_step[s]; g.H = H; g.K = K;
        g.R_next = (s + 1 < pk.max_len) ? pk.num_seqs_at_step[s + 1] : 0;
        g.idx = pk.select_inds + pk.step_offsets[s];
        g.idx_next = g.R_next ? pk.select_inds + pk.step_offsets[s + 1] : g.idx;
        g.next_keep = nullptr;
        g.dout = dout; g.w_hh_t = lp.w_hh_t; g.dh_direct = dh_direct; g.dc_carry = dc_carry;
        g.gates = wk.gates; g.hn = wk.hn; g.hprev = wk.hprev; g.cprev = wk.cprev; g.c = wk.c;
        g.dgi = wk.dgi; g.dgh = dgh;
        const dim3 grid(cdiv(g.R, 16), H / 16);
        if (rnn_type == RNN_GRU) {
            if (wide) rnn_bwd_step_kernel<3, 8><<<grid, 512, 0, stream>>>(g);
            else rnn_bwd_step_kernel<3, 4><<<grid, 256, 0, stream>>>(g);
        } else {
            if (wide) rnn_bwd_step_kernel<4, 8><<<grid, 512, 0, stream>>>(g);
            else rnn_bwd_step_kernel<4, 4><<<grid, 256, 0, stream>>>(g);
        }
        HAB_LAUNCH_CHECK();
    }
    // parameter gradients over all frames
    HAB_TRY(linear_wgrad(wk.dgi, G * H, x, ldx, lp.dw_ih, lp.in_dim, pk.P, G * H, lp.in_dim, 0, 0, 0, ws, ws_floats, stream));
    HAB_TRY(linear_wgrad(dgh, G * H, wk.hprev, H, lp.dw_hh, H, pk.P, G * H, H, 0, 0, 0, ws, ws_floats, stream));
    HAB_TRY(colsum(wk.dgi, G * H, pk.P, G * H, lp.db_ih, 0, ws, ws_floats, stream));
    HAB_TRY(colsum(dgh, G * H, pk.P, G * H, lp.db_hh, 0, ws, ws_floats, stream));
    if (dx) HAB_TRY(linear_dgrad(wk.dgi, G * H, lp.w_ih, lp.in_dim, dx_mask, ldmask, mask_cols, dx, lddx, pk.P, lp.in_dim, G * H, 0, ws, ws_floats, stream));
    return HAB_OK;
}

// BPTT of the L packed layers as a wavefront (the mirror of rnn_seq_wave_forward): launch w runs, for every layer at once, step
// max_len - 1 - (w - (L - 1 - l)) of layer l.  The top layer reads `dout` (the heads' gradient wrt the features); a layer below computes
// the gradient wrt its output inside its step kernel from the layer above's dgi of the same step (BwdStepArgs::up_dgi) -- the data
// gradient contraction of the upper layers is gone.  Then the parameter gradients of every layer and layer 0's data gradient as in
// rnn_seq_layer_backward.  w_ih_t[l]: W_ih of layer l transposed, [H][G * H] (l >= 1).  scratch: L * 2 * F * H floats.
// Returns 1 when the form does not apply.
int rnn_seq_wave_backward(int rnn_type, int H, int L, const RnnLayerParams* lp, const float* const* w_ih_t, const RnnWork* wk, const float* x0,
                          int ldx0, const float* dout, float* dx0, int lddx0, const float* dx_mask, int ldmask, int mask_cols,
                          const PackInfo& pk, float* scratch, float* ws, size_t ws_floats, hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    if (L < 2 || L > RNN_MAX_WAVE_LAYERS || (H % 64) || pk.max_len < 1) return 1;
    for (int l = 1; l < L; ++l)
        if (!w_ih_t[l] || lp[l].in_dim != H || (reinterpret_cast<uintptr_t>(w_ih_t[l]) & 15)) return 1;
    const int K = G * H;
    const bool wide = K % 128 == 0;
    for (int w = 0; w < pk.max_len + L - 1; ++w) {
        BwdStepArgsN n{};
        int rmax = 0;
        for (int l = 0; l < L; ++l) {
            BwdStepArgs& g = n.l[l];
            const int s = pk.max_len - 1 - (w - (L - 1 - l));
            g.H = H; g.K = K;
            g.R = (s >= 0 && s < pk.max_len) ? pk.num_seqs_at_step[s] : 0;
            if (g.R <= 0) { g.R = 0; continue; }
            rmax = std::max(rmax, g.R);
            g.R_next = (s + 1 < pk.max_len) ? pk.num_seqs_at_step[s + 1] : 0;
            g.idx = pk.select_inds + pk.step_offsets[s];
            g.idx_next = g.R_next ? pk.select_inds + pk.step_offsets[s + 1] : g.idx;
            g.next_keep = nullptr;
            g.dout = (l == L - 1) ? dout : nullptr;
            g.w_hh_t = lp[l].w_hh_t;
            g.dh_direct = scratch + (size_t)l * 2 * pk.F * H;
            g.dc_carry = g.dh_direct + (size_t)pk.F * H;
            g.gates = wk[l].gates; g.hn = wk[l].hn; g.hprev = wk[l].hprev; g.cprev = wk[l].cprev; g.c = wk[l].c;
            g.dgi = wk[l].dgi; g.dgh = (rnn_type == RNN_GRU) ? wk[l].dgh : wk[l].dgi;
            if (l < L - 1) { g.up_dgi = wk[l + 1].dgi; g.up_w_ih_t = w_ih_t[l + 1]; }
        }
        if (rmax == 0) continue;
        const dim3 grid(cdiv(rmax, 16), H / 16, L);
        if (rnn_type == RNN_GRU) {
            if (wide) rnn_bwd_step_wave_kernel<3, 8><<<grid, 512, 0, stream>>>(n);
            else rnn_bwd_step_wave_kernel<3, 4><<<grid, 256, 0, stream>>>(n);
        } else {
            if (wide) rnn_bwd_step_wave_kernel<4, 8><<<grid, 512, 0, stream>>>(n);
            else rnn_bwd_step_wave_kernel<4, 4><<<grid, 256, 0, stream>>>(n);
        }
        HAB_LAUNCH_CHECK();
    }
    for (int l = L - 1; l >= 0; --l) {
        const float* x = l == 0 ? x0 : wk[l - 1].out;
        const int ldx = l == 0 ? ldx0 : H;
        float* dgh = (rnn_type == RNN_GRU) ? wk[l].dgh : wk[l].dgi;
        HAB_TRY(linear_wgrad(wk[l].dgi, K, x, ldx, lp[l].dw_ih, lp[l].in_dim, pk.P, K, lp[l].in_dim, 0, 0, 0, ws, ws_floats, stream));
        HAB_TRY(linear_wgrad(dgh, K, wk[l].hprev, H, lp[l].dw_hh, H, pk.P, K, H, 0, 0, 0, ws, ws_floats, stream));
        HAB_TRY(colsum(wk[l].dgi, K, pk.P, K, lp[l].db_ih, 0, ws, ws_floats, stream));
        HAB_TRY(colsum(dgh, K, pk.P, K, lp[l].db_hh, 0, ws, ws_floats, stream));
    }
    if (dx0) HAB_TRY(linear_dgrad(wk[0].dgi, K, lp[0].w_ih, lp[0].in_dim, dx_mask, ldmask, mask_cols, dx0, lddx0, pk.P, lp[0].in_dim, K, 0, ws,
                                  ws_floats, stream));
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------------------
// TIME-MAJOR form of the same recurrence, for a regular T x n minibatch (frame f = t * n + j; RolloutStorage.data_generator's layout,
// rollout_storage.py:246).  Step t processes the n environments of time t; an episode start (masks == 0) zeroes the state OPERAND
// (h * masks, rnn_state_encoder.py:308-311) instead of opening a new packed fragment.  Per environment the chain of operations is the
// packed form's, operand for operand -- the two forms are bit-identical (tests/test_gpu_policy.py) -- but step t only needs the frames of
// times <= t, so the minibatch can be cut into time chunks and the recurrence of chunk c runs on a second stream BESIDE the encoder of
// chunk c + 1 (forward) / the data-gradient chain of chunk c + 1 (backward): the 2 x T dependent ~7 us launches of a minibatch
// (12.8 % of the C2 cycle, profiles/r02_c2_kernel_stats.txt) leave the critical path.
// ------------------------------------------------------------------------------------------------------
__global__ void gather_u8_kernel(const uint8_t* __restrict__ src, const int* __restrict__ rows, int n, uint8_t* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[rows ? rows[i] : i];
}
__global__ void iota_kernel(int* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i;
}
int rnn_tm_prepare(const uint8_t* masks, const int* rows, int B, uint8_t* frame_mask, int* iota, hipStream_t stream) {
    if (!masks || !frame_mask || !iota || B <= 0) return HAB_ERR_ARG;
    gather_u8_kernel<<<cdiv(B, 256), 256, 0, stream>>>(masks, rows, B, frame_mask);
    iota_kernel<<<cdiv(B, 256), 256, 0, stream>>>(iota, B);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// steps t0 .. t1-1 of one layer: input projection of the chunk's frames, then the recurrence.  hinit / cinit: [n][H] state entering t = 0
// (episode-start mask already applied, rnn_frag_init).  ws: split-K scratch private to the calling stream.
// The persistent form applies to H = 128 / 256 / 512 (8 waves x 1 / 2 / 4 K-chunks), <= 15 row tiles, 32-bit byte offsets into the
// per-frame arrays.  HAB_RNN_PERSIST=0 selects the step-per-launch form (bit-identical results).
static bool tm_persist_ok(int rnn_type, int H, int n, int T, const float* ws, size_t ws_floats) {
    static const bool on = hab_env_int("HAB_RNN_PERSIST", 1) != 0;
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    if (!on || (matrix_path_bits() & 4096) || !ws || ws_floats < 64) return false;
    if (H != 128 && H != 256 && H != 512) return false;
    if (cdiv(n, 16) > RNNP_MAX_ROW_TILES) return false;
    if ((size_t)T * n * G * H * 4 >= ((size_t)1 << 31)) return false;
    return true;
}

int rnn_tm_layer_forward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx, const float* hinit,
                         const float* cinit, const uint8_t* frame_mask, int n, int T_total, int t0, int t1, float* ws, size_t ws_floats,
                         hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    const size_t f0 = (size_t)t0 * n;
    HAB_TRY(linear_fwd(x + f0 * ldx, ldx, lp.w_ih, lp.in_dim, lp.b_ih, wk.gi + f0 * G * H, G * H, (t1 - t0) * n, G * H, lp.in_dim, 0, 0, ws,
                       ws_floats, stream));
    if (tm_persist_ok(rnn_type, H, n, T_total, ws, ws_floats) && wk.gates) {  // the chunk's steps as ONE persistent launch (rnn_persist.h)
        TmPersistFwdArgs a;
        a.n = n; a.H = H; a.T = T_total; a.t0 = t0; a.t1 = t1;
        a.hinit = hinit; a.cinit = cinit; a.frame_mask = frame_mask; a.gi = wk.gi; a.w_hh = lp.w_hh; a.b_hh = lp.b_hh;
        a.gates = wk.gates; a.hn = wk.hn; a.hprev = wk.hprev; a.cprev = wk.cprev; a.c = wk.c; a.out = wk.out;
        a.counters = reinterpret_cast<rnnp_u64*>(ws);
        if (hipMemsetAsync(ws, 0, 16 * sizeof(rnnp_u64), stream) != hipSuccess) return HAB_ERR_ARG;
        const dim3 grid(H / 16, cdiv(n, 16));  // row tile = slow dimension: rnn_persist.h (residency)
        const int kch = H / 128;
        if (rnn_type == RNN_GRU) {
            if (kch == 4) rnn_tm_persist_fwd_kernel<3, 4><<<grid, 512, 0, stream>>>(a);
            else if (kch == 2) rnn_tm_persist_fwd_kernel<3, 2><<<grid, 512, 0, stream>>>(a);
            else rnn_tm_persist_fwd_kernel<3, 1><<<grid, 512, 0, stream>>>(a);
        } else {
            if (kch == 4) rnn_tm_persist_fwd_kernel<4, 4><<<grid, 512, 0, stream>>>(a);
            else if (kch == 2) rnn_tm_persist_fwd_kernel<4, 2><<<grid, 512, 0, stream>>>(a);
            else rnn_tm_persist_fwd_kernel<4, 1><<<grid, 512, 0, stream>>>(a);
        }
        HAB_LAUNCH_CHECK();
        return HAB_OK;
    }
    for (int t = t0; t < t1; ++t) {
        const size_t f = (size_t)t * n, fp = (size_t)(t - 1) * n;
        StepArgs a;
        a.R = n; a.H = H;
        if (t == 0) {
            a.hp_base = hinit; a.cp_base = cinit; a.row_mask = nullptr;
        } else {
            a.hp_base = wk.out + fp * H; a.cp_base = wk.c + fp * H; a.row_mask = frame_mask + f;
        }
        a.hp_idx = nullptr; a.hp_stride = H; a.cp_idx = nullptr; a.cp_stride = H;
        a.out_idx = nullptr;  // row q = frame f + q: every per-frame array is passed at the step's first frame
        a.gi = wk.gi + f * G * H; a.w_hh = lp.w_hh; a.b_hh = lp.b_hh;
        a.gates = wk.gates + f * G * H; a.hn = wk.hn ? wk.hn + f * H : nullptr; a.hprev = wk.hprev + f * H;
        a.cprev = wk.cprev ? wk.cprev + f * H : nullptr; a.c = wk.c ? wk.c + f * H : nullptr;
        a.out = wk.out + f * H; a.out_stride = H; a.c_out = nullptr; a.c_out_stride = 0;
        HAB_TRY(launch_step(rnn_type, a, stream));
    }
    return HAB_OK;
}

// BPTT steps t1-1 .. t0 of one layer (chunks must be walked from the last to the first), then the chunk's gradient wrt the layer input.
// scratch: [2][n][H] carries, kept between the chunk calls of one backward.
int rnn_tm_layer_backward(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* dout, float* dx, int lddx,
                          const float* dx_mask, int ldmask, int mask_cols, const uint8_t* frame_mask, const int* iota, int n, int T, int t0,
                          int t1, float* scratch, float* ws, size_t ws_floats, hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    if (H % 64) return HAB_ERR_UNSUPPORTED;
    float* dh_direct = scratch;
    float* dc_carry = scratch + (size_t)n * H;
    float* dgh = (rnn_type == RNN_GRU) ? wk.dgh : wk.dgi;
    const int K = G * H;
    const bool wide = K % 128 == 0;
    bool persisted = false;
    if (tm_persist_ok(rnn_type, H, n, T, ws, ws_floats) && dout) {  // the chunk's BPTT steps as ONE persistent launch (rnn_persist.h)
        TmPersistBwdArgs a;
        a.n = n; a.H = H; a.T = T; a.t0 = t0; a.t1 = t1;
        a.frame_mask = frame_mask; a.dout = dout; a.w_hh_t = lp.w_hh_t; a.dh_direct = dh_direct; a.dc_carry = dc_carry;
        a.gates = wk.gates; a.hn = wk.hn; a.hprev = wk.hprev; a.cprev = wk.cprev; a.c = wk.c; a.dgi = wk.dgi; a.dgh = dgh;
        a.counters = reinterpret_cast<rnnp_u64*>(ws);
        if (hipMemsetAsync(ws, 0, 16 * sizeof(rnnp_u64), stream) != hipSuccess) return HAB_ERR_ARG;
        const dim3 grid(H / 16, cdiv(n, 16));  // row tile = slow dimension: rnn_persist.h (residency)
        const int kch = H / 128;
        if (rnn_type == RNN_GRU) {
            if (kch == 4) rnn_tm_persist_bwd_kernel<3, 12><<<grid, 512, 0, stream>>>(a);
            else if (kch == 2) rnn_tm_persist_bwd_kernel<3, 6><<<grid, 512, 0, stream>>>(a);
            else rnn_tm_persist_bwd_kernel<3, 3><<<grid, 512, 0, stream>>>(a);
        } else {
            if (kch == 4) rnn_tm_persist_bwd_kernel<4, 16><<<grid, 512, 0, stream>>>(a);
            else if (kch == 2) rnn_tm_persist_bwd_kernel<4, 8><<<grid, 512, 0, stream>>>(a);
            else rnn_tm_persist_bwd_kernel<4, 4><<<grid, 512, 0, stream>>>(a);
        }
        HAB_LAUNCH_CHECK();
        persisted = true;
    }
    for (int t = t1 - 1; t >= t0 && !persisted; --t) {
        BwdStepArgs g;
        g.R = n; g.H = H; g.K = K;
        g.R_next = (t + 1 < T) ? n : 0;
        g.idx = iota + (size_t)t * n;
        g.idx_next = g.R_next ? iota + (size_t)(t + 1) * n : g.idx;
        g.next_keep = g.R_next ? frame_mask + (size_t)(t + 1) * n : nullptr;
        g.dout = dout; g.w_hh_t = lp.w_hh_t; g.dh_direct = dh_direct; g.dc_carry = dc_carry;
        g.gates = wk.gates; g.hn = wk.hn; g.hprev = wk.hprev; g.cprev = wk.cprev; g.c = wk.c;
        g.dgi = wk.dgi; g.dgh = dgh;
        const dim3 grid(cdiv(g.R, 16), H / 16);
        if (rnn_type == RNN_GRU) {
            if (wide) rnn_bwd_step_kernel<3, 8><<<grid, 512, 0, stream>>>(g);
            else rnn_bwd_step_kernel<3, 4><<<grid, 256, 0, stream>>>(g);
        } else {
            if (wide) rnn_bwd_step_kernel<4, 8><<<grid, 512, 0, stream>>>(g);
            else rnn_bwd_step_kernel<4, 4><<<grid, 256, 0, stream>>>(g);
        }
        HAB_LAUNCH_CHECK();
    }
    const size_t f0 = (size_t)t0 * n;
    if (dx) HAB_TRY(linear_dgrad(wk.dgi + f0 * G * H, G * H, lp.w_ih, lp.in_dim, dx_mask ? dx_mask + f0 * ldmask : nullptr, ldmask, mask_cols,
                                 dx + f0 * lddx, lddx, (t1 - t0) * n, lp.in_dim, G * H, 0, ws, ws_floats, stream));
    return HAB_OK;
}

// parameter gradients of one layer over ALL frames (after the last chunk of rnn_tm_layer_backward)
int rnn_tm_layer_param_grads(int rnn_type, int H, const RnnLayerParams& lp, const RnnWork& wk, const float* x, int ldx, int P, float* ws,
                             size_t ws_floats, hipStream_t stream) {
    const int G = rnn_type == RNN_GRU ? 3 : 4;
    float* dgh = (rnn_type == RNN_GRU) ? wk.dgh : wk.dgi;
    HAB_TRY(linear_wgrad(wk.dgi, G * H, x, ldx, lp.dw_ih, lp.in_dim, P, G * H, lp.in_dim, 0, 0, 0, ws, ws_floats, stream));
    HAB_TRY(linear_wgrad(dgh, G * H, wk.hprev, H, lp.dw_hh, H, P, G * H, H, 0, 0, 0, ws, ws_floats, stream));
    HAB_TRY(colsum(wk.dgi, G * H, P, G * H, lp.db_ih, 0, ws, ws_floats, stream));
    HAB_TRY(colsum(dgh, G * H, P, G * H, lp.db_hh, 0, ws, ws_floats, stream));
    return HAB_OK;
}

}  // namespace hab
