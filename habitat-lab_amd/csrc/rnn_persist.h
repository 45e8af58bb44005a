// rnn_persist.h -- the time-major recurrence of one layer and one time chunk as ONE persistent launch (reference
// rl/models/rnn_state_encoder.py:318-350 walks the packed steps inside cuDNN / ATen).
//
// rnn.hip runs a step per launch: T dependent ~14 us kernels per layer whose duration is launch ramp + index / weight round trips to L2,
// not arithmetic.  That chain is the critical path of the SimpleCNN policy's backward phase and the exposed tail of its forward phase
// (NOTEBOOK R6).  Here a workgroup (row tile of 16 environments, 16 hidden units) stays resident for all steps of the chunk:
//   * its slice of W_hh (forward) / W_hh^T (BPTT) lives in REGISTERS for the whole launch: wave w of 8 holds the K range [w K/8, (w+1) K/8)
//     of the 16 x G rows as MFMA B fragments (48 - 64 VGPRs);
//   * the state entering a step is the only cross-workgroup traffic: 16 x 16 floats per workgroup and step, written write-through (sc1),
//     announced by one agent-scope atomic per workgroup on the row tile's counter, read back with sc1 loads by the H / 16 workgroups of the
//     row tile once the counter says all of them have arrived (cdna guide, Guideline 16 R1: private per-XCD L2s, L1 never refreshed);
//     row tiles never talk to each other;
//   * a thread keeps its own (row, unit) element of h (and c, and the BPTT carries) in a register from step to step;
//   * everything that does not depend on the recurrence (input projection, saved gates, masks) is loaded BEFORE the wait on the counter.
// Arithmetic, K order per wave and the cross-wave summation order are exactly rnn_step_kernel's / rnn_bwd_step_kernel's: results are
// bit-identical to the per-step launches (tests/test_gpu_rnn_persist.py), so every parity figure of the step kernels carries over.
// Residency: the H / 16 workgroups of a row tile wait for each other, so they must be resident together.  The row tile is the SLOW grid
// dimension: workgroups are dispatched in linear order, so every row tile that has a workgroup on the chip has all of its earlier
// workgroups there too, and the first incomplete row tile is completed by the very next free slots -- a launch larger than the chip
// (LSTM, H = 512: 154 registers = one workgroup per CU = 256 at a time; 15 row tiles are 480) runs its row tiles in rounds instead of
// dead-locking on half-resident ones (with the row tile as the fast dimension it would: every tile gets its first units, none its last).
// Spins are bounded all the same (RNNP_SPIN_LIMIT polls, seconds): a workgroup that gives up raises the error word behind the counters and
// traps -- the launch fails at the next synchronisation instead of hanging the device or returning unfinished outputs.
#pragma once
#include "ops.h"
#include "bf3_split.h"  // u32x4
#include "rnn_gates.h"

namespace hab {

constexpr unsigned RNNP_SPIN_LIMIT = 1u << 26;
constexpr int RNNP_MAX_ROW_TILES = 15;  // counters[0 .. 14], counters[15] = error word   (ws scratch: 16 x 8 bytes)
typedef unsigned long long rnnp_u64;

struct TmPersistFwdArgs {
    int n, H, T, t0, t1;
    const float* hinit; const float* cinit;   // [n][H]: state entering step 0 (episode-start mask already applied)
    const uint8_t* frame_mask;                // [T * n]
    const float* gi;                          // [T * n][G * H]
    const float* w_hh; const float* b_hh;     // [G * H][H], [G * H]
    float* gates; float* hn; float* hprev; float* cprev; float* c; float* out;   // per-frame arrays, bases at frame 0
    rnnp_u64* counters;
};

__device__ __forceinline__ bool rnnp_wait(rnnp_u64* ctr, rnnp_u64 target, rnnp_u64* err) {
    // thread 0 polls ONE word relaxed at agent scope; the caller's __syncthreads() releases the workgroup
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > RNNP_SPIN_LIMIT) {  // cannot happen with in-order dispatch (see the header); if it does, fail the launch loudly
            __hip_atomic_store(err, (rnnp_u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_trap();
        }
    }
    return true;
}

// Forward.  grid = (H / 16, row tiles), 512 threads.  KCH = (H / 8) / 16 K-chunks per wave (H = 128 KCH).
template <int G, int KCH>
__global__ void __launch_bounds__(512) rnn_tm_persist_fwd_kernel(const TmPersistFwdArgs a) {
    constexpr int NW = 8;
    __shared__ float red[NW][G][256];
    __shared__ int s_ok;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rt = blockIdx.y, row0 = rt * 16, u0 = blockIdx.x * 16;
    const int i = lane & 15, kg = lane >> 4;
    const int H = a.H, n = a.n;
    const int tiles = H / 16;
    const int kb = wave * (H / NW);
    const int q = min(row0 + i, n - 1);   // MFMA A row of this lane (clamped like the step kernel's)
    // W_hh slice of this wave as B fragments: row g H + u0 + i, k = kb + 16 c + 4 kg .. + 3
    f32x4 wreg[KCH][G];
#pragma unroll
    for (int c = 0; c < KCH; ++c)
#pragma unroll
        for (int g = 0; g < G; ++g) wreg[c][g] = *reinterpret_cast<const f32x4*>(a.w_hh + (size_t)(g * H + u0 + i) * H + kb + 16 * c + 4 * kg);
    // gate thread (t < 256): element (row r_, unit u_)
    const int r_ = t >> 4, u_ = t & 15, qq = row0 + r_, uu = u0 + u_;
    const bool gate_thread = (t < 256) & (qq < n);
    float bhh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bhh[g] = gate_thread ? a.b_hh[g * H + uu] : 0.f;
    float h_last = 0.f, c_last = 0.f;   // this thread's element of the state the previous step of THIS launch produced
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((size_t)a.T * n * H * 4), 0x00020000);
    rnnp_u64* ctr = a.counters + rt;
    rnnp_u64* err = a.counters + RNNP_MAX_ROW_TILES;

    for (int ts = a.t0; ts < a.t1; ++ts) {
        const size_t f = (size_t)ts * n, fp = (size_t)(ts - 1) * n;
        const bool first = ts == a.t0;
        // ---- loads that do not depend on the recurrence ----
        float gi_pre[G];
        float keep_g = 1.f, keep_q = 1.f;
        if (ts > 0) keep_q = a.frame_mask[f + q] ? 1.f : 0.f;
        if (gate_thread) {
            if (ts > 0) keep_g = a.frame_mask[f + qq] ? 1.f : 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) gi_pre[g] = a.gi[(f + qq) * G * H + g * H + uu];
        }
        // ---- the state entering the step ----
        f32x4 av[KCH];
        float hp = 0.f, cp = 0.f;
        if (first) {   // produced by an earlier launch (or the init kernel): plain loads
            const float* hrow = ts == 0 ? a.hinit + (size_t)q * H : a.out + (fp + q) * H;
#pragma unroll
            for (int c = 0; c < KCH; ++c) av[c] = *reinterpret_cast<const f32x4*>(hrow + kb + 16 * c + 4 * kg) * keep_q;
            if (gate_thread) {
                hp = (ts == 0 ? a.hinit[(size_t)qq * H + uu] : a.out[(fp + qq) * H + uu]);
                if (keep_g == 0.f) hp = 0.f;
                if constexpr (G == 4) {
                    cp = (ts == 0 ? a.cinit[(size_t)qq * H + uu] : a.c[(fp + qq) * H + uu]);
                    if (keep_g == 0.f) cp = 0.f;
                }
            }
        } else {       // produced by the other workgroups of this row tile in the previous iteration
            if (t == 0) s_ok = rnnp_wait(ctr, (rnnp_u64)tiles * (rnnp_u64)(ts - a.t0), err) ? 1 : 0;
            __syncthreads();
            if (!s_ok) return;
            const int voff = (int)(((fp + q) * H + kb + 4 * kg) * 4);
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rout, voff, 64 * c, 16);   // sc1: past the L1, write-through data
                av[c] = __builtin_bit_cast(f32x4, raw) * keep_q;
            }
            hp = keep_g == 0.f ? 0.f : h_last;
            if constexpr (G == 4) cp = keep_g == 0.f ? 0.f : c_last;
        }
        // ---- gh tile = h W_hh^T: this wave's K range, chunks in ascending k (rnn_step_body's order) ----
        f32x4 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g][0] = 0.f; acc[g][1] = 0.f; acc[g][2] = 0.f; acc[g][3] = 0.f; }
#pragma unroll
        for (int c = 0; c < KCH; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][s], wreg[c][g][s], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[wave][g][(kg * 4 + v) * 16 + i] = acc[g][v];
        __syncthreads();
        if (gate_thread) {
            float gh[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w += 4) sum += (red[w][g][t] + red[w + 1][g][t]) + (red[w + 2][g][t] + red[w + 3][g][t]);
                gh[g] = sum + bhh[g];
            }
            const size_t fo = (f + qq) * H + uu;
            float hnew;
            if constexpr (G == 3) {
                float rg, zg, ng;
                hnew = gru_cell_fwd(gi_pre[0], gi_pre[1], gi_pre[2], gh[0], gh[1], gh[2], hp, rg, zg, ng);
                float* gs = a.gates + (f + qq) * 3 * H;
                gs[uu] = rg; gs[H + uu] = zg; gs[2 * H + uu] = ng;
                a.hn[fo] = gh[2];
                a.hprev[fo] = hp;
            } else {
                float ig, fg, gg, og, cn;
                hnew = lstm_cell_fwd(gi_pre[0] + gh[0], gi_pre[1] + gh[1], gi_pre[2] + gh[2], gi_pre[3] + gh[3], cp, ig, fg, gg, og, cn);
                float* gs = a.gates + (f + qq) * 4 * H;
                gs[uu] = ig; gs[H + uu] = fg; gs[2 * H + uu] = gg; gs[3 * H + uu] = og;
                a.hprev[fo] = hp;
                a.cprev[fo] = cp;
                a.c[fo] = cn;
                c_last = cn;
            }
            h_last = hnew;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hnew), rout, (int)(fo * 4), 0, 16);   // write-through: the next step's operand
        }
        if (ts + 1 < a.t1) {   // publish: every storing wave drains, then ONE lane arrives
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // (also: `red` is free for the next step)
            if (t == 0) __hip_atomic_fetch_add(ctr, (rnnp_u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct TmPersistBwdArgs {
    int n, H, T, t0, t1;
    const uint8_t* frame_mask;      // [T * n]
    const float* dout;              // [T * n][H]
    const float* w_hh_t;            // [H][G * H]
    float* dh_direct; float* dc_carry;   // [n][H] carries between chunk launches (GRU: direct term, LSTM: cell carry)
    const float* gates; const float* hn; const float* hprev; const float* cprev; const float* c;
    float* dgi; float* dgh;         // [T * n][G * H]; LSTM: dgh == dgi
    rnnp_u64* counters;
};

// BPTT.  grid = (row tiles, H / 16), 512 threads.  KCB = (G H / 8) / 16 K-chunks per wave.
template <int G, int KCB>
__global__ void __launch_bounds__(512) rnn_tm_persist_bwd_kernel(const TmPersistBwdArgs a) {
    constexpr int NW = 8;
    __shared__ float red[NW][256];
    __shared__ int s_ok;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rt = blockIdx.y, row0 = rt * 16, u0 = blockIdx.x * 16;
    const int i = lane & 15, kg = lane >> 4;
    const int H = a.H, n = a.n, K = G * H;
    const int tiles = H / 16;
    const int kb = wave * (K / NW);
    const int q = min(row0 + i, n - 1);
    f32x4 wreg[KCB];   // W_hh^T row u0 + i, k = kb + 16 c + 4 kg .. + 3
#pragma unroll
    for (int c = 0; c < KCB; ++c) wreg[c] = *reinterpret_cast<const f32x4*>(a.w_hh_t + (size_t)(u0 + i) * K + kb + 16 * c + 4 * kg);
    const int r_ = t >> 4, uu = u0 + (t & 15), qq = row0 + r_;
    const bool gate_thread = (t < 256) & (qq < n);
    const size_t qo = (size_t)qq * H + uu;
    // carries of this thread's element: from the previous chunk launch (valid when a later step exists), kept in a register afterwards
    float carry = 0.f;
    if (gate_thread && a.t1 < a.T) carry = (G == 3) ? a.dh_direct[qo] : a.dc_carry[qo];
    const __amdgpu_buffer_rsrc_t rdgh = __builtin_amdgcn_make_buffer_rsrc(a.dgh, 0, (int)((size_t)a.T * n * K * 4), 0x00020000);
    rnnp_u64* ctr = a.counters + rt;
    rnnp_u64* err = a.counters + RNNP_MAX_ROW_TILES;

    for (int ts = a.t1 - 1; ts >= a.t0; --ts) {
        const size_t f = (size_t)ts * n, fn = (size_t)(ts + 1) * n;
        const bool first = ts == a.t1 - 1;
        const bool has_next = ts + 1 < a.T;
        // ---- loads that do not depend on the recurrence ----
        float dout_pre = 0.f, gs_pre[G], hp_pre = 0.f, hn_pre = 0.f, cn_pre = 0.f, cp_pre = 0.f;
        bool carried = false;
#pragma unroll
        for (int g = 0; g < G; ++g) gs_pre[g] = 0.f;
        if (gate_thread) {
            const size_t fo = (f + qq) * H + uu;
            dout_pre = a.dout[fo];
            carried = has_next && a.frame_mask[fn + qq] != 0;
#pragma unroll
            for (int g = 0; g < G; ++g) gs_pre[g] = a.gates[(f + qq) * G * H + g * H + uu];
            if constexpr (G == 3) { hp_pre = a.hprev[fo]; hn_pre = a.hn[fo]; }
            else { cn_pre = a.c[fo]; cp_pre = a.cprev[fo]; }
        }
        // ---- carry tile = dgh[step ts + 1] W_hh: this wave's K range, chunks in ascending k, one accumulator (rnn_bwd_step_body's order) ----
        if (has_next) {
            f32x4 av[KCB];
            if (first) {   // step ts + 1 belongs to the previous chunk launch: plain loads
                const float* arow = a.dgh + (fn + q) * K;
#pragma unroll
                for (int c = 0; c < KCB; ++c) av[c] = *reinterpret_cast<const f32x4*>(arow + kb + 16 * c + 4 * kg);
            } else {
                if (t == 0) s_ok = rnnp_wait(ctr, (rnnp_u64)tiles * (rnnp_u64)(a.t1 - 1 - ts), err) ? 1 : 0;
                __syncthreads();
                if (!s_ok) return;
                const int voff = (int)(((fn + q) * K + kb + 4 * kg) * 4);
#pragma unroll
                for (int c = 0; c < KCB; ++c) av[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdgh, voff, 64 * c, 16));
            }
            f32x4 acc;
            acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f; acc[3] = 0.f;
#pragma unroll
            for (int c = 0; c < KCB; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][s], wreg[c][s], acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) red[wave][(kg * 4 + v) * 16 + i] = acc[v];
            __syncthreads();
        }
        if (gate_thread) {
            float dh = dout_pre;
            if (carried) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w += 4) sum += (red[w][t] + red[w + 1][t]) + (red[w + 2][t] + red[w + 3][t]);
                dh += sum;
                if constexpr (G == 3) dh += carry;
            }
            const size_t go = (f + qq) * K;
            if constexpr (G == 3) {
                float dr_pre, dz_pre, dn_pre, dhn_pre, direct;
                gru_cell_bwd(dh, gs_pre[0], gs_pre[1], gs_pre[2], hp_pre, hn_pre, dr_pre, dz_pre, dn_pre, dhn_pre, direct);
                float* gi = a.dgi + go;
                gi[uu] = dr_pre; gi[H + uu] = dz_pre; gi[2 * H + uu] = dn_pre;
                // dgh is the next iteration's operand of every workgroup of the row tile: write-through
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dr_pre), rdgh, (int)((go + uu) * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz_pre), rdgh, (int)((go + H + uu) * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dhn_pre), rdgh, (int)((go + 2 * H + uu) * 4), 0, 16);
                carry = direct;
            } else {
                float di_pre, df_pre, dg_pre, do_pre, cc;
                lstm_cell_bwd(dh, carried, carry, gs_pre[0], gs_pre[1], gs_pre[2], gs_pre[3], cn_pre, cp_pre, di_pre, df_pre, dg_pre, do_pre, cc);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(di_pre), rdgh, (int)((go + uu) * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(df_pre), rdgh, (int)((go + H + uu) * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg_pre), rdgh, (int)((go + 2 * H + uu) * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(do_pre), rdgh, (int)((go + 3 * H + uu) * 4), 0, 16);
                carry = cc;
            }
        }
        if (ts > a.t0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(ctr, (rnnp_u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (gate_thread) {   // for the next (earlier) chunk launch
        if constexpr (G == 3) a.dh_direct[qo] = carry;
        else a.dc_carry[qo] = carry;
    }
}

}  // namespace hab
