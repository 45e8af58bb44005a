// RL-side kernels of the DD-PPO hot path for gfx950: synthetic observation source, GAE,
// advantage statistics, fused PPO loss fwd+bwd, global-norm clip + Adam, action sampling.
// All of these are HBM/latency-bound fp32 work (no MFMA): coalesced (T+1,N,*) row accesses,
// wave shuffles for the reductions, one launch each.
#include "hab_common.h"
#include "../../include/habitat_amd.h"

using namespace hab;

// ------------------------------------------------------------------------------------------
// Synthetic PointNav observation source.  Bit-identical to oracle/synth.py.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t stream_key(uint32_t seed, uint32_t sensor, uint32_t env, uint32_t t) {
    uint32_t h = mix32(seed + 0x9E3779B9u * (sensor + 1u));
    h = mix32(h ^ env);
    return mix32(h ^ t);
}
__device__ inline float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }

// grid: (blocks_per_env, N).  Writes rgb (u8, H*W*3 bytes as whole words) and depth (f32, H*W).
__global__ void __launch_bounds__(256) synth_images_kernel(uint8_t* __restrict__ rgb, float* __restrict__ depth,
                                                           const int64_t* __restrict__ env_t, uint32_t seed,
                                                           uint32_t env_offset, int rgb_words, int depth_words) {
    const int n = blockIdx.y;
    const uint32_t t = (uint32_t)env_t[n];
    const uint32_t env = env_offset + (uint32_t)n;
    if (rgb) {
        const uint32_t key = stream_key(seed, 0u, env, t);
        uint4* dst = reinterpret_cast<uint4*>(rgb + (size_t)n * rgb_words * 4);
        const int nv = rgb_words >> 2;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
            const uint32_t b = (uint32_t)i * 4u;
            dst[i] = make_uint4(mix32(key ^ b), mix32(key ^ (b + 1)), mix32(key ^ (b + 2)), mix32(key ^ (b + 3)));
        }
        for (int i = (nv << 2) + blockIdx.x * blockDim.x + threadIdx.x; i < rgb_words; i += gridDim.x * blockDim.x)
            reinterpret_cast<uint32_t*>(rgb + (size_t)n * rgb_words * 4)[i] = mix32(key ^ (uint32_t)i);
    }
    if (depth) {
        const uint32_t key = stream_key(seed, 1u, env, t);
        float4* dst = reinterpret_cast<float4*>(depth + (size_t)n * depth_words);
        const int nv = depth_words >> 2;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
            const uint32_t b = (uint32_t)i * 4u;
            dst[i] = make_float4(u01(mix32(key ^ b)), u01(mix32(key ^ (b + 1))), u01(mix32(key ^ (b + 2))),
                                 u01(mix32(key ^ (b + 3))));
        }
        for (int i = (nv << 2) + blockIdx.x * blockDim.x + threadIdx.x; i < depth_words; i += gridDim.x * blockDim.x)
            depth[(size_t)n * depth_words + i] = u01(mix32(key ^ (uint32_t)i));
    }
}

// One thread per env: advance the env clock, then goal / reward / done for the new step.
__global__ void synth_scalars_kernel(float* __restrict__ goal, float* __restrict__ reward, uint8_t* __restrict__ not_done,
                                     int64_t* __restrict__ env_t, int64_t* __restrict__ since_reset, uint32_t seed,
                                     uint32_t env_offset, int N, int advance) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int64_t t64 = env_t[n];
    if (advance) { t64 += 1; env_t[n] = t64; }
    const uint32_t t = (uint32_t)t64, env = env_offset + (uint32_t)n;
    {
        const uint32_t key = stream_key(seed, 2u, env, t);
        const float u0 = u01(mix32(key ^ 0u)), u1 = u01(mix32(key ^ 1u));
        if (goal) {
            goal[2 * n + 0] = __fmul_rn(u0, 10.0f);
            goal[2 * n + 1] = __fmul_rn(__fadd_rn(__fmul_rn(u1, 2.0f), -1.0f), 3.14159274101257324f);
        }
    }
    if (advance) {
        const uint32_t key = stream_key(seed, 3u, env, t);
        const float u0 = u01(mix32(key ^ 0u)), u1 = u01(mix32(key ^ 1u)), u2 = u01(mix32(key ^ 2u)),
                    u3 = u01(mix32(key ^ 3u));
        const float s = __fadd_rn(__fadd_rn(u0, u1), __fadd_rn(u2, u3));
        reward[n] = __fmul_rn(__fadd_rn(s, -2.0f), 1.73205077648162842f);
        const uint32_t d = mix32(stream_key(seed, 4u, env, t) ^ 0u);
        int64_t s2 = since_reset[n] + 1;
        const bool done = (d < 171798691u) || (s2 >= 500);
        since_reset[n] = done ? 0 : s2;
        not_done[n] = done ? 0 : 1;
    }
}

extern "C" int hab_synth_step(uint8_t* rgb, float* depth, float* goal, float* reward, uint8_t* not_done, int64_t* env_t,
                              int64_t* since_reset, uint32_t seed, uint32_t env_offset, int N, int H, int W, int advance,
                              hipStream_t stream) {
    if (N <= 0 || H <= 0 || W <= 0 || !env_t) return HAB_ERR_ARG;
    if (advance && (!reward || !not_done || !since_reset)) return HAB_ERR_ARG;
    if ((H * W * 3) % 4 != 0) return HAB_ERR_UNSUPPORTED;
    synth_scalars_kernel<<<cdiv(N, 64), 64, 0, stream>>>(goal, reward, not_done, env_t, since_reset, seed, env_offset, N,
                                                         advance);
    HAB_LAUNCH_CHECK();
    if (rgb || depth) {
        const int rgb_words = H * W * 3 / 4, depth_words = H * W;
        dim3 grid(cdiv(depth_words / 4, 256), N);
        if (grid.x > 64) grid.x = 64;
        synth_images_kernel<<<grid, 256, 0, stream>>>(rgb, depth, env_t, seed, env_offset, rgb_words, depth_words);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

// Per-step episode bookkeeping of the rollout loop (ppo_trainer.py:417-446) + the prev_actions write of
// RolloutStorage.insert (rollout_storage.py:124-130), one launch instead of eight elementwise ones.
__global__ void rollout_step_stats_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ not_done,
                                          float* __restrict__ cur, float* __restrict__ stat_reward, float* __restrict__ stat_count,
                                          const int64_t* __restrict__ actions, int64_t* __restrict__ prev_next, int N, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float c = cur[i] + rewards[i];          // current_episode_reward += rewards
    if (!not_done[i]) {
        stat_reward[i] += c;                // running_episode_stats["reward"] += current_episode_reward.where(done, 0)
        stat_count[i] += 1.0f;              // running_episode_stats["count"] += done
        c = 0.0f;                           // current_episode_reward.masked_fill_(done, 0)
    }
    cur[i] = c;
    if (actions)
        for (int a = 0; a < A; ++a) prev_next[(size_t)i * A + a] = actions[(size_t)i * A + a];
}
extern "C" int hab_rollout_step_stats(const float* rewards, const uint8_t* not_done, float* current_episode_reward, float* stat_reward,
                                      float* stat_count, const int64_t* actions, int64_t* prev_actions_next, int N, int action_dim,
                                      hipStream_t stream) {
    if (!rewards || !not_done || !current_episode_reward || !stat_reward || !stat_count || N <= 0) return HAB_ERR_ARG;
    if ((actions == nullptr) != (prev_actions_next == nullptr) || (actions && action_dim <= 0)) return HAB_ERR_ARG;
    rollout_step_stats_kernel<<<cdiv(N, 64), 64, 0, stream>>>(rewards, not_done, current_episode_reward, stat_reward, stat_count,
                                                              actions, prev_actions_next, N, action_dim);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ObjectNav sensor set for the current env clock (oracle/synth.py: semantic, objectgoal, compass, gps).
__global__ void __launch_bounds__(256) synth_semantic_kernel(int32_t* __restrict__ semantic, const int64_t* __restrict__ env_t,
                                                             uint32_t seed, uint32_t env_offset, int words) {
    const int n = blockIdx.y;
    const uint32_t key = stream_key(seed, 5u, env_offset + (uint32_t)n, (uint32_t)env_t[n]);
    int32_t* dst = semantic + (size_t)n * words;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) dst[i] = (int32_t)(mix32(key ^ (uint32_t)i) % 40u);
}
__global__ void synth_objectnav_scalars_kernel(int64_t* __restrict__ objectgoal, float* __restrict__ compass, float* __restrict__ gps,
                                               const int64_t* __restrict__ env_t, uint32_t seed, uint32_t env_offset, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t t = (uint32_t)env_t[n], env = env_offset + (uint32_t)n;
    if (objectgoal) objectgoal[n] = (int64_t)(mix32(stream_key(seed, 6u, env, 0u) ^ 0u) % 21u);
    if (compass) {
        const float u = u01(mix32(stream_key(seed, 7u, env, t) ^ 0u));
        compass[n] = __fmul_rn(__fadd_rn(__fmul_rn(u, 2.0f), -1.0f), 3.14159274101257324f);
    }
    if (gps) {
        const uint32_t key = stream_key(seed, 8u, env, t);
        float u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = u01(mix32(key ^ (uint32_t)i));
        const float a = __fadd_rn(__fadd_rn(u[0], u[1]), __fadd_rn(u[2], u[3]));
        const float b = __fadd_rn(__fadd_rn(u[4], u[5]), __fadd_rn(u[6], u[7]));
        gps[2 * n + 0] = __fmul_rn(__fadd_rn(a, -2.0f), 1.73205077648162842f);
        gps[2 * n + 1] = __fmul_rn(__fadd_rn(b, -2.0f), 1.73205077648162842f);
    }
}
extern "C" int hab_synth_objectnav_sensors(int32_t* semantic, int64_t* objectgoal, float* compass, float* gps, const int64_t* env_t,
                                           uint32_t seed, uint32_t env_offset, int N, int H, int W, hipStream_t stream) {
    if (N <= 0 || H <= 0 || W <= 0 || !env_t) return HAB_ERR_ARG;
    synth_objectnav_scalars_kernel<<<cdiv(N, 64), 64, 0, stream>>>(objectgoal, compass, gps, env_t, seed, env_offset, N);
    HAB_LAUNCH_CHECK();
    if (semantic) {
        dim3 grid(min(64, cdiv(H * W, 256)), N);
        synth_semantic_kernel<<<grid, 256, 0, stream>>>(semantic, env_t, seed, env_offset, H * W);
        HAB_LAUNCH_CHECK();
    }
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// GAE / discounted returns (K13).  Buffers are (T+1, N) fp32, masks (T+1, N) u8 (1 = not done).
// Variant A (exact): one lane per env, reverse sequential recurrence in the reference's operation
// order with explicit _rn ops (no fma contraction) -> bitwise equal to the PyTorch-CPU loop.
// ------------------------------------------------------------------------------------------
__global__ void gae_exact_kernel(const float* __restrict__ rewards, float* __restrict__ value_preds,
                                 const uint8_t* __restrict__ masks, float* __restrict__ returns,
                                 const float* __restrict__ next_value, int T, int N, float gamma, float gamma_tau,
                                 int use_gae) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (use_gae) {
        float v_next = next_value[n];
        value_preds[(size_t)T * N + n] = v_next;
        float gae = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const float m = masks[(size_t)(t + 1) * N + n] ? 1.0f : 0.0f;
            const float v = value_preds[(size_t)t * N + n];
            // delta = r + gamma*v_next*m - v ; gae = delta + (gamma*tau)*gae*m ; ret = gae + v
            const float delta = __fadd_rn(__fadd_rn(rewards[(size_t)t * N + n], __fmul_rn(__fmul_rn(gamma, v_next), m)), -v);
            gae = __fadd_rn(delta, __fmul_rn(__fmul_rn(gamma_tau, gae), m));
            returns[(size_t)t * N + n] = __fadd_rn(gae, v);
            v_next = v;
        }
    } else {
        float r_next = next_value[n];
        returns[(size_t)T * N + n] = r_next;
        for (int t = T - 1; t >= 0; --t) {
            const float m = masks[(size_t)(t + 1) * N + n] ? 1.0f : 0.0f;
            r_next = __fadd_rn(__fmul_rn(__fmul_rn(gamma, r_next), m), rewards[(size_t)t * N + n]);
            returns[(size_t)t * N + n] = r_next;
        }
    }
}

// Variant B (scan): one wavefront per env.  A_t = delta_t + c_t * A_{t+1} is an affine recurrence;
// each lane owns a contiguous chunk of time steps, composes its chunk's affine map (a,b), the wave
// does a suffix scan of the maps with shuffles (6 rounds), then each lane replays its chunk.
// Different association order than the reference loop => agrees to fp32 round-off, not bitwise.
__global__ void __launch_bounds__(64) gae_scan_kernel(const float* __restrict__ rewards, float* __restrict__ value_preds,
                                                      const uint8_t* __restrict__ masks, float* __restrict__ returns,
                                                      const float* __restrict__ next_value, int T, int N, float gamma,
                                                      float gamma_tau) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int per = cdiv(T, 64);
    const int t0 = lane * per, t1 = min(T, t0 + per);  // this lane's steps [t0, t1)
    if (lane == 0) value_preds[(size_t)T * N + n] = next_value[n];
    // chunk map applied to the incoming A_{t1}:  A_{t0} = a * A_{t1} + b
    float a = 1.0f, b = 0.0f;
    for (int t = t1 - 1; t >= t0; --t) {
        const float m = masks[(size_t)(t + 1) * N + n] ? 1.0f : 0.0f;
        const float vn = (t + 1 == T) ? next_value[n] : value_preds[(size_t)(t + 1) * N + n];
        const float delta = rewards[(size_t)t * N + n] + gamma * vn * m - value_preds[(size_t)t * N + n];
        const float c = gamma_tau * m;
        b = delta + c * b;
        a = c * a;
    }
    // exclusive suffix scan over lanes: carry_in(lane) = composition of maps of lanes > lane applied to 0
    float sa = a, sb = b;  // inclusive suffix composition: maps of lanes >= lane
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float oa = __shfl_down(sa, o, 64), ob = __shfl_down(sb, o, 64);
        if (lane + o < 64) { sb = sa * ob + sb; sa = sa * oa; }
    }
    float carry = __shfl_down(sb, 1, 64);  // value of A at t1 for this lane (suffix of lanes > lane applied to 0)
    if (lane == 63) carry = 0.0f;
    float gae = carry;
    for (int t = t1 - 1; t >= t0; --t) {
        const float m = masks[(size_t)(t + 1) * N + n] ? 1.0f : 0.0f;
        const float vn = (t + 1 == T) ? next_value[n] : value_preds[(size_t)(t + 1) * N + n];
        const float v = value_preds[(size_t)t * N + n];
        const float delta = rewards[(size_t)t * N + n] + gamma * vn * m - v;
        gae = delta + gamma_tau * m * gae;
        returns[(size_t)t * N + n] = gae + v;
    }
}

extern "C" int hab_compute_returns(const float* rewards, float* value_preds, const uint8_t* masks, float* returns,
                                   const float* next_value, int T, int N, float gamma, float tau, int use_gae,
                                   int variant, hipStream_t stream) {
    if (T < 0 || N <= 0 || !rewards || !value_preds || !masks || !returns || !next_value) return HAB_ERR_ARG;
    // the reference forms gamma*tau in double (python floats) before it meets the fp32 tensor
    const float gamma_tau = (float)((double)gamma * (double)tau);
    if (variant == HAB_GAE_SCAN && use_gae && T > 0) {
        gae_scan_kernel<<<N, 64, 0, stream>>>(rewards, value_preds, masks, returns, next_value, T, N, gamma, gamma_tau);
    } else {
        gae_exact_kernel<<<cdiv(N, 64), 64, 0, stream>>>(rewards, value_preds, masks, returns, next_value, T, N, gamma,
                                                         gamma_tau, use_gae);
    }
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// Advantages (K14): adv = returns - value_preds over ALL rows (incl. the bootstrap row), then
// optional (adv - mean) * rsqrt(var + 1e-5) over the finite entries.  Single 1024-thread block:
// count <= ~1e5 elements.  stats_out = {mean, var} for the distributed variant.
// Modes (HAB_ADV_*): RAW; LOCAL_NORMALIZE (torch.var_mean, unbiased -- ppo.py:147-153);
// STATS_MEAN / STATS_VAR / EXT_NORMALIZE = the three phases of distributed_var_mean
// (ddppo.py:59-84) with the two scalar all-reduces done by the caller in between.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) advantages_kernel(const float* __restrict__ returns, const float* __restrict__ value_preds,
                                                          float* __restrict__ adv, int count, int mode,
                                                          const float* ext_stats, float* stats_out) {  // may alias (one stats buffer)
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double s = 0.0, c = 0.0;
    for (int i = tid; i < count; i += 1024) {
        const float a = __fadd_rn(returns[i], -value_preds[i]);
        adv[i] = a;
        if (isfinite(a)) { s += (double)a; c += 1.0; }
    }
    if (mode == HAB_ADV_RAW) return;
    auto block_sum = [&](double v) -> double {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        return t;
    };
    const double C = block_sum(c);
    double mean;
    if (mode == HAB_ADV_LOCAL_NORMALIZE || mode == HAB_ADV_STATS_MEAN) {
        mean = (C > 0) ? block_sum(s) / C : 0.0;
        if (mode == HAB_ADV_STATS_MEAN) {
            if (tid == 0) { stats_out[0] = (float)mean; stats_out[2] = (float)C; }
            return;
        }
    } else {
        mean = (double)ext_stats[0];
    }
    double var;
    if (mode == HAB_ADV_EXT_NORMALIZE) {
        var = (double)ext_stats[1];
    } else {
        double q = 0.0;
        for (int i = tid; i < count; i += 1024) {
            const float a = adv[i];
            if (isfinite(a)) { const double d = (double)a - mean; q += d * d; }
        }
        const double Q = block_sum(q);
        if (mode == HAB_ADV_STATS_VAR) {  // biased second moment about the supplied (global) mean
            if (tid == 0) { stats_out[1] = (float)(Q / fmax(C, 1.0)); stats_out[2] = (float)C; }
            return;
        }
        var = Q / fmax(C - 1.0, 1.0);  // torch.var_mean default: unbiased
        if (stats_out && tid == 0) { stats_out[0] = (float)mean; stats_out[1] = (float)var; stats_out[2] = (float)C; }
    }
    const float meanf = (float)mean, rstd = rsqrtf((float)var + 1e-5f);
    for (int i = tid; i < count; i += 1024) adv[i] = __fmul_rn(__fadd_rn(adv[i], -meanf), rstd);
}

extern "C" int hab_advantages(const float* returns, const float* value_preds, float* adv, int count, int mode,
                              const float* ext_stats, float* stats_out, hipStream_t stream) {
    if (count <= 0 || !returns || !value_preds || !adv || mode < 0 || mode > 4) return HAB_ERR_ARG;
    if ((mode == HAB_ADV_EXT_NORMALIZE || mode == HAB_ADV_STATS_VAR) && !ext_stats) return HAB_ERR_ARG;
    if ((mode == HAB_ADV_STATS_MEAN || mode == HAB_ADV_STATS_VAR) && !stats_out) return HAB_ERR_ARG;
    advantages_kernel<<<1, 1024, 0, stream>>>(returns, value_preds, adv, count, mode, ext_stats, stats_out);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// Fused PPO clipped-surrogate loss, forward + backward (K15).
// Per frame f (B frames, minibatch order): the new (value, log_prob, entropy) are dense [B];
// the rollout-side quantities are gathered straight from the (T+1,N) storage through rows[f]
// (rows == nullptr -> dense).  Emits dL/dvalue, dL/dlogp, dL/dentropy and 12 scalars:
//  out[0..2] value_loss, action_loss, dist_entropy (means)   out[3] total loss
//  out[4..6] value_pred min/mean/max   out[7..9] prob_ratio min/mean/max   out[10] fraction clipped
//  out[11] B.  Single 1024-thread block -> deterministic reduction order.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) ppo_loss_kernel(const float* __restrict__ values, const float* __restrict__ logp,
                                                        const float* __restrict__ entropy, const float* __restrict__ old_logp,
                                                        const float* __restrict__ adv, const float* __restrict__ old_values,
                                                        const float* __restrict__ returns, const int* __restrict__ rows, int B,
                                                        float clip, float value_coef, float entropy_coef, int clip_value,
                                                        float* __restrict__ d_value, float* __restrict__ d_logp,
                                                        float* __restrict__ d_entropy, float* __restrict__ out,
                                                        const float* __restrict__ is_coeffs, const uint8_t* __restrict__ is_stale,
                                                        const int64_t* __restrict__ policy_version, long long current_version,
                                                        const float* __restrict__ log_alpha, float entropy_threshold) {
    // adaptive entropy penalty (LagrangeInequalityCoefficient, utils/common.py:749-806, greater_than form): the coefficient is
    // alpha = exp(log_alpha), a device scalar trained by its own Adam step; loss term alpha * (threshold - [ent]) - [alpha] * ent
    if (log_alpha) entropy_coef = expf(log_alpha[0]);
    __shared__ float red[16][10];
    __shared__ float redmm[16][8];
    const int tid = threadIdx.x;
    const float invB = 1.0f / (float)B;
    float s_vl = 0, s_al = 0, s_en = 0, s_v = 0, s_r = 0, s_clip = 0, s_is = 0, s_stale = 0, s_pv = 0;
    float v_min = INFINITY, v_max = -INFINITY, r_min = INFINITY, r_max = -INFINITY;
    float is_min = INFINITY, is_max = -INFINITY, pv_min = INFINITY, pv_max = -INFINITY;
    for (int f = tid; f < B; f += 1024) {
        const int g = rows ? rows[f] : f;
        const float v = values[f], lp = logp[f], en = entropy[f];
        const float a = adv[g], ov = old_values[g], ret = returns[g];
        const float ratio = expf(lp - old_logp[g]);
        const float lo = 1.0f - clip, hi = 1.0f + clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = a * ratio, s2 = a * rc;
        const float al = -fminf(s1, s2);
        // VER importance weights (rl/ppo/ppo.py:226-231): every per-frame loss term is multiplied by min(is_coeffs, 1) before the mean
        const float isc = is_coeffs ? is_coeffs[g] : 1.0f;
        const float wf = fminf(isc, 1.0f);
        const bool inr = (ratio >= lo) && (ratio <= hi);
        // torch.min backward: ties split the gradient evenly; clamp backward passes inside [lo, hi]
        const float w = (s1 < s2) ? 1.0f : ((s1 == s2) ? (0.5f + (inr ? 0.5f : 0.0f)) : (inr ? 1.0f : 0.0f));
        d_logp[f] = -a * w * ratio * invB * wf;
        float vv = v;
        float dv = 0.0f;
        if (clip_value) {
            const float delta = v - ov;
            const bool keep = fabsf(delta) < clip;
            vv = keep ? v : (ov + fminf(fmaxf(delta, -clip), clip));
            dv = keep ? (v - ret) : 0.0f;
        } else {
            dv = v - ret;
        }
        const float e = vv - ret;
        const float vl = 0.5f * (e * e);
        d_value[f] = value_coef * dv * invB * wf;
        d_entropy[f] = -entropy_coef * invB * wf;
        s_vl += wf * vl; s_al += wf * al; s_en += wf * en; s_v += v; s_r += ratio;
        if (is_coeffs) { s_is += isc; is_min = fminf(is_min, isc); is_max = fmaxf(is_max, isc); }
        if (is_stale) s_stale += is_stale[g] ? 1.0f : 0.0f;
        if (policy_version) {
            const float d = (float)(current_version - (long long)policy_version[g]);
            s_pv += d; pv_min = fminf(pv_min, d); pv_max = fmaxf(pv_max, d);
        }
        s_clip += ((ratio > hi) ? 1.0f : 0.0f) + ((ratio < lo) ? 1.0f : 0.0f);
        v_min = fminf(v_min, v); v_max = fmaxf(v_max, v); r_min = fminf(r_min, ratio); r_max = fmaxf(r_max, ratio);
    }
    float sums[9] = {s_vl, s_al, s_en, s_v, s_r, s_clip, s_is, s_stale, s_pv};
#pragma unroll
    for (int k = 0; k < 9; ++k) sums[k] = wave_sum(sums[k]);
    v_min = wave_min(v_min); r_min = wave_min(r_min); v_max = wave_max(v_max); r_max = wave_max(r_max);
    is_min = wave_min(is_min); is_max = wave_max(is_max); pv_min = wave_min(pv_min); pv_max = wave_max(pv_max);
    const int w = tid >> 6;
    if ((tid & 63) == 0) {
        for (int k = 0; k < 9; ++k) red[w][k] = sums[k];
        redmm[w][0] = v_min; redmm[w][1] = v_max; redmm[w][2] = r_min; redmm[w][3] = r_max;
        redmm[w][4] = is_min; redmm[w][5] = is_max; redmm[w][6] = pv_min; redmm[w][7] = pv_max;
    }
    __syncthreads();
    if (tid == 0) {
        float t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        float mn_v = INFINITY, mx_v = -INFINITY, mn_r = INFINITY, mx_r = -INFINITY;
        float mn_i = INFINITY, mx_i = -INFINITY, mn_p = INFINITY, mx_p = -INFINITY;
        for (int i = 0; i < 16; ++i) {
            for (int k = 0; k < 9; ++k) t[k] += red[i][k];
            mn_v = fminf(mn_v, redmm[i][0]); mx_v = fmaxf(mx_v, redmm[i][1]);
            mn_r = fminf(mn_r, redmm[i][2]); mx_r = fmaxf(mx_r, redmm[i][3]);
            mn_i = fminf(mn_i, redmm[i][4]); mx_i = fmaxf(mx_i, redmm[i][5]);
            mn_p = fminf(mn_p, redmm[i][6]); mx_p = fmaxf(mx_p, redmm[i][7]);
        }
        if (is_coeffs || is_stale || policy_version) {  // VER learner metrics (ppo.py:262-263,285-299): out[13..19]
            out[13] = mn_i; out[14] = t[6] * invB; out[15] = mx_i; out[16] = t[7] * invB;
            out[17] = mn_p; out[18] = t[8] * invB; out[19] = mx_p;
        }
        const float vl = t[0] * invB, al = t[1] * invB, en = t[2] * invB;
        out[0] = vl; out[1] = al; out[2] = en;
        out[3] = value_coef * vl + al - entropy_coef * en;
        if (log_alpha) {
            out[3] = value_coef * vl + al + (entropy_coef * (entropy_threshold - en) - entropy_coef * en);
            out[20] = entropy_coef * (entropy_threshold - en);  // d loss / d log_alpha
            out[21] = entropy_coef;                             // learner metric `entropy_coef`
        }
        out[4] = mn_v; out[5] = t[3] * invB; out[6] = mx_v;
        out[7] = mn_r; out[8] = t[4] * invB; out[9] = mx_r;
        out[10] = t[5] * invB; out[11] = (float)B;
    }
}

extern "C" int hab_ppo_loss(const float* values, const float* logp, const float* entropy, const float* old_logp,
                            const float* adv, const float* old_values, const float* returns, const int* rows, int B,
                            float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                            float* d_value, float* d_logp, float* d_entropy, float* out12, hipStream_t stream) {
    if (B <= 0 || !values || !logp || !entropy || !old_logp || !adv || !old_values || !returns || !d_value || !d_logp ||
        !d_entropy || !out12)
        return HAB_ERR_ARG;
    ppo_loss_kernel<<<1, 1024, 0, stream>>>(values, logp, entropy, old_logp, adv, old_values, returns, rows, B, clip_param,
                                            value_loss_coef, entropy_coef, use_clipped_value_loss, d_value, d_logp,
                                            d_entropy, out12, nullptr, nullptr, nullptr, 0, nullptr, 0.f);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

extern "C" int hab_ppo_loss_ver(const float* values, const float* logp, const float* entropy, const float* old_logp,
                                const float* adv, const float* old_values, const float* returns, const int* rows, int B,
                                float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                                const float* is_coeffs, const uint8_t* is_stale, const int64_t* policy_version,
                                int64_t current_policy_version, const float* log_alpha, float entropy_threshold, float* d_value,
                                float* d_logp, float* d_entropy, float* out20, hipStream_t stream) {
    if (B <= 0 || !values || !logp || !entropy || !old_logp || !adv || !old_values || !returns || !d_value || !d_logp ||
        !d_entropy || !out20)
        return HAB_ERR_ARG;
    ppo_loss_kernel<<<1, 1024, 0, stream>>>(values, logp, entropy, old_logp, adv, old_values, returns, rows, B, clip_param,
                                            value_loss_coef, entropy_coef, use_clipped_value_loss, d_value, d_logp,
                                            d_entropy, out20, is_coeffs, is_stale, policy_version, (long long)current_policy_version,
                                            log_alpha, entropy_threshold);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// Adam step + projection of the adaptive entropy coefficient's log_alpha (one scalar; rl/ppo/ppo.py:112-137 puts it in the same
// optimiser as the policy, :373-375 projects it into [log alpha_min, log alpha_max] after every step).  torch.optim.Adam arithmetic.
__global__ void lagrange_adam_kernel(float* __restrict__ log_alpha, float* __restrict__ m, float* __restrict__ v,
                                     const float* __restrict__ grad, float gscale, float lr, float b1, float b2, float eps, int step,
                                     float lo, float hi, float* __restrict__ alpha_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float g = grad[0] * gscale;
    const float mm = m[0] + (g - m[0]) * (1.0f - b1);
    const float vv = v[0] * b2 + (1.0f - b2) * g * g;
    m[0] = mm; v[0] = vv;
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
    const float p = log_alpha[0] - (lr / bc1) * (mm / denom);
    const float pc = fminf(fmaxf(p, lo), hi);
    log_alpha[0] = pc;
    if (alpha_out) alpha_out[0] = expf(pc);  // learner metric `entropy_coef`, read after the step (ppo.py:276-279)
}
extern "C" int hab_lagrange_adam_step(float* log_alpha, float* exp_avg, float* exp_avg_sq, const float* grad, float grad_scale, float lr,
                                      float beta1, float beta2, float eps, int step, float log_alpha_min, float log_alpha_max,
                                      float* alpha_out, hipStream_t stream) {
    if (!log_alpha || !exp_avg || !exp_avg_sq || !grad || step <= 0) return HAB_ERR_ARG;
    lagrange_adam_kernel<<<1, 64, 0, stream>>>(log_alpha, exp_avg, exp_avg_sq, grad, grad_scale, lr, beta1, beta2, eps, step, log_alpha_min,
                                               log_alpha_max, alpha_out);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// VER: returns over a LINEAR buffer of steps grouped into sequences (episode fragments), rl/ver/ver_rollout_storage.py:430-568.
// One lane per sequence walks its steps backwards (sequence i, step s lives at buffer index select_inds[step_offset[s] + i]).
// The reference runs this recursion in numpy float64 on float32 inputs and rounds the result to float32 on assignment: same
// here.  gae restarts from 0 and bootstraps from 0 at every sequence end, except that the LAST step of the last sequence of an
// environment is the bootstrap step itself: its return is NaN (marks "not a training step") and its value seeds last_value.
// A step keeps its previous return when it is stale and that return is finite.
// ------------------------------------------------------------------------------------------
__global__ void ver_returns_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                   const uint8_t* __restrict__ is_stale, float* __restrict__ returns,
                                   const int* __restrict__ select_inds, const int* __restrict__ step_offsets,
                                   const int* __restrict__ seq_len, const uint8_t* __restrict__ last_seq_mask, int F, double gamma,
                                   double tau_gamma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const int len = seq_len[i];
    const bool last_for_env = last_seq_mask[i] != 0;
    double gae = 0.0, last_value = 0.0;
    for (int s = len - 1; s >= 0; --s) {
        const int b = select_inds[step_offsets[s] + i];
        const double v = (double)values[b];
        const double q_est = (double)rewards[b] + gamma * last_value;
        const double delta = q_est - v;
        gae = delta + tau_gamma * gae;
        const bool boot = last_for_env && (s == len - 1);
        if (boot) gae = 0.0;
        const float old = returns[b];
        if (!is_stale[b] || !isfinite(old)) returns[b] = (float)(gae + v);
        if (boot) returns[b] = __builtin_nanf("");
        last_value = v;
    }
}

extern "C" int hab_ver_compute_returns(const float* rewards, const float* value_preds, const uint8_t* is_stale, float* returns,
                                       const int32_t* select_inds, const int32_t* step_offsets, const int32_t* sequence_lengths,
                                       const uint8_t* last_sequence_in_batch_mask, int F, double gamma, double tau,
                                       hipStream_t stream) {
    if (!rewards || !value_preds || !is_stale || !returns || !select_inds || !step_offsets || !sequence_lengths ||
        !last_sequence_in_batch_mask || F <= 0)
        return HAB_ERR_ARG;
    ver_returns_kernel<<<cdiv(F, 64), 64, 0, stream>>>(rewards, value_preds, is_stale, returns, select_inds, step_offsets, sequence_lengths,
                                                       last_sequence_in_batch_mask, F, gamma, tau * gamma);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// VER importance-sampling coefficients (ver_rollout_storage.py:399-428): steps per environment are counted over the whole buffer,
// is_coeffs[b] = (num_steps + 1) / count[environment_ids[b]]  (fp32 division, as torch does).  counts: num_envs ints of scratch.
__global__ void ver_count_kernel(const int64_t* __restrict__ env_ids, int n, int num_envs, int* __restrict__ counts) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) { const int64_t e = env_ids[b]; if (e >= 0 && e < num_envs) atomicAdd(counts + e, 1); }
}
__global__ void ver_is_coeffs_kernel(const int64_t* __restrict__ env_ids, int n, int num_envs, const int* __restrict__ counts,
                                     float per_env, float* __restrict__ is_coeffs) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) { const int64_t e = env_ids[b]; is_coeffs[b] = (e >= 0 && e < num_envs) ? per_env / (float)counts[e] : 1.0f; }
}
extern "C" int hab_ver_is_coeffs(const int64_t* environment_ids, int n, int num_envs, int num_steps, int32_t* counts_scratch,
                                 float* is_coeffs, hipStream_t stream) {
    if (!environment_ids || !counts_scratch || !is_coeffs || n <= 0 || num_envs <= 0) return HAB_ERR_ARG;
    hipError_t e = hipMemsetAsync(counts_scratch, 0, sizeof(int32_t) * num_envs, stream);
    if (e != hipSuccess) return (int)e;
    ver_count_kernel<<<cdiv(n, 256), 256, 0, stream>>>(environment_ids, n, num_envs, counts_scratch);
    HAB_LAUNCH_CHECK();
    ver_is_coeffs_kernel<<<cdiv(n, 256), 256, 0, stream>>>(environment_ids, n, num_envs, counts_scratch, (float)(num_steps + 1), is_coeffs);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// Global-norm clip + Adam on the flat parameter arena (K16): 2 launches.
//  1) sumsq partials (grid-stride, float4) -> partial[blocks] doubles
//  2) every block re-reduces the (<=1024) partials, derives clip coefficient, applies Adam.
// The update mirrors torch.optim.Adam(foreach): lerp, mul/addcmul, sqrt/div/add eps, addcdiv.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, size_t n, float gscale,
                                                    double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0.0;
    const size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = g4[i];
        v.x *= gscale; v.y *= gscale; v.z *= gscale; v.w *= gscale;
        s += (double)(v.x * v.x) + (double)(v.y * v.y) + (double)(v.z * v.z) + (double)(v.w * v.w);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = g[i] * gscale;
        s += (double)(v * v);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, size_t n,
                                                        const double* __restrict__ partial, int npartial, float gscale,
                                                        float max_norm, float w1, float beta2, float w2, float step_size,
                                                        float bc2_sqrt, float eps, float* __restrict__ norm_out) {
    __shared__ double red[4];
    __shared__ float coef_s;
    double s = 0.0;
    for (int i = threadIdx.x; i < npartial; i += 256) s += partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        float c = (max_norm > 0.0f) ? max_norm / (norm + 1e-6f) : 1.0f;
        coef_s = fminf(c, 1.0f) * gscale;
        if (blockIdx.x == 0 && norm_out) norm_out[0] = norm;
    }
    __syncthreads();
    const float coef = coef_s;
    const size_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= coef;
        mm = mm + w1 * (gg - mm);
        vv = vv * beta2 + w2 * (gg * gg);
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = pp - step_size * (mm / denom);
    };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        upd(p[i], g[i], m[i], v[i]);
}

extern "C" int hab_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                                  double* scratch_partials, int scratch_len, float grad_scale, float max_grad_norm,
                                  float lr, float beta1, float beta2, float eps, int step, float* grad_norm_out,
                                  hipStream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n == 0 || !scratch_partials || scratch_len < 1 || step < 1)
        return HAB_ERR_ARG;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return HAB_ERR_ARG;
    int blocks = (int)fmin((double)scratch_len, (double)cdivl((long long)(n >> 2) + 1, 256));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    sumsq_kernel<<<blocks, 256, 0, stream>>>(grads, n, grad_scale, scratch_partials);
    HAB_LAUNCH_CHECK();
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    int ablocks = (int)cdivl((long long)(n >> 2) + 1, 256);
    if (ablocks > 2048) ablocks = 2048;
    clip_adam_kernel<<<ablocks, 256, 0, stream>>>(params, grads, exp_avg, exp_avg_sq, n, scratch_partials, blocks,
                                                  grad_scale, max_grad_norm, (float)(1.0 - (double)beta1), beta2,
                                                  (float)(1.0 - (double)beta2), step_size, bc2_sqrt, eps, grad_norm_out);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// Action sampling (K12): torch.multinomial(probs, 1, True) on CPU draws q ~ Exp(1) per
// (row, class) and returns argmax(probs / q).  The noise is drawn on the host from the CPU
// generator (same stream the reference consumes); the division/argmax run here.  First maximal
// index wins (torch.argmax semantics).  deterministic=1 -> mode (argmax of probs).
// ------------------------------------------------------------------------------------------
__global__ void sample_kernel(const float* __restrict__ probs, const float* __restrict__ noise, int64_t* __restrict__ actions,
                              int n, int A, int deterministic) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = -INFINITY;
    int arg = 0;
    for (int k = 0; k < A; ++k) {
        const float p = probs[(size_t)i * A + k];
        const float s = deterministic ? p : __fdiv_rn(p, noise[(size_t)i * A + k]);
        if (s > best) { best = s; arg = k; }
    }
    actions[i] = arg;
}

extern "C" int hab_sample_actions(const float* probs, const float* exp_noise, int64_t* actions, int n, int A,
                                  int deterministic, hipStream_t stream) {
    if (n <= 0 || A <= 0 || !probs || !actions || (!deterministic && !exp_noise)) return HAB_ERR_ARG;
    sample_kernel<<<cdiv(n, 64), 64, 0, stream>>>(probs, exp_noise, actions, n, A, deterministic);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}
