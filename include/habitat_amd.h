/*
 * habitat_amd.h -- C-ABI of libhabitat_amd.so, the MI355X (gfx950) kernel library behind the
 * habitat_baselines PPO / DD-PPO training path.
 *
 * The reference (facebookresearch/habitat-lab) has NO native boundary on this path: every op is a
 * PyTorch call reached from Python.  The drop-in boundary is therefore the
 * Python plugin registry; THIS header is the boundary one level below it that a maintainer binds
 * with ctypes (INTEGRATION.md shows the stubs).  Each entry point cites the reference op chain
 * (file:line under habitat-baselines/habitat_baselines/) that it replaces.
 *
 * Conventions: plain device pointers + sizes, `hipStream_t` last, returns 0 on success, a negative
 * HAB_ERR_* for bad arguments, or a positive hipError_t.  Nothing here owns memory except the
 * `hab_policy` engine handle, whose arenas are caller-provided.  All tensors fp32 unless noted;
 * masks are 1 byte per element (torch.bool), 1 = "not done".  Functions are asynchronous on
 * `stream`; none of them synchronises the device.
 */
#ifndef HABITAT_AMD_H
#define HABITAT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define HAB_ABI_VERSION 1
int hab_abi_version(void);
/* Human-readable name of the last HAB_ERR_* / hipError_t code. */
const char* hab_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Synthetic PointNav environment source (replaces VectorEnv.wait_step_at + batch_obs for the
 * synthetic-observation benchmark: habitat/core/vector_env.py:402-410, utils/common.py:191-330).
 * Writes the observations of N envs straight into rollout-storage rows.  Counter-based integer
 * hash, bit-identical to oracle/synth.py.  advance=0: (re)emit obs for the current env clock
 * (reset); advance=1: env_t += 1 then emit obs, reward, not_done.  rgb/depth may be NULL.
 * ------------------------------------------------------------------------------------------- */
int hab_synth_step(uint8_t* rgb /*N,H,W,3*/, float* depth /*N,H,W,1*/, float* goal /*N,2*/, float* reward /*N*/,
                   uint8_t* not_done /*N*/, int64_t* env_t /*N*/, int64_t* since_reset /*N*/, uint32_t seed,
                   uint32_t env_offset, int N, int H, int W, int advance, hipStream_t stream);
/* ObjectNav sensor set for the CURRENT env clock (call after hab_synth_step): semantic int32 (N,H,W,1) in [0,40),
 * objectgoal int64 (N,1) in [0,21) (fixed per env), compass f32 (N,1), gps f32 (N,2).  Any pointer may be NULL.
 * Bit-identical to oracle/synth.py. */
int hab_synth_objectnav_sensors(int32_t* semantic, int64_t* objectgoal, float* compass, float* gps, const int64_t* env_t,
                                uint32_t seed, uint32_t env_offset, int N, int H, int W, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Observation transformers, fused: ResizeShortestEdge followed by CenterCropper
 * (habitat_baselines/common/obs_transformers.py:70-231; utils/common.py:481-557: F.interpolate(mode="area") -- "nearest" for
 * the semantic sensor -- on the float NCHW view, cast back to the sensor dtype, then a center slice).  src (N,H,W,C) NHWC of
 * `dtype`; the image is virtually resized to (resized_h, resized_w) and the window [crop_y0, crop_y0+out_h) x
 * [crop_x0, crop_x0+out_w) of it is written to dst (N,out_h,out_w,C).  Resize only: crop = the whole resized image; crop only:
 * resized extent = source extent.  Bit-identical to the ATen CPU kernels the reference calls.  C <= 4.
 * ------------------------------------------------------------------------------------------- */
#define HAB_DTYPE_U8 0
#define HAB_DTYPE_F32 1
#define HAB_DTYPE_I32 2
#define HAB_RESIZE_AREA 0
#define HAB_RESIZE_NEAREST 1
int hab_obs_resize_crop(const void* src, void* dst, int dtype, int N, int H, int W, int C, int resized_h, int resized_w,
                        int crop_y0, int crop_x0, int out_h, int out_w, int mode, hipStream_t stream);

/* Per-step episode bookkeeping of the rollout loop, fused (rl/ppo/ppo_trainer.py:417-446: current_episode_reward += rewards;
 * running_episode_stats["reward"] += current_episode_reward.where(done, 0); ["count"] += done;
 * current_episode_reward.masked_fill_(done, 0)) plus RolloutStorage.insert's prev_actions[t+1] = actions[t]
 * (common/rollout_storage.py:124-130; actions / prev_actions_next may both be NULL).  All arrays have N rows. */
int hab_rollout_step_stats(const float* rewards, const uint8_t* not_done, float* current_episode_reward, float* stat_reward,
                           float* stat_count, const int64_t* actions /*N,action_dim*/, int64_t* prev_actions_next, int N,
                           int action_dim, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * RolloutStorage.compute_returns (common/rollout_storage.py:174-205).  Buffers are (T+1, N).
 * Writes value_preds[T] = next_value and returns[0..T-1] (use_gae) / returns[0..T].
 * variant HAB_GAE_EXACT: lane-per-env sequential recurrence in the reference's operation order,
 * bitwise equal to the PyTorch-CPU loop.  HAB_GAE_SCAN: wavefront-per-env affine suffix scan
 * with shuffles (fp32 round-off level agreement).
 * ------------------------------------------------------------------------------------------- */
#define HAB_GAE_EXACT 0
#define HAB_GAE_SCAN 1
int hab_compute_returns(const float* rewards, float* value_preds, const uint8_t* masks, float* returns,
                        const float* next_value /*N*/, int T, int N, float gamma, float tau, int use_gae,
                        int variant, hipStream_t stream);

/* PPO.get_advantages (rl/ppo/ppo.py:139-153) and distributed_var_mean (rl/ddppo/algo/ddppo.py:59-84).
 * count = (T+1)*N.  stats = {mean, var, finite_count}. */
#define HAB_ADV_RAW 0             /* adv = returns - value_preds */
#define HAB_ADV_LOCAL_NORMALIZE 1 /* + (adv-mean)*rsqrt(var+1e-5), unbiased var (torch.var_mean) */
#define HAB_ADV_STATS_MEAN 2      /* adv + local mean -> stats_out[0] (caller all-reduces) */
#define HAB_ADV_EXT_NORMALIZE 3   /* normalise with ext_stats = {global mean, global var} */
#define HAB_ADV_STATS_VAR 4       /* biased 2nd moment about ext_stats[0] -> stats_out[1] */
int hab_advantages(const float* returns, const float* value_preds, float* adv, int count, int mode,
                   const float* ext_stats, float* stats_out, hipStream_t stream);

/* PPO._update_from_batch loss terms + their gradient (rl/ppo/ppo.py:195-250, metrics :260-275).
 * values/logp/entropy: dense [B] outputs of evaluate_actions.  old_logp/adv/old_values/returns are
 * (T+1,N) storage buffers gathered through rows[f] (NULL = dense).  out12 = {value_loss,
 * action_loss, dist_entropy, total, value_pred min/mean/max, prob_ratio min/mean/max,
 * fraction_clipped, B}. */
int hab_ppo_loss(const float* values, const float* logp, const float* entropy, const float* old_logp,
                 const float* adv, const float* old_values, const float* returns, const int* rows, int B,
                 float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                 float* d_value, float* d_logp, float* d_entropy, float* out12, hipStream_t stream);

/* The same with VER's importance weights and learner statistics (rl/ppo/ppo.py:226-231,262-263,285-299): every per-frame loss term
 * is multiplied by min(is_coeffs, 1) before the batch mean.  is_coeffs / is_stale / policy_version are storage buffers gathered
 * through rows (each nullable).  out20 = out12 + {[12] reserved (grad norm), ver_is_coeffs min/mean/max, fraction_stale,
 * policy_version_difference min/mean/max}.
 * log_alpha (nullable, device scalar): adaptive entropy penalty (LagrangeInequalityCoefficient, utils/common.py:749-806; ppo.py:85-96,
 * 236-239): the entropy coefficient is exp(*log_alpha) instead of entropy_coef, the total loss carries alpha * (threshold - [ent]) -
 * [alpha] * ent, and out[20] = d loss / d log_alpha, out[21] = alpha (out must then hold 24 floats). */
int hab_ppo_loss_ver(const float* values, const float* logp, const float* entropy, const float* old_logp, const float* adv,
                     const float* old_values, const float* returns, const int* rows, int B, float clip_param, float value_loss_coef,
                     float entropy_coef, int use_clipped_value_loss, const float* is_coeffs, const uint8_t* is_stale,
                     const int64_t* policy_version, int64_t current_policy_version, const float* log_alpha, float entropy_threshold,
                     float* d_value, float* d_logp, float* d_entropy, float* out20, hipStream_t stream);
/* Adam step (torch.optim.Adam arithmetic, gradient first multiplied by grad_scale) + projection into [log_alpha_min, log_alpha_max] of
 * the adaptive entropy coefficient (rl/ppo/ppo.py:112-137,373-375). */
int hab_lagrange_adam_step(float* log_alpha, float* exp_avg, float* exp_avg_sq, const float* grad, float grad_scale, float lr, float beta1,
                           float beta2, float eps, int step, float log_alpha_min, float log_alpha_max,
                           float* alpha_out /* nullable: exp(log_alpha) after the step */, hipStream_t stream);

/* VERRolloutStorage.compute_returns (rl/ver/ver_rollout_storage.py:430-568) on the linear step buffer: GAE per sequence (episode
 * fragment) in float64 like the numpy loop, result rounded to float32; the last step of an environment's last sequence is the
 * bootstrap step (return = NaN); stale steps keep a finite previous return.  select_inds / step_offsets ([max_len+1] prefix sums
 * of num_seqs_at_step) / sequence_lengths / last_sequence_in_batch_mask: device copies of the buffer's pack info (F sequences). */
int hab_ver_compute_returns(const float* rewards, const float* value_preds, const uint8_t* is_stale, float* returns,
                            const int32_t* select_inds, const int32_t* step_offsets, const int32_t* sequence_lengths,
                            const uint8_t* last_sequence_in_batch_mask, int F, double gamma, double tau, hipStream_t stream);
/* VERRolloutStorage.after_rollout's importance coefficients (:399-428): is_coeffs[b] = (num_steps + 1) / #steps of
 * environment_ids[b] in the buffer.  counts_scratch: num_envs int32. */
int hab_ver_is_coeffs(const int64_t* environment_ids, int n, int num_envs, int num_steps, int32_t* counts_scratch, float* is_coeffs,
                      hipStream_t stream);

/* nn.utils.clip_grad_norm_ + optim.Adam(foreach).step (rl/ppo/ppo.py:347-371,112-137,257) over the
 * flat parameter arena.  grads are first multiplied by grad_scale (1/world_size after a sum
 * all-reduce).  scratch_partials: >= 1024 doubles.  step counts from 1.  All pointers 16-B aligned. */
int hab_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                       double* scratch_partials, int scratch_len, float grad_scale, float max_grad_norm, float lr,
                       float beta1, float beta2, float eps, int step, float* grad_norm_out, hipStream_t stream);

/* CustomFixedCategorical.sample (utils/common.py:64-68) = torch.multinomial(probs,1,True):
 * argmax_k probs[i,k] / exp_noise[i,k] with exp_noise ~ Exp(1) drawn by the caller from the CPU
 * generator; deterministic=1 gives .mode(). */
int hab_sample_actions(const float* probs, const float* exp_noise, int64_t* actions, int n, int A, int deterministic,
                       hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Layer-level contractions (all on the fp32 MFMA igemm kernel).  Activations NHWC fp32 with
 * channels % 4 == 0; conv weights are consumed from packed copies made by hab_repack_conv_weight;
 * weight gradients are written in the reference OIHW / [out][in] layouts.  ws = split-K / reduction
 * scratch (may be NULL -> no split-K).
 *   conv fwd/dgrad/wgrad  <- nn.Conv2d fwd + autograd (rl/models/simple_cnn.py:68-93,
 *                            rl/ddppo/policy/resnet.py:19-34,207-219)
 *   obs conv              <- the permute/.float()/255/cat ingest fused into conv1 (simple_cnn.py:139-156)
 *   linear fwd/dgrad/wgrad<- nn.Linear (simple_cnn.py:92, policy.py:416-424, rnn input projections)
 *
 * Matrix path.  The contractions are fp32 in, fp32 out, fp32-equivalent arithmetic on either of two instruction paths:
 *   v_mfma_f32_32x32x2_f32 (igemm.h), or -- hab_set_matrix_path bits, env HAB_BF3 -- the exact three-term bf16 split of both
 *   operands with the six partial products >= 2^-16 on v_mfma_f32_32x32x16_bf16 (igemm_bf3.h; dropped terms <= 2^-24 relative):
 *     bit 0  r-contiguous x r-contiguous problems (conv fwd / dgrad incl. the merged stride-2 dgrad, Linear fwd)
 *     bit 1  observation-ingest convolution, forward and weight gradient (uint8 rgb is exact in ONE bf16 plane; obs_conv_bf3.h,
 *            obs_wgrad_bf3.h)
 *     bit 2  problems with an i/j-contiguous operand (weight gradients, Linear dgrad): register-transposed staging
 *     bit 3  prefer bit 2's kernel over the fp32 patch / LDS-DMA weight-gradient kernels
 *     bit 4  stride-1 3x3 convolutions with N = 32 (fwd + dgrad): input patch resident in LDS (conv_patch_bf3.h)
 *     bit 5  producer / consumer waves where they measured faster (igemm_bf3_ws.h: 128 x 128 forward-form tiles with K >= 2048;
 *            obs_conv_bf3_ws.h: the observation-ingest convolution); bit-identical results
 *     bit 6  the observation-ingest convolution (8x8 / 4, RGB-D -> 32) with the input patch resident in LDS (obs_conv_patch.h)
 *     bit 7  (with bit 3) weight gradient of the small-channel convolutions -- SimpleCNN conv2 (4x4 / 2) and conv3, ResNet layer1 and
 *            layer2 at 128^2 -- with the strip resident in LDS in pixel-major layout, fragments by LDS transpose reads
 *            (wgrad3x3_bf3.h)
 *     bit 8  (with bit 0) SimpleCNN conv2's forward (4x4 / 2, 32 -> 64 at 63 x 63, >= 16 frames) with the input strip resident in LDS and
 *            the filter slices resident in the waves' registers (conv2_fwd_strip.h)
 *     bit 9  (with bit 0) SimpleCNN conv2's data gradient on the same scheme: dY strip in LDS, filter slices in registers, the four taps
 *            of a row class folded in LDS, ReLU mask fused (conv2_dgrad_strip.h)
 *     bit 10 (with bit 0) large Linear layers (>= 4 GFLOP: SimpleCNN's visual fc forward / data gradient / weight gradient, the ResNet
 *            policies' recurrent input projection) on the plain dense GEMM kernel: 256 x 128 tiles, double-buffered LDS, transpose-read
 *            fragments for k-strided operands, LDS-staged epilogue incl. the flatten permutation (dense_bf3.h)
 *     bit 11 (tests) bit 10's kernel for every shape it applies to
 *     bit 12 (tests) the time-major recurrence as one launch per step instead of the persistent kernels (rnn_persist.h; bit-identical)
 *   Default 2047 (bits 0-10), env HAB_BF3 overrides.  hab_set_matrix_path(mode >= 0) sets the mask and returns the previous one;
 *   mode < 0 only queries.  Results are fp32-equivalent on either path (tests/test_gpu_bf3.py: error vs float64 of both).
 * ------------------------------------------------------------------------------------------- */
int hab_set_matrix_path(int mode);

int hab_conv2d_fwd(const float* x, const float* w_fwd, const float* bias, float* y, int B, int H, int W, int C, int Cout,
                   int KH, int KW, int stride, int pad, int relu, float* ws, size_t ws_floats, hipStream_t stream);
int hab_obs_conv2d_fwd(const uint8_t* rgb, const float* depth, const int* rows, const float* w_fwd, const float* bias,
                       float* y, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu, float* ws,
                       size_t ws_floats, hipStream_t stream);
int hab_conv2d_dgrad(const float* dy, const float* w_dgrad, const float* relu_mask, const float* add, float* dx, int B,
                     int H, int W, int C, int Cout, int KH, int KW, int stride, int pad, float* ws, size_t ws_floats,
                     hipStream_t stream);
/* dbias (nullable): bias gradient [Cout] = column sums of dy, produced by the same launch. */
int hab_conv2d_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, int B, int H, int W, int C, int Cout,
                     int KH, int KW, int stride, int pad, float* ws, size_t ws_floats, hipStream_t stream);
int hab_obs_conv2d_wgrad(const uint8_t* rgb, const float* depth, const int* rows, const float* dy, float* dw_oihw,
                         float* dbias, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, float* ws,
                         size_t ws_floats, hipStream_t stream);
int hab_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* y, int ldy, int M, int N,
                   int K, int relu, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
int hab_linear_dgrad(const float* dy, int lddy, const float* w, int ldw, const float* relu_mask, int ldmask, float* dx,
                     int lddx, int M, int n_in, int n_out, int accumulate, float* ws, size_t ws_floats,
                     hipStream_t stream);
int hab_linear_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int lddw, int M, int n_out, int n_in,
                     int perm_c, int perm_hw, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
int hab_colsum(const float* a, int lda, int M, int N, float* out, int accumulate, float* ws, size_t ws_floats,
               hipStream_t stream);
/* OIHW -> w_fwd[co][(kh,kw,ci)] (ci padded to cin_padded) and w_dgrad[ci][(kh,kw,co)] (either may be NULL). */
int hab_repack_conv_weight(const float* w_oihw, float* w_fwd, float* w_dgrad, int Cout, int Cin, int KH, int KW,
                           int cin_padded, hipStream_t stream);
/* Linear weight behind nn.Flatten of an NCHW map: w_packed[n][hw*C + c] = w[n][c*HW + hw]. */
int hab_repack_flatten_weight(const float* w, float* w_packed, int N, int C, int HW, hipStream_t stream);
int hab_transpose2d(const float* w, float* wt, int R, int C, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * HBM-bound kernels of the GroupNorm-ResNet encoder (rl/ddppo/policy/resnet_policy.py:255-276, resnet.py:196-281,
 * running_mean_and_var.py:24-78).  All NHWC fp32, channel counts multiples of 4.
 * ------------------------------------------------------------------------------------------- */
/* Ingest: per visual key permute -> uint8 * fp32(1/255) -> cat (rgb,depth or depth,rgb) -> avg_pool2d(2); channels
 * zero-padded to cpad (4 or 8); observations read in place through rows[] (resnet_policy.py:259-271). */
int hab_obs_ingest_pool(const uint8_t* rgb, const float* depth, const int32_t* semantic, const int* rows, float* y, int B, int H, int W,
                        int cpad, int c_rgb, int c_depth, int c_sem /* first output channel of each key, -1 = absent */,
                        hipStream_t stream);
/* Channel moments over all pixels: mode 0 -> out[c] = mean, mode 1 -> out[c] = mean((x - mean[c])^2)
 * (running_mean_and_var.py:33-45).  scratch: >= 1024*cpad doubles. */
int hab_channel_moments(const float* x, int64_t npix, int cpad, int mode, const float* mean, float* out, double* scratch,
                        int scratch_len, hipStream_t stream);
/* Chan merge of the running statistics with a batch of n frames (running_mean_and_var.py:54-71). */
int hab_running_mean_var_update(float* r_mean, float* r_var, float* r_count, const float* b_mean, const float* b_var, float n, int C,
                                hipStream_t stream);
/* x = addcmul(-mean*inv_std, x, inv_std), inv_std = rsqrt(max(var, 1e-2)), in place (running_mean_and_var.py:73-78). */
int hab_running_mean_var_normalize(float* x, int64_t npix, int cpad, int C, const float* mean, const float* var, hipStream_t stream);
/* nn.GroupNorm forward [+ residual] [+ ReLU] (resnet.py:51-57,67-69); mean / rstd ([B][groups]) are kept for backward.
 * ws (nullable): scratch for the chunk-parallel path of frames > 128 KB, B * ceil(HW*C / 16384) * groups * 2 floats
 * (backward: B * ceil(HW*C / 8192) * 2 * C); without it such frames take the slower streaming kernel. */
int hab_groupnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, const float* residual, float* mean,
                      float* rstd, int B, int HW, int C, int groups, int relu, float eps, float* ws, int64_t ws_floats,
                      hipStream_t stream);
/* GroupNorm backward with the ReLU mask of the fused output (relu_out, nullable) applied to dy first; dy_masked (nullable)
 * receives the masked dy (gradient of the residual branch); chan_sums [B][2][C] = per-frame sum dy', sum dy'*xhat
 * (reduce over frames with hab_colsum -> dbeta, dgamma). */
int hab_groupnorm_bwd(const float* x, const float* dy, const float* relu_out, float* dx, float* dy_masked, const float* gamma,
                      const float* mean, const float* rstd, float* chan_sums, int B, int HW, int C, int groups, float* ws,
                      int64_t ws_floats, hipStream_t stream);
/* Convolution (bias-free) + GroupNorm [+ residual] [+ ReLU] of one ResNet layer in ONE launch, for small batches: the rollout's
 * 64-frame `act`, hab_policy_encode, small minibatches (csrc/conv_gn_slab.h).  Replaces nn.Conv2d -> nn.GroupNorm -> (+ identity) ->
 * ReLU of BasicBlock / Bottleneck / downsample / compression (rl/ddppo/policy/resnet.py:19-34,51-69,129-152,207-219;
 * resnet_policy.py:213-234) as PPOTrainer._compute_actions_and_step_envs reaches them once per environment step
 * (rl/ppo/ppo_trainer.py:343-399).
 *   hab_split_weight_planes: forward-packed weight w_fwd [Cout][K = KH*KW*C] (Cout % 32 == 0, K % 16 == 0) -> its exact three-term
 *       bf16 split (x = p0 + p1 + p2) as three planes of Cout*K uint16 in MFMA fragment order:
 *       planes[((p * Cout/32 + co/32) * K/16 + k/16) * 512 + ((k % 16) / 8 * 32 + co % 32) * 8 + k % 8] = plane p of w_fwd[co][k],
 *       so the 64 lanes of a wave read an operand fragment as 1 KB of contiguous memory; done once per optimiser step, shared by
 *       every step of a rollout.
 *   hab_conv_gn_fwd: x NHWC [B][H][W][C] (C % 16 == 0), w_planes from above; y [B][Ho*Wo][Cout]; residual (nullable) same shape as y;
 *       raw / mean / rstd (nullable, mean and rstd together): convolution output before the normalisation and the statistics
 *       [B][groups], kept for hab_groupnorm_bwd.  Covered: Ho*Wo <= 256, Cout % 32 == 0, group size Cout / groups in {4 .. 128}
 *       (> 32 only for Ho*Wo <= 32), the input of one workgroup's frames (256 / (Ho*Wo) frames x H*W*C, 1x1: x Ho*Wo*C) <= ~25 K
 *       elements; anything else returns HAB_ERR_UNSUPPORTED and the caller runs hab_conv2d_fwd + hab_groupnorm_fwd.
 *   fp32 in / out, fp32-equivalent arithmetic (csrc/igemm_bf3.h), exact two-pass statistics; deterministic. */
int hab_split_weight_planes(const float* w_fwd, int Cout, int K, uint16_t* planes, hipStream_t stream);
/* Stem convolution of the GroupNorm-ResNet (7x7 / stride 2 / padding 3, 4 input channels -> 32, bias-free; resnet.py:207-219 `conv1`
 * on the 2x2-averaged observation of resnet_policy.py:259-272) with the input strip resident in LDS (csrc/stem_conv_strip.h).
 *   hab_stem_split_weights: forward-packed filter w_fwd [32][7][7][4] -> 3 x 14 x 512 uint16: its exact three-term bf16 split in MFMA
 *       fragment order (k-step 2 kh + j holds reduction slots 16 j .. 16 j + 15 of filter row kh = (kw, ci) pairs, kw = 7 zero-padded).
 *   hab_stem_conv_fwd: x NHWC [B][H][W][4] -> y [B][Ho][Wo][32], Ho = (H - 1) / 2 + 1; covered: Wo <= 64 (observations up to 256 wide);
 *       wider inputs return HAB_ERR_UNSUPPORTED (the caller runs hab_conv2d_fwd).  norm: NULL, or 16 floats (16-byte aligned) of the
 *       RunningMeanAndVar normalisation (running_mean_and_var.py:72-78) applied to x while it is staged -- x_hat[c] = fma(x[c], norm[c],
 *       norm[8 + c]) with norm[c] = rsqrt(max(var[c], 1e-2)), norm[8 + c] = -mean[c] * norm[c] -- so that the training forward does not
 *       write a normalised copy of the observation tensor (bit-identical to hab_running_mean_var_normalize followed by norm = NULL).
 *       gn_part: NULL, or [B][ceil(Ho / 8)][gn_groups][2] floats: per frame, strip of 8 output rows and GroupNorm group (gn_groups in
 *       {8, 16, 32}) the mean and the centred sum of squares M2 of the strip's outputs (exact local two-pass on the accumulators), which
 *       the GroupNorm that follows (resnet.py:215) merges with Chan's formula instead of reading the output tensor for its statistics. */
int hab_stem_split_weights(const float* w_fwd, uint16_t* planes, hipStream_t stream);
int hab_stem_conv_fwd(const float* x, const uint16_t* w_planes, float* y, int B, int H, int W, const float* norm, float* gn_part, int gn_groups,
                      hipStream_t stream);
/* Its weight gradient (csrc/stem_wgrad_strip.h; the autograd backward of resnet.py:207-219 `conv1` under rl/ppo/ppo.py:253): x NHWC
 * [B][H][W][4] (channels >= creal are padding), dy [B][Ho][Wo][32] -> dw_oihw [32][creal][7][7] (overwritten).  Both operands are
 * resident in LDS, fragments come from transpose reads; covered: Wo a multiple of 16, <= 64; ws >= 256 * 7168 floats of scratch;
 * otherwise HAB_ERR_UNSUPPORTED (the caller runs hab_conv2d_wgrad).  norm: as for hab_stem_conv_fwd. */
int hab_stem_conv_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int creal, float* ws, size_t ws_floats,
                        const float* norm, hipStream_t stream);
int hab_conv_gn_fwd(const float* x, const uint16_t* w_planes, const float* gamma, const float* beta, const float* residual, float* y,
                    float* raw, float* mean, float* rstd, int B, int H, int W, int C, int Cout, int KH, int KW, int stride, int pad,
                    int groups, int relu, float eps, hipStream_t stream);
/* nn.MaxPool2d(3, stride 2, padding 1) (resnet.py:220); idx = window offset of the first maximum (1 byte per output). */
int hab_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, hipStream_t stream);
int hab_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, hipStream_t stream);
/* 1-D sensor embeddings of PointNavResNetNet.forward (resnet_policy.py:662-753), 32 wide each, written side by side to
 * out[:, col0 + 32*slot ...]; saved [B][nslots][4] keeps the features / tokens for the backward pass. */
#define HAB_EMB_POLAR 0   /* pointgoal_with_gps_compass (rho, phi): Linear(3,32)([rho, cos(-phi), sin(-phi)]) */
#define HAB_EMB_TOKEN 1   /* objectgoal id: Embedding(num_tokens, 32) */
#define HAB_EMB_COSSIN 2  /* compass x: Linear(2,32)([cos x, sin x]) */
#define HAB_EMB_LIN2 3    /* gps (x, y): Linear(2,32) */
#define HAB_EMB_PREV 4    /* previous action: Embedding(A+1, 32)(mask ? a+1 : 0) */
#define HAB_EMB_PREVLIN 5 /* continuous previous action, float (rows, A <= 4): Linear(A, 32)(mask * a); num_tokens carries A */
#define HAB_EMB_LINN 6    /* pointgoal (PointGoalSensor) / proximity, float (rows, d <= 4): Linear(d, 32) on the raw values; num_tokens carries d */
typedef struct hab_embed_slot {
    int32_t kind;
    const void* input;     /* sensor values in arena rows: f32[2] / i64[1] / f32[1] / f32[2] / i64[1] per row */
    const float* weight;   /* Linear weight [32][nfeat] or table [num_tokens][32] */
    const float* bias;     /* Linear bias or NULL */
    float* d_weight;       /* backward outputs (NULL in forward) */
    float* d_bias;
    int32_t num_tokens;
} hab_embed_slot;
int hab_nav_embed_fwd(const hab_embed_slot* slots, int nslots, const uint8_t* masks, const int* rows, float* out, int ld, int col0, int B,
                      float* saved, hipStream_t stream);
int hab_nav_embed_bwd(const hab_embed_slot* slots, int nslots, const float* saved, const float* dout, int ld, int col0, int B, float* ws,
                      size_t ws_floats, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * build_pack_info_from_dones (rl/models/rnn_state_encoder.py:35-168), HOST routine (no GPU).
 * dones: (T,N) row-major bytes.  Output arrays sized: select_inds T*N; num_seqs_at_step T; the
 * per-fragment arrays T*N (worst case); the per-env arrays N.  Ties between equal-length fragments
 * are ordered by (episode id, env).
 * ------------------------------------------------------------------------------------------- */
int hab_build_pack_info(const uint8_t* dones, int T, int N, int64_t* select_inds, int64_t* num_seqs_at_step,
                        int64_t* sequence_starts, int64_t* sequence_lengths, int64_t* rnn_state_batch_inds,
                        uint8_t* last_sequence_in_batch_mask, uint8_t* first_sequence_in_batch_mask,
                        int64_t* last_sequence_in_batch_inds, int64_t* first_episode_in_batch_inds,
                        int64_t* first_step_for_env, int32_t* num_fragments, int32_t* max_len);
/* build_pack_info_from_episode_ids (rnn_state_encoder.py:35-150): P frames in any order tagged (episode, environment, step).
 * Outputs as above; environments are renumbered 0..n-1 in increasing id order, first_step_for_env has n entries. */
int hab_build_pack_info_from_ids(const int64_t* episode_ids, const int64_t* environment_ids, const int64_t* step_ids, int P,
                                 int64_t* select_inds, int64_t* num_seqs_at_step, int64_t* sequence_starts, int64_t* sequence_lengths,
                                 int64_t* rnn_state_batch_inds, uint8_t* last_sequence_in_batch_mask,
                                 uint8_t* first_sequence_in_batch_mask, int64_t* first_step_for_env, int32_t* num_fragments,
                                 int32_t* max_len, int32_t* num_envs);

/* ---------------------------------------------------------------------------------------------
 * Policy engine: NetPolicy.act / get_value / evaluate_actions (rl/ppo/policy.py:324-402) and the
 * backward pass that total_loss.backward() runs through them (rl/ppo/ppo.py:253), for
 * PointNavBaselinePolicy (policy.py:427-589) [arch 0] and PointNavResNetPolicy
 * (rl/ddppo/policy/resnet_policy.py:50-767) [arch 1].  The engine defines the flat parameter
 * arena layout (names/shapes = the reference state_dict()); all arenas are caller-allocated.
 * ------------------------------------------------------------------------------------------- */
#define HAB_ARCH_SIMPLE_CNN 0
#define HAB_ARCH_RESNET 1
#define HAB_RNN_GRU 0
#define HAB_RNN_LSTM 1

typedef struct hab_policy_desc {
    int32_t arch;          /* HAB_ARCH_* */
    int32_t backbone;      /* HAB_BACKBONE_* (arch 1) */
    int32_t baseplanes;    /* 32 */
    int32_t normalize_visual_inputs;
    int32_t rnn_type;      /* HAB_RNN_* */
    int32_t rnn_layers;
    int32_t hidden;        /* 512 */
    int32_t num_actions;   /* Discrete(n) */
    int32_t H, W;          /* observation size */
    int32_t has_rgb, has_depth;
    int32_t goal_dim;      /* pointgoal_with_gps_compass dims (2) */
    int32_t max_frames;    /* largest T*n of an evaluate call */
    int32_t max_envs;      /* largest n of an act call */
    /* arch 1 (PointNavResNetPolicy) only: */
    int32_t visual_order;  /* visual keys in observation-space order, 2 bits each from the LSB: 1 rgb, 2 depth, 3 semantic, 0 end */
    int32_t has_semantic;  /* int32 (H,W,1) semantic sensor */
    int32_t num_object_categories; /* > 0: objectgoal sensor (ObjectNav) with that many categories */
    int32_t has_compass, has_gps;  /* compass f32 (1,), gps f32 (2,) */
    /* action distribution (rl/ppo/policy.py:266-286).  HAB_DIST_GAUSSIAN (arch 1): GaussianNet / CustomNormal (utils/common.py:99-175)
     * over a Box action space of num_actions <= 4 dimensions: actions / prev_actions are FLOAT (rows, num_actions) buffers (passed
     * through the int64 pointers of the entry points), the previous-action embedding is Linear(num_actions, 32) on
     * masks * prev_actions (resnet_policy.py:424-428,754-757), exp_noise of hab_policy_act carries N(0,1) draws (rsample). */
    int32_t action_dist;           /* HAB_DIST_* */
    int32_t gauss_flags;           /* HAB_GAUSS_* bits (ActionDistributionConfig, default_structured_configs.py:70-85) */
    float gauss_min_std, gauss_max_std;  /* clamp range of the RAW std output (min_log_std / max_log_std when USE_LOG_STD) */
    /* further 1-D goal sensors (resnet_policy.py:489-494,510-515,694-700): dims of the PointGoalSensor / ProximitySensor vectors, 0 = absent.
     * arch 0 (PointNavBaselinePolicy): a `pointgoal` sensor is passed as `goal` (policy.py:509-514), these stay 0. */
    int32_t pointgoal_dim, proximity_dim;
} hab_policy_desc;
/* rl/ddppo/policy/resnet.py:296-345 */
#define HAB_BACKBONE_RESNET18 18
#define HAB_BACKBONE_RESNET50 50
#define HAB_BACKBONE_RESNEXT50 51      /* resneXt50: expansion 2, base width x2, grouped 3x3 (cardinality base_planes / 2) */
#define HAB_BACKBONE_SE_RESNET50 52    /* se_resnet50: Bottleneck + squeeze-and-excitation gate */
#define HAB_BACKBONE_SE_RESNEXT50 53   /* se_resneXt50 */
#define HAB_BACKBONE_SE_RESNEXT101 101 /* se_resneXt101: stages 3, 4, 23, 3 */
#define HAB_DIST_CATEGORICAL 0
#define HAB_DIST_GAUSSIAN 1
#define HAB_GAUSS_TANH_MU 1        /* action_activation == "tanh" */
#define HAB_GAUSS_USE_LOG_STD 2    /* std = exp(raw) */
#define HAB_GAUSS_USE_SOFTPLUS 4   /* std = softplus(std) */
#define HAB_GAUSS_USE_STD_PARAM 8  /* state-independent std parameter `action_distribution.std` instead of 2*A linear outputs */
#define HAB_GAUSS_CLAMP_STD 16

typedef struct hab_obs {   /* arena base pointers; frame f lives at row rows[f] (or f) */
    const uint8_t* rgb;    /* (rows, H, W, 3) */
    const float* depth;    /* (rows, H, W, 1) */
    const float* goal;     /* (rows, goal_dim) pointgoal_with_gps_compass */
    const int64_t* prev_actions; /* (rows, 1) */
    const int32_t* semantic;     /* (rows, H, W, 1) */
    const int64_t* objectgoal;   /* (rows, 1) */
    const float* compass;        /* (rows, 1) */
    const float* gps;            /* (rows, 2) */
    const float* visual_features; /* arch 1, optional: (rows, C, Hf, Wf) output of the visual encoder stored by the rollout
                                    (PointNavResNetNet.PRETRAINED_VISUAL_FEATURES_KEY, resnet_policy.py:399,636-646: frozen-encoder
                                    training, rl.ddppo.train_encoder=False).  When non-NULL, act / evaluate use it instead of running
                                    the encoder and backward stops at visual_fc (no encoder gradients are written). */
    const float* pointgoal;      /* (rows, pointgoal_dim)  PointGoalSensor, arch 1 */
    const float* proximity;      /* (rows, proximity_dim)  ProximitySensor, arch 1 */
} hab_obs;

typedef struct hab_pack_info { /* int32 copies of hab_build_pack_info's arrays */
    const int32_t* select_inds;           /* device [P] */
    const int32_t* frag_env;              /* device [F]  rnn_state_batch_inds */
    const int32_t* frag_start;            /* device [F]  sequence_starts */
    const int32_t* step_offsets_host;     /* host  [max_len+1] */
    const int32_t* num_seqs_at_step_host; /* host  [max_len] */
    int32_t P, F, max_len;
    /* VER minibatches (frames in any order, hab_build_pack_info_from_ids): device [n] first_step_for_env -- the frame whose arena
     * row holds the hidden state of batch environment e.  NULL: frame e (the t = 0 row of a time-major T x n minibatch). */
    const int32_t* env_first_frame;
} hab_pack_info;

typedef struct hab_policy hab_policy;

int hab_policy_create(const hab_policy_desc* desc, hab_policy** out);
void hab_policy_destroy(hab_policy* p);
int hab_policy_num_params(const hab_policy* p);
int hab_policy_param_info(const hab_policy* p, int i, char* name, int name_cap, int64_t* shape4, int* ndim,
                          int64_t* offset_floats);
/* 1 if entry i is a registered buffer of the reference module (RunningMeanAndVar statistics), 0 for a parameter. */
int hab_policy_param_is_buffer(const hab_policy* p, int i);
int64_t hab_policy_param_floats(const hab_policy* p);
int64_t hab_policy_packed_floats(const hab_policy* p);
int64_t hab_policy_work_floats(const hab_policy* p);
/* All arenas 256-byte aligned device memory; grads may be NULL for inference-only use. */
int hab_policy_bind(hab_policy* p, float* params, float* grads, float* packed, float* work, int64_t work_floats);
int hab_policy_repack(hab_policy* p, hipStream_t stream);
/* nn.Module.train()/eval(): RunningMeanAndVar updates its statistics only in training mode
 * (rl/ddppo/policy/running_mean_and_var.py:24-25). */
int hab_policy_set_training(hab_policy* p, int training);
/* DD-PPO: in-place all-reduce (sum, then * scale) of n floats at device pointer buf, ordered on the engine's stream
 * (running_mean_and_var.py:38-41,47-49).  The engine calls it during evaluate in training mode when world_size > 1. */
typedef void (*hab_allreduce_fn)(float* buf, int n, float scale, void* ctx);
int hab_policy_set_allreduce(hab_policy* p, hab_allreduce_fn fn, void* ctx, int world_size);
/* DD-PPO: early gradient exchange.  During hab_policy_backward the engine calls fn(first, count, ctx) once, right after the last
 * kernel that writes grads[first .. first + count) has been enqueued: the contiguous TAIL of the gradient arena from the visual
 * fc weight to the end (fc / visual_fc, recurrent encoder, heads: 99.5 % of the bytes of the SimpleCNN policy, 63 % of ResNet18's),
 * whose producers run BEFORE the convolution stack's backward.  The caller starts an asynchronous all-reduce of that range there
 * (DistributedDataParallel's bucket overlap, ddppo.py:128-140, with one bucket) and reduces [0, first) after backward returns. */
typedef void (*hab_grad_ready_fn)(int64_t first, int64_t count, void* ctx);
int hab_policy_set_grad_ready(hab_policy* p, hab_grad_ready_fn fn, void* ctx);
/* DD-PPO, device-side form of the two hooks above (csrc/comm.hip): an RCCL communicator owned by the library.  Replaces what
 * DistributedDataParallel's reducer does for the reference (rl/ddppo/algo/ddppo.py:128-140: buckets all-reduced from autograd hooks
 * while backward runs) and the all_reduce calls of rl/ddppo/policy/running_mean_and_var.py:38-49, with no interpreter in between.
 *   hab_comm_available    1 when librccl's symbols were found (dlopen; the copy PyTorch loaded when there is one)
 *   hab_comm_unique_id    rank 0: 128 bytes to hand to every rank (e.g. through torch.distributed's store)
 *   hab_comm_create       ncclCommInitRank on the CURRENT device + a private high-priority stream for the exchange
 *   hab_comm_allreduce_sum  in-place fp32 sum over the ranks, in order on `stream` (self-test, statistics)
 *   hab_policy_set_comm   from now on hab_policy_backward enqueues the all-reduce of every finished tail of the gradient arena on
 *                         the communicator's stream the moment it is final, and the training forward sums the RunningMeanAndVar
 *                         moments on the compute stream; the hab_policy_set_allreduce / _set_grad_ready callbacks are not called
 *   hab_policy_grad_sync  after hab_policy_backward: exchanges the rest of the arena (its head) and makes `stream` wait for all of
 *                         it; the arena then holds the SUMS over ranks (1 / world_size goes into hab_clip_adam_step's grad_scale) */
typedef struct hab_comm hab_comm;
int hab_comm_available(void);
int hab_comm_unique_id(uint8_t* out128);
int hab_comm_create(const uint8_t* id128, int world, int rank, hab_comm** out);
void hab_comm_destroy(hab_comm* c);
int hab_comm_world_size(const hab_comm* c);
int hab_comm_allreduce_sum(hab_comm* c, float* buf, int64_t count, hipStream_t stream);
int hab_policy_set_comm(hab_policy* p, hab_comm* c);
int hab_policy_grad_sync(hab_policy* p, hipStream_t stream);
/* The visual encoder alone (ResNetEncoder.forward, resnet_policy.py:255-276) on n frames: out (n, C, Hf, Wf) fp32 NCHW, the tensor
 * ppo_trainer.py:271-279,467-471 stores under "visual_features" when the encoder is frozen.  Uses the current training flag
 * (RunningMeanAndVar statistics are updated iff training, exactly like calling the module).  arch 1 only. */
int hab_policy_encode(hab_policy* p, const hab_obs* obs, int n, float* out, hipStream_t stream);
/* (C, Hf, Wf) of that tensor = ResNetEncoder.output_shape. */
int hab_policy_visual_feature_shape(const hab_policy* p, int* c, int* hf, int* wf);
/* actions == NULL -> get_value only.  hidden_*: (n, Lh, H), Lh = layers (GRU) / 2*layers (LSTM); hidden_out must not overlap hidden_in
 * (the episode-start mask is applied to hidden_in on the fly and the layers update the state through row strides). */
int hab_policy_act(hab_policy* p, const hab_obs* obs, const float* hidden_in, const uint8_t* masks,
                   const float* exp_noise, int deterministic, int n, float* values, int64_t* actions,
                   float* action_log_probs, float* hidden_out, float* probs_out /* (n,8) or NULL */, hipStream_t stream);
int hab_policy_evaluate(hab_policy* p, const hab_obs* obs, const int* rows, const float* hidden0, int hidden_env_stride,
                        const uint8_t* masks, const int64_t* actions, const hab_pack_info* pack, int B, int n,
                        float* value, float* log_prob, float* entropy, hipStream_t stream);
int hab_policy_final_hidden(hab_policy* p, float* hidden_out, hipStream_t stream);
int hab_policy_backward(hab_policy* p, const hab_obs* obs, const int* rows, const int64_t* actions,
                        const hab_pack_info* pack, const float* d_value, const float* d_log_prob, const float* d_entropy,
                        hipStream_t stream);

/* Auxiliary-loss hook.  NetPolicy.evaluate_actions hands `aux_loss_state` = {rnn_output, perception_embed} to the policy's
 * aux_loss_modules and adds their losses to the PPO loss (habitat_baselines/rl/ppo/policy.py:253-291,386-394, rl/ppo/ppo.py:248); autograd
 * then delivers gradients wrt those two tensors next to the heads' own.  d_rnn_output / d_perception_embed: device arrays [B][hidden] in
 * the frame order of the last hab_policy_evaluate (either may be NULL; the tensors themselves are HAB_TAP_RNN_OUT and the first `hidden`
 * columns of HAB_TAP_RNN_IN).  The NEXT hab_policy_backward adds them where the tensors sit in its chain and forgets them; the time-major
 * chunked form (rows-indirected minibatches of the fused updater) returns HAB_ERR_UNSUPPORTED, a blind net refuses d_perception_embed. */
int hab_policy_set_extra_grads(hab_policy* p, const float* d_rnn_output, const float* d_perception_embed);

/* HIP-event probe around one tagged call site (roofline measurement in bench.py). */
#define HAB_PROBE_CONV1_FWD 0
#define HAB_PROBE_CONV2_FWD 1
#define HAB_PROBE_CONV3_FWD 2
#define HAB_PROBE_FC_FWD 3
#define HAB_PROBE_CONV1_WGRAD 4
#define HAB_PROBE_CONV2_WGRAD 5
#define HAB_PROBE_CONV3_WGRAD 6
#define HAB_PROBE_CONV2_DGRAD 7
#define HAB_PROBE_CONV3_DGRAD 8
#define HAB_PROBE_FC_WGRAD 9
#define HAB_PROBE_FC_DGRAD 10
#define HAB_PROBE_ENC_FWD 11  /* arch 1: whole visual encoder forward (ingest .. visual_fc) */
#define HAB_PROBE_ENC_BWD 12  /* arch 1: whole visual encoder backward */
#define HAB_PROBE_RNN_FWD 13  /* packed-sequence recurrent layer forward of evaluate (one event pair per layer) */
#define HAB_PROBE_RNN_BWD 14  /* its BPTT */
#define HAB_PROBE_RN_WGRAD_IM2COL 15  /* arch 1: the weight gradients that run on the generic implicit-GEMM kernel (3x3 convolutions with
                                         >= 128 channels on either side and the strided ones: ResNet layer3 / layer4 / compression) */
int hab_policy_probe_enable(hab_policy* p, int tag /* -1 = off */);
/* several call sites at once: bit t of mask = HAB_PROBE_<t> (per-kernel table of bench.py, measured outside its timed region) */
int hab_policy_probe_enable_mask(hab_policy* p, uint64_t mask);
/* summed duration / number of bracketed calls: of ALL enabled sites (and forgets them), or of one tag (keeps them) */
int hab_policy_probe_read(hab_policy* p, double* total_ms, int* count);
int hab_policy_probe_read_tag(hab_policy* p, int tag, double* total_ms, int* count);
/* algorithmic work (FLOPs: 2 x M x N x K of each bracketed contraction) and bytes (operands read once + result written once) the call
 * sites of `tag` have launched since the probe was enabled; only the tags whose call sites report it (HAB_PROBE_RN_WGRAD_IM2COL) */
int hab_policy_probe_work(hab_policy* p, int tag, double* flops, double* bytes);

/* Test taps into the activation workspace of the last evaluate (NHWC). */
#define HAB_TAP_CONV1 0
#define HAB_TAP_CONV2 1
#define HAB_TAP_CONV3 2
#define HAB_TAP_RNN_IN 3
#define HAB_TAP_RNN_OUT 4
#define HAB_TAP_ENC_IN 5       /* arch 1: avg-pooled, normalised encoder input (NHWC, channels padded to 4) */
#define HAB_TAP_STEM 6
#define HAB_TAP_POOL 7
#define HAB_TAP_COMPRESSION 8
#define HAB_TAP_LAYER1 9       /* 9..12: output of stage 1..4 */
#define HAB_TAP_POOL_IDX 13    /* arch 1: arg-max bytes of the 3x3/2 max-pool, (B, Ho, Wo, C) uint8 = kh * 3 + kw (reinterpret the floats) */
#define HAB_TAP_CONV_OUT 100   /* arch 1: 100 + k = normalised (+ReLU / +residual+ReLU) output of backbone conv k, k in build order:
                                * per block its main-branch convs, then its downsample conv if it has one (resnet.py:37-69,116-152) */
int hab_policy_tap(hab_policy* p, int which, const float** ptr, int64_t* floats);

#ifdef __cplusplus
}
#endif
#endif /* HABITAT_AMD_H */
