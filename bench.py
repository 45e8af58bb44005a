#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the DD-PPO training hot path on synthetic PointNav RGB-D 256x256.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 runs in this process.  N > 1 needs one rank per GPU: when WORLD_SIZE is not in the environment the script re-launches itself
as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` (the form
the driver uses directly); ranks talk RCCL over xGMI (backend "nccl").

A "step" is ONE full cycle of the hot path on one batch of synthetic input: rollout collection of 64 envs x 128 steps (policy
forward per step) -> GAE -> PPO update (ppo_epoch x num_mini_batch forward + backward + clip + Adam) [-> gradient all-reduce per
minibatch when N > 1].  Workload at N=1 is BASELINE.json configs[1]: PointNav SimpleCNN+GRU, 64 envs x 128 steps, 256x256 RGB-D,
hyper-parameters of config/pointnav/ppo_pointnav_habitat_iccv19.yaml (E=4, M=4).  `value` = env-steps collected by all ranks /
max-over-ranks wall time, observations generated on the device (inputs never cross PCIe).

Extra objects on the JSON line (rank 0; the sub-records only at N = 1, all measured in THIS run, after the timed region):
  roofline      the call site with the largest share of the step ON THE CRITICAL STREAM -- chosen in this run from one untimed
                cycle with every probe on (`kernels`: ms, share, roofline position of every contraction call site), then HIP-event timed on
                its launch stream inside the timed region.  The recurrent layers run on the engine's second stream underneath the
                encoder (csrc/engine.hip): they are reported as `roofline.overlapped` (summed kernel time, launches, serial chain
                length) and are not candidates for the critical-stream site
  cpu_baseline  the CPU oracle restatement of the reference path over a FULL 64 envs x 128 steps update cycle (kind "port": the
                reference itself cannot travel to the GPU box; tests/golden pins the oracle to it), torch threads stated
  parity        the same rollout the CPU leg produced, pushed through the HIP path: relative error of the update's losses, of
                the GAE returns, and of one full-size minibatch's values / log-probs
  phases        rollout / update split of the timed cycles (HIP events around collect_rollout; also in the c3 sub-record)
  c3            BASELINE.json configs[2] (ResNet18 + 2-layer LSTM) cycles: env-steps/s, ms
  encoder_r18_b8192   the north-star kernel target: ResNet18 encoder alone on 2 x 4096 frames, forward / backward TFLOP/s and
                fraction of the fp32 MFMA peak
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL between ranks needs it)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E
NUM_ENVS, NUM_STEPS, OBS = 64, 128, 256

WORKLOADS = {
    "c2": dict(yaml="pointnav/ppo_pointnav_habitat_iccv19.yaml", name="PointNav SimpleCNN+GRU, 64 envs x 128 steps, 256x256 RGB-D synthetic",
               overrides=[]),
    # BASELINE.json configs[2]: ResNet18 + 2-layer LSTM with the ddppo_pointnav.yaml hyper-parameters (E=2, M=2)
    "c3": dict(yaml="pointnav/ddppo_pointnav.yaml", name="PointNav ResNet18+LSTM, 64 envs x 128 steps, 256x256 RGB-D synthetic",
               overrides=["habitat_baselines.rl.ddppo.backbone=resnet18"]),
    # configs[2] with rl.ddppo.train_encoder=False: the rollout stores `visual_features`, the update never runs the encoder
    "c3_frozen": dict(yaml="pointnav/ddppo_pointnav.yaml",
                      name="PointNav ResNet18+LSTM, frozen encoder (visual_features), 64 envs x 128 steps, 256x256 RGB-D synthetic",
                      overrides=["habitat_baselines.rl.ddppo.backbone=resnet18", "habitat_baselines.rl.ddppo.train_encoder=False"]),
    # BASELINE.json configs[4] (per GPU): ObjectNav ResNet50 on rgb + depth + semantic, 32 envs x 64 steps, E=4, M=2
    "c5": dict(yaml="objectnav/ddppo_objectnav.yaml", name="ObjectNav ResNet50+LSTM RGB-D+semantic, 32 envs x 64 steps, 256x256 synthetic",
               overrides=["habitat.simulator.sensors.semantic.height=256", "habitat.simulator.sensors.semantic.width=256"],
               envs=32, steps=64),
}

# probe name -> (HAB_PROBE tag, algorithmic flops per frame)   SimpleCNN @256^2 RGB-D (a4)
PROBES = {
    "conv1_fwd": (0, 2.0 * 63 * 63 * 32 * 256), "conv2_fwd": (1, 2.0 * 30 * 30 * 64 * 512), "conv3_fwd": (2, 2.0 * 28 * 28 * 32 * 576),
    "fc_fwd": (3, 2.0 * 25088 * 512), "conv1_wgrad": (4, 2.0 * 63 * 63 * 32 * 256), "conv2_wgrad": (5, 2.0 * 30 * 30 * 64 * 512),
    "conv3_wgrad": (6, 2.0 * 28 * 28 * 32 * 576), "conv2_dgrad": (7, 2.0 * 30 * 30 * 64 * 512), "conv3_dgrad": (8, 2.0 * 28 * 28 * 32 * 576),
    "fc_wgrad": (9, 2.0 * 25088 * 512), "fc_dgrad": (10, 2.0 * 25088 * 512),
    # ResNet18 encoder @256^2 RGB-D (a5): 168.82 MMAC forward; backward = dgrad + wgrad of every conv except the
    # stem's dgrad (25.69 MMAC)
    "enc_fwd": (11, 2.0 * 168.82e6), "enc_bwd": (12, 2.0 * (2 * 168.82e6 - 25.69e6)),
    # recurrent layers (packed sequence forward / BPTT): input projection + recurrence, GRU 1 x 512 on a 514-wide input
    "rnn_fwd": (13, 2.0 * 3 * 512 * (514 + 512)), "rnn_bwd": (14, 2.0 * 2 * 3 * 512 * (514 + 512)),
}
C2_TABLE = ["conv1_fwd", "conv2_fwd", "conv3_fwd", "fc_fwd", "fc_wgrad", "fc_dgrad", "conv3_wgrad", "conv3_dgrad", "conv2_wgrad", "conv2_dgrad",
            "conv1_wgrad", "rnn_fwd", "rnn_bwd"]
# kernel launched by a probed call site (rocprofv3 names), for the HBM-traffic lookup in the committed --pmc passes
PROBE_KERNELS = {"conv2_dgrad": "conv2_dgrad_strip_kernel", "conv1_fwd": "obs_conv_patch_kernel",
                 "conv1_wgrad": "obs_wgrad_bf3_kernel", "conv2_wgrad": "wgrad3x3_bf3_kernel<1, 2, 63",
                 "conv3_wgrad": "wgrad3x3_bf3_kernel<2, 1, 30", "conv2_fwd": "conv2_fwd_strip_kernel",
                 "conv3_fwd": "conv_patch_bf3_kernel<ConvFwdProb", "conv3_dgrad": "igemm_bf3_kernel<ConvDgradProb",
                 "fc_fwd": "dense_bf3_kernel<0, 0>", "fc_dgrad": "dense_bf3_kernel<0, 1>",
                 "fc_wgrad": "dense_bf3_kernel<1, 1>"}
# Roofline model of a contraction call site on the split-bf16 matrix path (csrc/igemm_bf3.h): (algorithmic HBM bytes per frame:
# every operand read once, the result written once; bf16 MFMA flops issued per useful fp32 flop).  The fp32-equivalent MFMA ceiling
# of a site is PEAK_BF16 / factor: 6 partial products in general; the observation-ingest convolutions need 3 for the uint8 rgb
# operand (exact in one bf16 plane) and 6 for depth = 3.75 on the 3:1 channel mix (csrc/obs_conv_bf3.h, obs_wgrad_bf3.h).
_A0, _A1, _A2, _A3 = 256 * 256 * 7, 63 * 63 * 32 * 4, 30 * 30 * 64 * 4, 28 * 28 * 32 * 4  # obs (u8 rgb + f32 depth), conv1..3 outputs
SITE_MODEL = {
    "conv1_fwd": (_A0 + _A1, 3.75), "conv1_wgrad": (_A0 + _A1, 3.75),
    "conv2_fwd": (_A1 + _A2, 6.0), "conv2_wgrad": (_A1 + _A2, 6.0), "conv2_dgrad": (_A2 + 2 * _A1, 6.0),  # dgrad reads the ReLU mask
    "conv3_fwd": (_A2 + _A3, 6.0), "conv3_wgrad": (_A2 + _A3, 6.0), "conv3_dgrad": (_A3 + 2 * _A2, 6.0),
    # (no ReLU behind conv3 -- simple_cnn.py:84-86 has it commented out --, so the fc data gradient reads no mask: dY in, dX out)
    "fc_fwd": (_A3 + 2048, 6.0), "fc_wgrad": (_A3 + 2048, 6.0), "fc_dgrad": (_A3 + 2048, 6.0),
}


# Bytes per frame the COMPUTATION needs, where that is less than what the kernel is written to read: the data gradients read the
# producing layer's fp32 activation only for its sign -- a 1-bit mask would do (conv2_dgrad: 63*63*32 bits = 15.9 KB instead of 508 KB).
# `frac` is quoted on the bytes the kernel moves (SITE_MODEL, what the counters are compared with); `frac_of_bytes_needed` says where the
# kernel stands against the leaner formulation, so that `frac` cannot flatter a kernel for traffic it chose.
SITE_NEEDED_BYTES = {"conv2_dgrad": _A2 + _A1 + 63 * 63 * 32 // 8, "conv3_dgrad": _A3 + _A2 + 30 * 30 * 64 // 8}


# algorithmic bytes of a call site that do not scale with the frames of a launch: the 25088 x 512 weight matrix of the visual fc (read by the
# forward and the data gradient, written once by the weight gradient) -- 51 MB per launch, as much as 2000 frames of its activations
SITE_LAUNCH_BYTES = {"fc_fwd": 25088 * 512 * 4, "fc_dgrad": 25088 * 512 * 4, "fc_wgrad": 25088 * 512 * 4}


def site_roofline(site, flops_per_frame, frames, ms):
    """Which roofline bounds the call site (time per frame at the HBM peak vs at the split-bf16 MFMA ceiling) and where it is."""
    tfl = flops_per_frame * frames / (ms * 1e-3) / 1e12
    if site not in SITE_MODEL:
        return {"bound": "mfma", "achieved": round(tfl, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / PEAK_FP32_MFMA_TFLOPS, 4)}
    bytes_pf, factor = SITE_MODEL[site]
    peak_eq = PEAK_BF16_MFMA_TFLOPS / factor
    gbs = bytes_pf * frames / (ms * 1e-3) / 1e9
    t_hbm, t_mfma = bytes_pf / (PEAK_HBM_GBS * 1e9), flops_per_frame / (peak_eq * 1e12)
    r = {"bound": "hbm" if t_hbm >= t_mfma else "mfma"}
    if r["bound"] == "hbm":
        r.update(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
    else:
        r.update(achieved=round(tfl, 2), peak=round(peak_eq, 1), unit="TFLOP/s", frac=round(tfl / peak_eq, 4))
    r["fp32_equiv_tflops"] = round(tfl, 2)
    r["frac_of_split_ceiling"] = round(tfl / peak_eq, 4)  # against the matrix pipe in use, whichever roofline bounds the site
    r["algorithmic_bytes_per_frame"] = bytes_pf
    r["algorithmic_gbs"] = round(gbs, 1)
    r["mfma_ceiling_fp32_equiv_tflops"] = round(peak_eq, 1)
    r["frac_of_fp32_mfma_peak"] = round(tfl / PEAK_FP32_MFMA_TFLOPS, 4)  # the round-1 yardstick (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s)
    r["roofline_floor_ms_per_kframe"] = round(max(t_hbm, t_mfma) * 1e6, 4)
    if site in SITE_NEEDED_BYTES:
        need = SITE_NEEDED_BYTES[site]
        t_need = max(need / (PEAK_HBM_GBS * 1e9), t_mfma)  # the roofline of the leaner formulation (may turn MFMA-bound)
        r["algorithmic_bytes_needed"] = need
        r["frac_of_bytes_needed"] = round(t_need / (ms * 1e-3 / frames), 4)
        r["bound_with_bytes_needed"] = "hbm" if need / (PEAK_HBM_GBS * 1e9) >= t_mfma else "mfma"
    return r


# call sites whose kernel runs a FIXED (persistent) grid: the counter passes cannot tell its rollout-sized launches from the update-sized
# ones (tools/pmc_traffic.py keys by kernel and grid), so their traffic figure is the mean over all launches and is compared with the
# algorithmic bytes of the mean launch
PERSISTENT_GRID_SITES = {"conv1_fwd", "conv2_fwd"}


def site_chunks(site):
    """Launches of a call site per minibatch: the time-major chunked recurrence (csrc/engine.hip, HAB_RNN_CHUNKS, default 4) runs the
    forward sites and the data-gradient chain (fc / conv3 / conv2 dgrad) once per time chunk; weight gradients once per minibatch."""
    chunks = int(os.environ.get("HAB_RNN_CHUNKS", "4")) or 1
    return chunks if (site.endswith("_fwd") or site.endswith("_dgrad")) else 1


def make_trainer(workload: str, total_updates: int, envs: int = 0, steps: int = 0, extra_overrides=()):
    from habitat_amd.config.default import get_config
    import habitat_amd.rl.ppo.ppo_trainer as tr
    w = WORKLOADS[workload]
    cfg = get_config(w["yaml"], [f"habitat_baselines.num_environments={envs or w.get('envs', NUM_ENVS)}",
                                 f"habitat_baselines.rl.ppo.num_steps={steps or w.get('steps', NUM_STEPS)}",
                                 f"habitat_baselines.num_updates={total_updates}", "habitat_baselines.total_num_steps=-1",
                                 "habitat_baselines.num_checkpoints=-1", f"habitat_baselines.checkpoint_interval={10 ** 9}",
                                 "habitat_baselines.rl.ddppo.distrib_backend=" + os.environ.get("HAB_BENCH_DISTRIB_BACKEND", "NCCL"),
                                 "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000",
                                 "habitat_baselines.checkpoint_folder=/tmp/habitat_amd_bench_ckpt",
                                 f"habitat.simulator.sensors.rgb.height={OBS}", f"habitat.simulator.sensors.rgb.width={OBS}",
                                 f"habitat.simulator.sensors.depth.height={OBS}", f"habitat.simulator.sensors.depth.width={OBS}"] + w["overrides"]
                     + list(extra_overrides) + [o for o in os.environ.get("HAB_BENCH_OVERRIDES", "").split(",") if o])  # development hook
    trainer = tr.PPOTrainer(cfg)
    return trainer, cfg


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU leg + parity leg (rank 0, N = 1)
# ---------------------------------------------------------------------------------------------------------------------------------
def cpu_baseline_and_parity(trainer, cfg, sample_envs=NUM_ENVS, sample_steps=NUM_STEPS, parity=True, cpu_threads=0, workload="c2"):
    """(a) `cpu_baseline`: the oracle (oracle/functional.py: CPU restatement of the reference PPOTrainer path, pinned to the live
    reference by tests/golden) runs ONE full update cycle of the workload -- rollout of `sample_envs` x `sample_steps` with policy.act
    per step, GAE, PPO update E x M -- on the host cores and is timed.  (b) `parity`: the rollout the oracle produced (observations,
    actions, old log-probs / values, rewards, masks, hidden states) is loaded into a device RolloutStorage and the HIP path does the
    same update with the same minibatch permutations from the same parameters: GAE returns, one minibatch's per-frame outputs, and
    the whole update twice (oracle/parity.py::update_parity) -- TEACHER-FORCED (every minibatch step starts from the oracle's
    pre-step parameters + Adam moments: per-step loss / gradient-norm / parameter-step errors that chaos cannot amplify) and
    free-running (how the trainer runs it: the drift before each step and the clip decisions that flipped)."""
    import types
    import numpy as np
    import torch
    from oracle import parity as PR
    # default 16 threads: more slow the small-batch CPU convolutions of the rollout down (both settings were timed:
    # profiles/r03_cpu_leg_threads.json, --cpu-threads)
    threads = cpu_threads if cpu_threads > 0 else min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    N, T = sample_envs, sample_steps
    pol = trainer._agent.actor_critic
    hidden, hl = pol.recurrent_hidden_size, pol.num_recurrent_layers
    params = {k: v.detach().cpu().clone() for k, v in pol.state_dict().items()}  # the policy as the timed cycles left it
    spec = PR.spec_of(pol)
    trainable = [k for k, p_ in pol.named_parameters() if p_.requires_grad]
    ppo = cfg.habitat_baselines.rl.ppo
    ocfg = types.SimpleNamespace(clip_param=ppo.clip_param, ppo_epoch=ppo.ppo_epoch, num_mini_batch=ppo.num_mini_batch,
                                 value_loss_coef=ppo.value_loss_coef, entropy_coef=ppo.entropy_coef, lr=ppo.lr, eps=ppo.eps,
                                 max_grad_norm=ppo.max_grad_norm, use_normalized_advantage=ppo.use_normalized_advantage,
                                 use_clipped_value_loss=ppo.use_clipped_value_loss, gamma=ppo.gamma, tau=ppo.tau)
    t0 = time.perf_counter()
    buf, nv, perms, t_env = PR.oracle_rollout(params, spec, N, T, OBS, OBS, hidden, hl, ocfg, task="objectnav" if workload == "c5" else "pointnav")
    ref_metrics, trace, final = PR.oracle_update_trace(params, spec, buf, T, ocfg, trainable, perms)
    dt = time.perf_counter() - t0 - t_env
    base = {"value": round(N * T / dt, 2), "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"one FULL update cycle of the workload ({N} envs x {T} steps, 256x256 RGB-D, {WORKLOADS[workload]['name'].split(',')[0]}, "
                      f"E={ocfg.ppo_epoch} x M={ocfg.num_mini_batch}): oracle/functional.py (CPU restatement of the reference path, pinned to the "
                      f"reference by tests/golden; the reference itself is absent on the GPU box) on torch-CPU fp32 with {threads} threads of "
                      f"{os.cpu_count()} host cores, {dt:.1f} s (synthetic obs generation {t_env:.1f} s excluded; the per-step state snapshots "
                      f"of the parity leg, ~0.1 s each, included)"}
    if not parity:
        return base, None
    # ---- the same rollout through the HIP path ----------------------------------------------------------------------------------
    from habitat_amd.common.rollout_storage import MiniBatch
    from habitat_amd.rl.ppo import PPO
    es = trainer._env_spec
    trainer._agent._rollouts = None  # release the trainer's arena before allocating this one
    torch.cuda.empty_cache()
    st = PR.storage_from_oracle(buf, nv, T, N, es.observation_space, es.action_space, pol, trainer.device, ocfg)  # GAE: the scan kernel
    B = st.buffers
    out = {"returns_max_rel": PR.rel(B["returns"].cpu().numpy()[:T], buf["returns"].numpy()[:T])}
    pol.load_state_dict(params)
    pol.train()
    upd = PPO.from_config(pol, ocfg)
    if workload == "c2":  # one minibatch's per-frame outputs against the chunked oracle evaluation (the c3 / c5 legs read them off the trace)
        adv = upd.get_advantages(st)
        batch = MiniBatch(st, perms[0][0], T, adv, torch.logical_not(B["masks"]).cpu().view(-1, N).numpy())  # first minibatch of epoch 0
        mb = PR.minibatch_parity(pol, upd, st, batch, ocfg, env_chunk=8, with_grads=False)
        out.update({"minibatch_frames": mb["frames"], "value_max_rel": mb["value_max_rel"], "log_prob_max_rel": mb["log_prob_max_rel"],
                    "minibatch_value_loss_rel": mb["value_loss_rel"], "minibatch_action_loss_rel": mb["action_loss_rel"]})
        pol.load_state_dict(params)  # (RunningMeanAndVar-free policy: nothing changed; kept for symmetry with the ResNet leg)
    up = PR.update_parity(pol, upd, st, buf, trace, final, T, ocfg, trainable)
    out["teacher_forced"] = up["teacher_forced"]
    out["teacher_forced_max_rel"] = up["teacher_forced"]["max_rel"]
    out["free_running"] = up["free_running"]
    fr = up["free_running"]
    # the four figures earlier rounds' lines carried under these names (free-running, means over the update's steps)
    for k in ("value_loss", "action_loss", "dist_entropy", "grad_norm"):
        out[k + "_rel"] = fr[k + "_rel_of_update_means"]
    out["post_update_param_max_abs_diff"] = fr["post_update_param_max_abs_diff"]
    # the oracle's own figures beside the relative errors: the action loss of a normalised-advantage minibatch is a mean near zero
    # (|.| ~ 1e-3), so its RELATIVE error is an absolute error of ~1e-9 .. 1e-6 divided by that
    out["reference"] = {k: float(f"{ref_metrics[k]:.6e}") for k in ("value_loss", "action_loss", "dist_entropy", "grad_norm")}
    out["what"] = (f"HIP path vs the CPU oracle on the SAME {N} x {T} rollout (the one the oracle produced), same parameters, same minibatch "
                   f"permutations, {len(trace)} minibatch steps of {T * N // ocfg.num_mini_batch} frames.  teacher_forced: every step "
                   f"restarted from the oracle's pre-step parameters and Adam moments (per-step arrays; max_rel = worst of value loss, "
                   f"action loss / mean |surrogate|, entropy, gradient norm; bar 1e-4).  free_running: the update as the trainer runs "
                   f"it; `*_rel` = relative error of the update's MEAN figures (what earlier rounds reported), per-step drift and flipped "
                   f"clip decisions explain it.  GAE returns: scan kernel")
    out = {k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in out.items()}
    return base, out


def c3_parity_record(state, envs=8, steps=NUM_STEPS, workload="c3"):
    """Whole-update parity of the ResNet policies on a BOUNDED sample, from the parameters the sub-record's cycles left (`state`);
    RunningMeanAndVar updates every step.  c3 (BASELINE.json configs[2], ResNet18 + 2-layer LSTM): `envs` x 128 steps of 256x256 RGB-D,
    E = 2 x M = 2 -> 4 minibatch steps of envs / 2 x 128 frames (the CPU oracle needs ~20 s per 1000 frames of forward + backward).
    c5 (configs[4], ObjectNav ResNet50 on rgb + depth + semantic with objectgoal / compass / gps embeddings): 2 envs x 64 steps,
    E = 4 x M = 2 -> 8 steps of 64 frames."""
    import torch
    trainer, cfg = make_trainer(workload, 2, envs=envs, steps=steps)
    trainer._init_train()
    trainer._agent.actor_critic.load_state_dict(state)
    t0 = time.perf_counter()
    base, par = cpu_baseline_and_parity(trainer, cfg, sample_envs=envs, sample_steps=steps, workload=workload)
    par["oracle_and_hip_seconds"] = round(time.perf_counter() - t0, 1)
    par["sample"] = f"{envs} envs x {steps} steps (bounded: the full update takes the CPU oracle minutes)"
    base["sample"] = (f"BOUNDED sample of the workload: {envs} envs x {steps} steps instead of the full rollout (same observation size, network, E and M; "
                      f"the CPU oracle needs minutes for the full update), otherwise as the headline leg: ") + base["sample"]
    trainer.envs.close()
    del trainer
    torch.cuda.empty_cache()
    return par, base


def encoder_record_one_call(frames=8192, timeout_s=240):
    """The north-star kernel target as ONE 8192-frame call (an engine whose arenas are sized for 8192 frames: ~46 GB of activations and
    gradients), in a CHILD process: a batch twice the size any training minibatch or test reaches is run where a fault cannot take
    the benchmark line with it.  Falls back to 2 x 4096 in this process, with the reason in the record."""
    cmd = [sys.executable, os.path.abspath(__file__), "--encoder-only", str(frames)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        why = f"exit code {r.returncode}: {r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else 'no output'}"
    except subprocess.TimeoutExpired:
        why = f"no result within {timeout_s} s"
    rec = encoder_record(4096, 2)
    rec["one_call_of_8192"] = f"failed ({why}); this record is 2 calls of 4096 frames"
    return rec


def encoder_record(frames=4096, calls=2):
    """ResNet18 encoder alone (ingest .. visual_fc) at `calls` x `frames` frames of 256x256 RGB-D: HIP events around the engine's encoder
    call sites, algorithmic FLOPs of SURVEY.md 8(d) (forward 0.3376 GFLOP/frame; backward = dgrad + wgrad without the stem's dgrad)."""
    import numpy as np
    import torch
    from habitat_amd.engine import DevicePackInfo, PolicyEngine
    B, T = frames, 128
    n = B // T
    eng = PolicyEngine(arch="resnet", backbone=18, baseplanes=32, normalize_visual_inputs=True, rnn_type="LSTM", rnn_layers=2, hidden=512,
                       H=256, W=256, max_frames=B, max_envs=64)
    g = torch.Generator(device="cuda").manual_seed(0)
    eng.params_flat.copy_(torch.randn(eng.params_flat.shape, device="cuda", generator=g) * 0.05)
    for nm, v in eng.views.items():
        if v.dim() == 1 and (nm.endswith(".1.weight") or nm.endswith(".4.weight") or nm.endswith(".7.weight")):
            v.fill_(1.0)
    eng.repack()
    rgb = torch.randint(0, 256, (B, 256, 256, 3), dtype=torch.uint8, device="cuda", generator=g)
    depth = torch.rand(B, 256, 256, 1, device="cuda", generator=g)
    goal = torch.rand(B, 2, device="cuda", generator=g)
    masks = torch.ones(B, 1, dtype=torch.bool, device="cuda")
    actions = torch.zeros(B, 1, dtype=torch.long, device="cuda")
    h0 = torch.zeros(n, 4, 512, device="cuda")
    pack = DevicePackInfo(np.zeros((T, n), np.uint8), "cuda")
    dv = torch.randn(B, device="cuda", generator=g) * 1e-3

    def cycle():
        eng.evaluate(rgb, depth, goal, None, h0, masks, actions, pack, B, n, prev_actions=actions)
        eng.backward(rgb, depth, goal, None, actions, pack, dv, dv, dv, prev_actions=actions)

    cycle()
    torch.cuda.synchronize()
    eng.probe_enable_mask([11, 12])
    for _ in range(calls):
        cycle()
    torch.cuda.synchronize()
    fwd, bwd = (eng.probe_read_tag(t)[0] / calls for t in (11, 12))
    eng.probe_enable(-1)
    f_fwd, f_bwd = PROBES["enc_fwd"][1], PROBES["enc_bwd"][1]
    tf = lambda fl, ms: B * fl / ms / 1e9
    rec = {"frames": B * calls, "frames_per_call": B, "forward_ms": round(fwd, 2), "backward_ms": round(bwd, 2),
           "forward_tflops": round(tf(f_fwd, fwd), 1), "backward_tflops": round(tf(f_bwd, bwd), 1),
           "fwd_bwd_tflops": round(tf(f_fwd + f_bwd, fwd + bwd), 1), "peak_tflops": PEAK_FP32_MFMA_TFLOPS,
           "forward_frac": round(tf(f_fwd, fwd) / PEAK_FP32_MFMA_TFLOPS, 4), "backward_frac": round(tf(f_bwd, bwd) / PEAK_FP32_MFMA_TFLOPS, 4),
           "fwd_bwd_frac": round(tf(f_fwd + f_bwd, fwd + bwd) / PEAK_FP32_MFMA_TFLOPS, 4),
           "split_bf16_ceiling_tflops": round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1),
           "fwd_bwd_frac_of_split_bf16_ceiling": round(tf(f_fwd + f_bwd, fwd + bwd) / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4),
           "what": "ResNet18 GroupNorm encoder (ingest, RunningMeanAndVar, 17 convs + GroupNorms, compression, visual_fc), fp32, 256x256 RGB-D; "
                   "every non-contraction kernel (GroupNorm, pooling, ingest) is inside the times"}
    del eng
    torch.cuda.empty_cache()
    return rec


def time_rollouts(trainer):
    """Brackets every collect_rollout of `trainer` with a HIP event pair on the stream the engine launches on (no synchronisation is
    added); the returned function gives the mean GPU time of a rollout in ms once the stream has been synchronised."""
    import torch
    pairs = []
    inner = trainer.collect_rollout

    def timed():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = inner()
        e1.record()
        pairs.append((e0, e1))
        return r

    trainer.collect_rollout = timed

    def mean_ms(last_n):
        sel = pairs[-last_n:]
        return sum(a.elapsed_time(b) for a, b in sel) / max(1, len(sel))

    return mean_ms


def phase_record(rollout_ms, cycle_ms, ppo):
    upd = cycle_ms - rollout_ms
    nmb = ppo.ppo_epoch * ppo.num_mini_batch
    return {"rollout_ms": round(rollout_ms, 2), "update_ms": round(upd, 2), "minibatches": nmb, "update_ms_per_minibatch": round(upd / nmb, 2),
            "rollout_share": round(rollout_ms / cycle_ms, 4),
            "how": "HIP events around collect_rollout on the engine's stream (mean over the timed cycles); update = cycle - rollout "
                   "(GAE, E x M minibatch passes, Adam, statistics)"}


RN_WGRAD_PROBE = 15  # HAB_PROBE_RN_WGRAD_IM2COL (include/habitat_amd.h)
RN_WGRAD_KERNEL = "igemm_bf3_kernel<ConvWgradProb, 2, 2, 2, 2>"


def pmc_kernel_traffic(workload, kernel_substr):
    """(mean HBM bytes per launch, calls, file) of a kernel in the newest committed --pmc traffic file of the workload (tools/pmc_traffic.py)."""
    for tag in ("r06", "r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{tag}_{workload}_hbm_traffic.json")
        if not os.path.exists(path):
            continue
        tot, calls = 0.0, 0
        for k, r in json.load(open(path)).items():
            if kernel_substr in k:
                tot += r["calls"] * (r["fetch_bytes_per_call"] + r["write_bytes_per_call"])
                calls += r["calls"]
        if calls:
            return tot / calls, calls, os.path.relpath(path, ROOT)
    return None, 0, None


def resnet_site_roofline(eng, workload, cycle_ms_total):
    """`roofline` of a ResNet sub-record: the weight gradients on the generic implicit-GEMM kernel -- the largest single kernel of the
    C3 cycle by summed time (profiles/r0x_c3_kernel_stats.txt) -- HIP-event bracketed per call (incl. its split-K second pass) inside the
    timed cycles, priced with the work the engine reports for exactly those calls."""
    ms, cnt = eng.probe_read_tag(RN_WGRAD_PROBE)
    flops, nbytes = eng.probe_work(RN_WGRAD_PROBE)
    if not cnt or ms <= 0:
        return None
    tfl = flops / (ms * 1e-3) / 1e12
    peak_eq = PEAK_BF16_MFMA_TFLOPS / 6.0
    traffic, pcalls, src = pmc_kernel_traffic(workload, RN_WGRAD_KERNEL)
    rec = {"bound": "mfma", "kernel": f"{RN_WGRAD_KERNEL} + split-K second pass (ResNet layer3 / layer4 / compression and strided 3x3 weight gradients)",
           "launches": cnt, "avg_launch_ms": round(ms / cnt, 4), "achieved": round(tfl, 2), "peak": round(peak_eq, 1), "unit": "TFLOP/s",
           "frac": round(tfl / peak_eq, 4), "frac_of_fp32_mfma_peak": round(tfl / PEAK_FP32_MFMA_TFLOPS, 4),
           "share_of_step": round(ms / cycle_ms_total, 4), "algorithmic_bytes_per_launch": round(nbytes / cnt),
           "algorithmic_gflop_per_launch": round(flops / cnt / 1e9, 3), "traffic": round(traffic) if traffic else None, "traffic_source": src,
           "traffic_ratio": round(traffic / (nbytes / cnt), 3) if traffic else None,
           "how": "HIP events around every bracketed call on the launch stream inside the timed cycles (probe HAB_PROBE_RN_WGRAD_IM2COL); FLOPs = "
                  "2 x pixels x Cout x 9 Cin and bytes = x + dY + dW once, both summed by the engine over the bracketed calls; traffic = mean FETCH_SIZE + "
                  "WRITE_SIZE per launch of the kernel in the committed counter pass (all its launches, the split-K pass not included)"}
    return rec


def run_cycles(workload, steps, warmup, keep_state=None, distributed=False, keep_flat=None):
    """A second workload inside the same run (sub-record): (env-steps/s, ms per cycle).  keep_state: dict that receives a CPU copy of
    the policy's state_dict as the cycles left it (the c3 parity leg starts from it)."""
    import torch
    # keep_flat (the exchange A/B): both trainers must collect IDENTICAL rollouts, so DD-PPO's preemptive straggler rule (a rank ends its
    # rollout early once sync_frac of the ranks are done: timing-dependent) is switched off for these runs (sync_frac > 1 never triggers)
    trainer, cfg = make_trainer(workload, warmup + steps + 1,
                                extra_overrides=["habitat_baselines.rl.ddppo.sync_frac=2.0"] if keep_flat is not None else ())
    trainer._init_train()
    rollout_ms = time_rollouts(trainer)
    def sync():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        trainer.run_update_cycle()
    sync()
    eng_ = trainer._agent.actor_critic.engine
    site = workload in ("c3", "c5") and not distributed
    if site:
        eng_.probe_enable(RN_WGRAD_PROBE)
    s0 = trainer.num_steps_done
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.run_update_cycle()
    sync()
    dt = time.perf_counter() - t0
    site_rec = resnet_site_roofline(eng_, workload, dt * 1e3) if site else None
    if site:
        eng_.probe_read()
        eng_.probe_enable(-1)
    del eng_
    if distributed:  # max over ranks, like the headline figure; num_steps_done is the all-reduced counter
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        trainer.shutdown()
    n = trainer.num_steps_done - s0
    trainer.envs.close()
    if keep_state is not None:
        keep_state.update({k: v.detach().cpu().clone() for k, v in trainer._agent.actor_critic.state_dict().items()})
    if keep_flat is not None:  # the parameter arena as the cycles left it + which exchange carried the gradients (exchange A/B at N > 1)
        keep_flat["params"] = trainer._agent.actor_critic.engine.params_flat.detach().clone()
        nc = getattr(trainer._agent.updater, "_native_comm", None)
        keep_flat["comm"] = "rccl-native" if nc is not None else "torch-callbacks"
    world = torch.distributed.get_world_size() if distributed else 1
    rec = {"workload": WORKLOADS[workload]["name"] + (f", DD-PPO x {world} ranks" if distributed else ""), "n_gpus": world,
           "value": round(n / dt, 1), "unit": "env-steps/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 2), "phases": phase_record(rollout_ms(steps), dt / steps * 1e3, cfg.habitat_baselines.rl.ppo),
           "frac_of_mfma_roofline": round(n / dt / world * 2.2632e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4) if workload == "c3" else None,
           "frac_of_split_ceiling": round(n / dt / world * 2.2632e9 / (PEAK_BF16_MFMA_TFLOPS / 6.0 * 1e12), 4) if workload == "c3" else None,
           "frac_basis": "executed contraction FLOPs (2.263 GFLOP per env-step: the stem's data gradient is not computed) / fp32 MFMA peak 157.3; "
                         "frac_of_split_ceiling = the same FLOPs / (bf16 MFMA peak / 6 partial products = 416.7)"}
    if site_rec:
        rec["roofline"] = site_rec
    del trainer
    torch.cuda.empty_cache()
    return rec


def hbm_traffic(workload, probe):
    """HBM bytes per probed call from the committed rocprofv3 --pmc passes (profiles/r0x_c2_hbm_traffic.json, produced by
    tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE collected in separate runs of this script, FETCH_SIZE doubled per the gfx950
    note of MI355X_MICROARCH.md).  Hardware counters cannot be read from inside the timed run, so this is the profiled figure of the
    same command (the newest committed round); null when the probed call site has no entry."""
    if workload != "c2" or probe not in PROBE_KERNELS:
        return None, None
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_c2_hbm_traffic.json")
        if not os.path.exists(path):
            continue
        best = None
        for k, r in json.load(open(path)).items():
            if PROBE_KERNELS[probe] in k:
                tot = r["fetch_bytes_per_call"] + r["write_bytes_per_call"]
                if best is None or r["calls"] * tot > best[0]:  # the update-sized launch when one kernel serves several sizes
                    best = (r["calls"] * tot, tot)
        if best:
            return round(best[1]), os.path.relpath(path, ROOT)
    return None, None


def relaunch_ranks(a):
    """--gpus N > 1 without a torchrun environment: start the N ranks ourselves (same command line the driver uses)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--probe", default=None, choices=list(PROBES))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg, the parity leg and the sub-records (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the c3 / encoder sub-records only")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU leg (default min(16, host cores); -1 = every host core)")
    ap.add_argument("--encoder-only", type=int, default=0, help="print the ResNet18 encoder record of ONE call of this many frames and exit")
    a = ap.parse_args()
    if a.encoder_only:
        print(json.dumps(encoder_record(a.encoder_only, 1)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_ranks(a)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop WORLD_SIZE and let bench.py start them)")
    import torch
    auto_probe = a.probe is None and a.workload == "c2" and world == 1  # site chosen from the all-probes cycle below
    if a.probe is None:
        a.probe = "conv1_fwd" if a.workload == "c2" else ("enc_fwd" if a.workload == "c3_frozen" else "enc_bwd")
    if a.workload == "c5" and a.probe.startswith("enc_"):
        # ResNet50 on 5 channels: 375.0 MMAC forward (a5); the stem's data gradient (7x7x8 pad -> 5 real ch) is not computed
        PROBES["enc_fwd"], PROBES["enc_bwd"] = (11, 2.0 * 375.0e6), (12, 2.0 * (2 * 375.0e6 - 32.1e6))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU execution path")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    trainer, cfg = make_trainer(a.workload, a.warmup + a.steps + 2)
    trainer._init_train()
    eng = trainer._agent.actor_critic.engine
    dist = torch.distributed.is_initialized()

    def barrier():
        if dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.run_update_cycle()
    ppo = cfg.habitat_baselines.rl.ppo
    n_envs, n_steps = WORKLOADS[a.workload].get("envs", NUM_ENVS), WORKLOADS[a.workload].get("steps", NUM_STEPS)

    def frames_seen(probe, local_steps, cycles):  # frames that passed a call site: E update passes (+ rollout + bootstrap for forward sites)
        f = local_steps * ppo.ppo_epoch
        return f + (local_steps + n_envs * cycles if probe.endswith("_fwd") and not probe.startswith("rnn") else 0)

    table, overlapped = None, None
    if a.workload == "c2" and world == 1:
        # per-call-site table from ONE untimed cycle with every probe on (~1400 event records), BEFORE the timed region: it decides which
        # site the timed probe brackets.  Single-rank runs only (an extra cycle per rank is fine, but the table is rank 0's business).
        tags = {k: PROBES[k] for k in C2_TABLE}
        eng.probe_enable_mask([t for t, _ in tags.values()])
        l0 = trainer.local_steps_done
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        trainer.run_update_cycle()
        torch.cuda.synchronize()
        cyc_ms = (time.perf_counter() - c0) * 1e3
        ls = trainer.local_steps_done - l0
        table = []
        for k, (t_, fl) in tags.items():
            ms, cnt = eng.probe_read_tag(t_)
            if cnt:
                fs = frames_seen(k, ls, 1)
                sr = site_roofline(k, fl, fs, ms)
                row = {"site": k, "ms": round(ms, 2), "calls": cnt, "share": round(ms / cyc_ms, 4),
                       "tflops": sr.get("fp32_equiv_tflops", sr["achieved"]), "bound": sr["bound"], "achieved": sr["achieved"],
                       "peak": sr["peak"], "unit": sr["unit"], "frac": sr["frac"], "stream": "second (overlapped)" if k.startswith("rnn") else "critical"}
                if "frac_of_split_ceiling" in sr:
                    row["frac_of_split_ceiling"] = sr["frac_of_split_ceiling"]
                # HBM bytes of the update-sized launch (committed --pmc passes of this command) over its algorithmic bytes
                tr, _ = hbm_traffic(a.workload, k)
                if tr is not None and k in SITE_MODEL:
                    upd_frames = fs / cnt if k in PERSISTENT_GRID_SITES else n_envs * n_steps // ppo.num_mini_batch // site_chunks(k)
                    row["traffic"] = tr
                    row["traffic_ratio"] = round(tr / (SITE_MODEL[k][0] * upd_frames + SITE_LAUNCH_BYTES.get(k, 0)), 3)
                table.append(row)
        eng.probe_read()
        eng.probe_enable(-1)
        table.sort(key=lambda r: -r["ms"])
        rnn = [r for r in table if r["site"].startswith("rnn")]
        if rnn:
            # the recurrence of a minibatch is a chain of dependent step launches (T forward + T backward per layer) on the second stream
            overlapped = {"sites": rnn, "summed_kernel_ms_per_cycle": round(sum(r["ms"] for r in rnn), 2),
                          # one persistent launch per layer, time chunk and direction (csrc/rnn_persist.h); one launch per step with HAB_RNN_PERSIST=0
                          "serial_chain_launches_per_cycle": int(ppo.ppo_epoch * ppo.num_mini_batch * 2 *
                                                                 (n_steps if os.environ.get("HAB_RNN_PERSIST") == "0" else site_chunks("rnn_fwd"))),
                          "serial_steps_per_cycle": int(ppo.ppo_epoch * ppo.num_mini_batch * 2 * n_steps),
                          "note": "recurrent layers: time-major chunks on the engine's second stream underneath the encoder (forward) / the "
                                  "data-gradient chain (backward), each chunk ONE persistent launch (state exchange between its workgroups "
                                  "per step instead of a launch per step); event pairs are per layer call incl. the chunk's input projection "
                                  "/ data gradient, so `ms` is the wall time beside the convolutions, not exclusive GPU time"}
        crit = [r for r in table if not r["site"].startswith("rnn")]
        if auto_probe and crit:
            a.probe = crit[0]["site"]
    tag, flops_per_frame = PROBES[a.probe]
    eng.probe_enable(tag)
    rollout_ms = time_rollouts(trainer)
    barrier()
    steps_before, local_before = trainer.num_steps_done, getattr(trainer, "local_steps_done", 0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.run_update_cycle()
    barrier()
    dt = time.perf_counter() - t0
    probe_ms, probe_cnt = eng.probe_read()
    eng.probe_enable(-1)
    phases = phase_record(rollout_ms(a.steps), dt / a.steps * 1e3, ppo)  # (this rank's; before the max over ranks)
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if os.environ.get("HAB_BENCH_CHECKSUM"):  # development: parameters must not depend on how the gradient exchange is scheduled
        pf = eng.params_flat.double()
        print(f"[rank {rank}] params checksum {pf.sum().item():.12e} {pf.abs().sum().item():.12e}", flush=True)
    trainer.shutdown()  # joins the straggler-counter poller before the store goes away
    # which exchange carried the gradients: the library-owned RCCL communicator (csrc/comm.hip; ranks as IT counts them) or the
    # torch.distributed callbacks -- so that a scaling line says what it measured
    comm_rec = None
    if dist:
        nc = getattr(trainer._agent.updater, "_native_comm", None)
        comm_rec = {"comm": "rccl-native" if nc is not None else "torch-callbacks", "backend": torch.distributed.get_backend(),
                    "rccl_ranks": nc.world_size() if nc is not None else None, "process_group_ranks": torch.distributed.get_world_size(),
                    "grad_overlap": os.environ.get("HAB_NO_GRAD_OVERLAP") is None}
    # env-steps actually collected over all ranks (the trainer's all-reduced counter): equals world * n_envs * n_steps * K unless
    # DD-PPO's preemptive straggler rule cut a rollout short (ppo_trainer.py:641-653), in which case only the collected steps count
    steps_total = trainer.num_steps_done - steps_before
    assert 0 < steps_total <= world * n_envs * n_steps * a.steps
    local_steps = trainer.local_steps_done - local_before
    c4 = None
    if world > 1 and a.workload == "c2" and not a.no_extras and not a.no_cpu_baseline:
        # BASELINE.json configs[3] (the workload north_star's >= 6x scaling target is written on): DD-PPO ResNet18 + 2-layer LSTM, 64 envs per
        # rank, run by EVERY rank after the headline cycles (same barrier + max-over-ranks timing)
        trainer.envs.close()
        trainer._agent._rollouts = None
        del eng
        torch.cuda.empty_cache()
        try:
            c4 = run_cycles("c3", 5, 2, distributed=True)
        except Exception as exc:  # noqa: BLE001 -- a failure every rank shares (e.g. memory) must not take the headline figure with it
            c4 = {"error": repr(exc)[:300]}
    exchange_ab = None
    if world > 1 and a.workload == "c2" and not a.no_extras and not a.no_cpu_baseline:
        # Both gradient exchanges in ONE run: the torch.distributed callbacks (default) and the library-owned RCCL
        # communicator (HAB_NATIVE_COMM=1; falls back by a vote of all ranks when it cannot be opened), each on a fresh trainer from the
        # same seeds -- same rollouts, same minibatches -- so the parameter arenas after the cycles must be EQUAL bit for bit on every
        # rank if the two exchanges sum the same segments in the same order.
        exchange_ab = {}
        flats = {}
        saved = os.environ.get("HAB_NATIVE_COMM")
        try:
            for name, env in (("torch-callbacks", None), ("rccl-native", "1")):
                if env is None:
                    os.environ.pop("HAB_NATIVE_COMM", None)
                else:
                    os.environ["HAB_NATIVE_COMM"] = env
                kf = {}
                try:
                    r_ = run_cycles("c2", 3, 1, distributed=True, keep_flat=kf)
                    exchange_ab[name] = {"value": r_["value"], "unit": r_["unit"], "ms_per_step": r_["ms_per_step"], "steps": 3, "warmup": 1,
                                         "exchange_that_ran": kf.get("comm")}
                    flats[name] = kf.get("params")
                except Exception as exc:  # noqa: BLE001
                    exchange_ab[name] = {"error": repr(exc)[:300]}
        finally:
            if saved is None:
                os.environ.pop("HAB_NATIVE_COMM", None)
            else:
                os.environ["HAB_NATIVE_COMM"] = saved
        same = torch.tensor([1 if (len(flats) == 2 and all(v is not None for v in flats.values())
                                   and torch.equal(flats["torch-callbacks"], flats["rccl-native"])) else 0], device="cuda")
        torch.distributed.all_reduce(same, op=torch.distributed.ReduceOp.MIN)
        exchange_ab["bit_identical"] = bool(same.item())
        # how far apart, should the two differ (RCCL may pick another algorithm / channel count for the second communicator: a different
        # summation order is rounding noise, a wrong segment is not)
        diff = torch.tensor([float("inf")], device="cuda")
        if len(flats) == 2 and all(v is not None for v in flats.values()):
            diff = (flats["torch-callbacks"].double() - flats["rccl-native"].double()).abs().max().reshape(1).float()
        torch.distributed.all_reduce(diff, op=torch.distributed.ReduceOp.MAX)
        exchange_ab["param_max_abs_diff"] = float(diff.item())
        exchange_ab["native_ran"] = exchange_ab.get("rccl-native", {}).get("exchange_that_ran") == "rccl-native"
        exchange_ab["how"] = ("two fresh DD-PPO trainers of the headline workload from the same seeds, 1 warm-up + 3 timed cycles each, barrier + "
                              "max-over-ranks timing, preemptive straggler rule off (sync_frac = 2: identical rollouts); bit_identical = parameter arenas "
                              "equal on EVERY rank (MIN over ranks)")
        del flats
    if rank != 0:
        torch.distributed.destroy_process_group()
        return
    frames = frames_seen(a.probe, local_steps, a.steps)
    if a.probe.startswith("enc_"):
        kname = f"resnet encoder {a.probe[4:]} (all kernels)"
    elif a.probe in PROBE_KERNELS:
        kname = f"{a.probe}: {PROBE_KERNELS[a.probe]}...>"
    else:
        kname = f"contraction at call site {a.probe}"
    rl = site_roofline(a.probe, flops_per_frame, frames, probe_ms) if probe_ms > 0 else {"bound": "mfma", "achieved": None, "peak": None, "frac": None}
    traffic, traffic_src = hbm_traffic(a.workload, a.probe)
    out = {
        "metric": "env-steps/sec (SPS) PointNav RGB-D 256x256, 64 envs x 128 rollout" if a.workload != "c5" else
                  "env-steps/sec (SPS) ObjectNav RGB-D+semantic 256x256, 32 envs x 64 rollout",
        "value": round(steps_total / dt, 1), "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "phases": phases,
        # ranks that took part, read from the process group after init (1 without one): a launcher that silently started fewer
        # ranks than --gpus would show here
        "ranks_seen": torch.distributed.get_world_size() if dist else 1,
        "exchange": comm_rec,
        "config": {"workload": WORKLOADS[a.workload]["name"], "envs_per_gpu": n_envs, "rollout_steps": n_steps,
                   "ppo_epoch": ppo.ppo_epoch, "num_mini_batch": ppo.num_mini_batch, "parallelism": f"dp{world}",
                   "matrix_path": "fp32 in / fp32 out; exact 3-term bf16 operand split, 6 (uint8 operand: 3) partial products on "
                                  "v_mfma_f32_32x32x16_bf16, fp32 accumulate (csrc/igemm_bf3.h); HAB_BF3=0 selects v_mfma_f32_32x32x2_f32"},
        "roofline": dict(rl, kernel=kname, traffic=traffic, traffic_source=traffic_src, launches=probe_cnt,
                         avg_launch_ms=round(probe_ms / max(probe_cnt, 1), 4), share_of_step=round(probe_ms / (dt * 1e3), 4),
                         frames_per_launch=round(frames / max(probe_cnt, 1), 1)),
    }
    if traffic is not None and a.probe in SITE_MODEL and probe_cnt:
        upd_frames = (frames / probe_cnt if a.probe in PERSISTENT_GRID_SITES else
                      n_envs * n_steps // ppo.num_mini_batch // site_chunks(a.probe))
        out["roofline"]["traffic_ratio"] = round(traffic / (SITE_MODEL[a.probe][0] * upd_frames + SITE_LAUNCH_BYTES.get(a.probe, 0)), 3)
        out["roofline"]["traffic_basis"] = (f"{'mean' if a.probe in PERSISTENT_GRID_SITES else 'update-sized'} launch of {upd_frames:.1f} frames; "
                                            f"algorithmic bytes {SITE_MODEL[a.probe][0]} per frame"
                                            + (f" + {SITE_LAUNCH_BYTES[a.probe]} per launch (weights)" if a.probe in SITE_LAUNCH_BYTES else ""))
    if a.workload in ("c2", "c3"):
        # FLOPs per env-step of the contractions that are EXECUTED: F = 2 MAC_fwd (1 + 1/T) + E (2 (3 MAC_fwd - MAC_first_dgrad)) -- the data
        # gradient of the first convolution (wrt the observation) is never computed.  SURVEY.md 8(d)'s formula counts it (C2 2.365,
        # C3 2.442 GFLOP); both are reported, priced against the fp32 MFMA peak (the round-1 yardstick) for comparability.
        mac_fwd, mac_first, f_survey = (89.3e6, 32.5e6, 2.365e9) if a.workload == "c2" else (168.82e6, 25.69e6, 2.442e9)
        f_exec = 2.0 * (mac_fwd * (1.0 + 1.0 / n_steps) + ppo.ppo_epoch * (3.0 * mac_fwd - mac_first))
        rate = steps_total / world / dt
        out["roofline"]["whole_cycle_frac"] = round(rate * f_exec / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4)
        out["roofline"]["whole_cycle"] = {"executed_gflop_per_env_step": round(f_exec / 1e9, 4), "tflops": round(rate * f_exec / 1e12, 2),
                                          "frac_of_fp32_mfma_peak": out["roofline"]["whole_cycle_frac"],
                                          "frac_with_survey_formula": round(rate * f_survey / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                                          # against the pipe in use: bf16 MFMA peak / 6 partial products (conv1: 3.75, priced at 6 here)
                                          "frac_of_split_ceiling": round(rate * f_exec / (PEAK_BF16_MFMA_TFLOPS / 6.0 * 1e12), 4),
                                          "split_ceiling_tflops": round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)}
    if table is not None:
        out["roofline"]["kernels"] = table
        out["roofline"]["site_choice"] = ("largest share of the step among the call sites of the critical stream in this run's all-probes cycle"
                                          if auto_probe else "--probe")
    if overlapped is not None:
        out["roofline"]["overlapped"] = overlapped
    if world == 1 and not a.no_cpu_baseline and a.workload == "c2":
        base, par = cpu_baseline_and_parity(trainer, cfg, cpu_threads=(os.cpu_count() or 1) if a.cpu_threads < 0 else a.cpu_threads)
        other = next((p_ for p_ in (os.path.join(ROOT, "profiles", f"{t_}_cpu_leg_threads.json") for t_ in ("r06", "r05", "r04", "r03")) if os.path.exists(p_)), "")
        if other:  # the same leg timed once at every thread setting on the GPU box's host (committed measurement)
            base["thread_settings_measured"] = json.load(open(other))
        out["cpu_baseline"] = base
        if par:
            out["parity"] = par
        trainer.envs.close()
        del trainer, eng
        torch.cuda.empty_cache()
        if not a.no_extras:
            c3_state = {}
            out["c3"] = run_cycles("c3", 10, 2, keep_state=c3_state)
            if par:
                out["c3"]["parity"], out["c3"]["cpu_baseline"] = c3_parity_record(c3_state)
            del c3_state
            c5_state = {}
            out["c5"] = run_cycles("c5", 3, 1, keep_state=c5_state)  # BASELINE.json configs[4], per GPU
            if par:
                out["c5"]["parity"], out["c5"]["cpu_baseline"] = c3_parity_record(c5_state, envs=2, steps=64, workload="c5")
            del c5_state
            out["encoder_r18_b8192"] = encoder_record_one_call()
    if world > 1:
        if c4 is not None:
            out["c4"] = c4
        if exchange_ab is not None:
            out["exchange_ab"] = exchange_ab
        out["note"] = ("n_gpus > 1: `cpu_baseline`, `parity`, the per-site `kernels` table and the c3 / c5 / encoder sub-records are reported by "
                       "the N = 1 run only (rank 0 would have to run them while the other ranks have left); `c4` = the ResNet18 + LSTM "
                       "DD-PPO workload on all ranks")
    print(json.dumps(out), flush=True)
    if dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
