#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the DD-PPO training hot path on synthetic PointNav RGB-D 256x256.

  python bench.py [--gpus N] [--steps K] [--warmup W]                       (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                             (N > 1, one rank per GPU, RCCL)

A "step" is ONE full cycle of the hot path on one batch of synthetic input: rollout collection of
64 envs x 128 steps (policy forward per step) -> GAE -> PPO update (ppo_epoch x num_mini_batch
forward+backward+clip+Adam) [-> gradient all-reduce per minibatch when N > 1].  Workload at N=1 is
BASELINE.json configs[1]: PointNav SimpleCNN+GRU, 64 envs x 128 steps, 256x256 RGB-D, hyper-parameters of
config/pointnav/ppo_pointnav_habitat_iccv19.yaml (E=4, M=4).  `value` = N * 64 * 128 * K / max-over-ranks wall
time.  The JSON line also carries `roofline` (dominant conv kernel, HIP-event timed inside the timed region)
and `cpu_baseline` (the CPU oracle restatement of the reference path on a bounded sample, rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL between ranks needs it)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
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

# probe tag -> (description, flops per frame)   SimpleCNN @256^2 RGB-D (a4)
PROBES = {
    "conv1_fwd": (0, 2.0 * 63 * 63 * 32 * 256), "conv2_fwd": (1, 2.0 * 30 * 30 * 64 * 512), "conv3_fwd": (2, 2.0 * 28 * 28 * 32 * 576),
    "fc_fwd": (3, 2.0 * 25088 * 512), "conv1_wgrad": (4, 2.0 * 63 * 63 * 32 * 256), "conv2_wgrad": (5, 2.0 * 30 * 30 * 64 * 512),
    "conv3_wgrad": (6, 2.0 * 28 * 28 * 32 * 576), "conv2_dgrad": (7, 2.0 * 30 * 30 * 64 * 512), "conv3_dgrad": (8, 2.0 * 28 * 28 * 32 * 576),
    "fc_wgrad": (9, 2.0 * 25088 * 512), "fc_dgrad": (10, 2.0 * 25088 * 512),
    # ResNet18 encoder @256^2 RGB-D (a5): 168.82 MMAC forward; backward = dgrad + wgrad of every conv except the
    # stem's dgrad (25.69 MMAC)
    "enc_fwd": (11, 2.0 * 168.82e6), "enc_bwd": (12, 2.0 * (2 * 168.82e6 - 25.69e6)),
}


def make_trainer(workload: str, total_updates: int):
    from habitat_amd.config.default import get_config
    import habitat_amd.rl.ppo.ppo_trainer as tr
    w = WORKLOADS[workload]
    cfg = get_config(w["yaml"], [f"habitat_baselines.num_environments={w.get('envs', NUM_ENVS)}",
                                 f"habitat_baselines.rl.ppo.num_steps={w.get('steps', NUM_STEPS)}",
                                 f"habitat_baselines.num_updates={total_updates}", "habitat_baselines.total_num_steps=-1",
                                 "habitat_baselines.num_checkpoints=-1", f"habitat_baselines.checkpoint_interval={10 ** 9}",
                                 "habitat_baselines.rl.ddppo.distrib_backend=" + os.environ.get("HAB_BENCH_DISTRIB_BACKEND", "NCCL"),
                                 "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000",
                                 "habitat_baselines.checkpoint_folder=/tmp/habitat_amd_bench_ckpt",
                                 f"habitat.simulator.sensors.rgb.height={OBS}", f"habitat.simulator.sensors.rgb.width={OBS}",
                                 f"habitat.simulator.sensors.depth.height={OBS}", f"habitat.simulator.sensors.depth.width={OBS}"] + w["overrides"]
                     + [o for o in os.environ.get("HAB_BENCH_OVERRIDES", "").split(",") if o])  # development hook
    trainer = tr.PPOTrainer(cfg)
    return trainer, cfg


def cpu_baseline(sample_envs=64, sample_steps=32):
    """The oracle (CPU restatement of the reference path, pinned to the reference by tests/golden) timed on the host
    cores on a bounded sample of the same workload: same obs size, same E=4 x M=4 update, fewer envs x steps."""
    import types
    import numpy as np
    from oracle import functional as O
    from oracle import synth
    from oracle.fixtures import baseline_param_shapes, det_params, synth_rollout_inputs
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # more threads than this slow the small-batch CPU convs down
    N, T, hidden = sample_envs, sample_steps, 512
    params = det_params(baseline_param_shapes(4, OBS, OBS, hidden), 1)
    spec = O.NetSpec(kind="baseline", hidden=hidden)
    cfg = types.SimpleNamespace(clip_param=0.1, ppo_epoch=4, num_mini_batch=4, value_loss_coef=0.5, entropy_coef=0.01, lr=2.5e-4,
                                eps=1e-5, max_grad_norm=0.5, use_normalized_advantage=True, use_clipped_value_loss=True,
                                gamma=0.99, tau=0.95)
    t0 = time.perf_counter()
    envs = synth.SyntheticEnvs(N, OBS, OBS, seed=100)
    obs, rew, done = synth_rollout_inputs(envs, T)
    t_env = time.perf_counter() - t0
    t0 = time.perf_counter()
    buf = dict(observations={k: torch.from_numpy(np.stack([o[k] for o in obs])) for k in obs[0]})
    buf["recurrent_hidden_states"] = torch.zeros(T + 1, N, 1, hidden)
    buf["rewards"] = torch.zeros(T + 1, N, 1)
    buf["rewards"][:T] = torch.from_numpy(rew).unsqueeze(-1)
    buf["masks"] = torch.zeros(T + 1, N, 1, dtype=torch.bool)
    buf["masks"][1:] = torch.from_numpy(~done).unsqueeze(-1)
    for k in ("value_preds", "action_log_probs"):
        buf[k] = torch.zeros(T + 1, N, 1)
    buf["actions"] = torch.zeros(T + 1, N, 1, dtype=torch.long)
    buf["prev_actions"] = torch.zeros(T + 1, N, 1, dtype=torch.long)
    with torch.no_grad():
        for t in range(T):
            r = O.act(params, spec, {k: v[t] for k, v in buf["observations"].items()}, buf["recurrent_hidden_states"][t],
                      buf["prev_actions"][t], buf["masks"][t])
            buf["actions"][t], buf["action_log_probs"][t], buf["value_preds"][t] = r["actions"], r["action_log_probs"], r["values"]
            buf["recurrent_hidden_states"][t + 1], buf["prev_actions"][t + 1] = r["rnn_hidden_states"], r["actions"]
        feats, _ = O.net_forward(params, spec, {k: v[T] for k, v in buf["observations"].items()}, buf["recurrent_hidden_states"][T],
                                 buf["prev_actions"][T], buf["masks"][T])
        nv = O.heads(params, feats)[2]
    buf["returns"], buf["value_preds"] = O.compute_returns(buf["rewards"], buf["value_preds"], buf["masks"], nv, T, True, 0.99, 0.95)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = dict(step=0, m={k: torch.zeros_like(v) for k, v in p.items()}, v={k: torch.zeros_like(v) for k, v in p.items()})
    O.ppo_update(p, spec, buf, T, cfg, opt, list(p.keys()))
    dt = time.perf_counter() - t0
    return {"value": round(N * T / dt, 2), "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{N} envs x {T} steps of the same workload (256x256 RGB-D, SimpleCNN+GRU, E=4 x M=4), oracle/functional.py on "
                      f"torch-CPU fp32, {dt:.1f} s (synthetic obs generation {t_env:.1f} s excluded)"}


# kernel(s) launched by a probed call site, for the HBM-traffic lookup (conv2 dgrad = one merged-stride-class launch)
PROBE_KERNELS = {"conv2_dgrad": ("igemm_dma_kernel<ConvDgradMergedProb, 2, 2, 2, 2, false>", 1), "conv1_fwd": ("igemm_kernel<ObsConvFwdProb, 2, 1, 4, 1>", 1),
                 "conv1_wgrad": ("igemm_kernel<ObsConvWgradProb, 2, 1, 4, 1>", 1)}


def hbm_traffic(workload, probe):
    """HBM bytes per probed call, from the committed rocprofv3 --pmc passes (profiles/r01_c2_hbm_traffic.json, produced by
    tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE collected in separate runs of this script, FETCH_SIZE doubled per the gfx950
    note of MI355X_MICROARCH.md).  Hardware counters cannot be read from inside the timed run, so the figure is the profiled one;
    null when the probed call site has no entry."""
    path = os.path.join(ROOT, "profiles", "r01_c2_hbm_traffic.json")
    if workload != "c2" or probe not in PROBE_KERNELS or not os.path.exists(path):
        return None
    name, launches = PROBE_KERNELS[probe]
    for k, r in json.load(open(path)).items():
        if name in k:
            return round(launches * (r["fetch_bytes_per_call"] + r["write_bytes_per_call"]))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--probe", default=None, choices=list(PROBES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    if a.probe is None:
        a.probe = "conv2_dgrad" if a.workload == "c2" else ("enc_fwd" if a.workload == "c3_frozen" else "enc_bwd")
    if a.workload == "c5" and a.probe.startswith("enc_"):
        # ResNet50 on 5 channels: 375.0 MMAC forward (a5); the stem's data gradient (7x7x8 pad -> 5 real ch) is not computed
        PROBES["enc_fwd"], PROBES["enc_bwd"] = (11, 2.0 * 375.0e6), (12, 2.0 * (2 * 375.0e6 - 32.1e6))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        if a.gpus > 1:
            raise SystemExit(f"--gpus {a.gpus} needs `python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py ...` (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU execution path")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    trainer, cfg = make_trainer(a.workload, a.warmup + a.steps + 1)
    trainer._init_train()
    eng = trainer._agent.actor_critic.engine
    dist = torch.distributed.is_initialized()

    def barrier():
        if dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.run_update_cycle()
    tag, flops_per_frame = PROBES[a.probe]
    eng.probe_enable(tag)
    barrier()
    steps_before, local_before = trainer.num_steps_done, getattr(trainer, "local_steps_done", 0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.run_update_cycle()
    barrier()
    dt = time.perf_counter() - t0
    probe_ms, probe_cnt = eng.probe_read()
    eng.probe_enable(-1)
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if os.environ.get("HAB_BENCH_CHECKSUM"):  # development: parameters must not depend on how the gradient exchange is scheduled
        pf = eng.params_flat.double()
        print(f"[rank {rank}] params checksum {pf.sum().item():.12e} {pf.abs().sum().item():.12e}", flush=True)
    if rank != 0:
        torch.distributed.destroy_process_group()
        return
    ppo = cfg.habitat_baselines.rl.ppo
    n_envs, n_steps = WORKLOADS[a.workload].get("envs", NUM_ENVS), WORKLOADS[a.workload].get("steps", NUM_STEPS)
    # env-steps actually collected over all ranks (the trainer's all-reduced counter): equals world * n_envs * n_steps * K unless
    # DD-PPO's preemptive straggler rule cut a rollout short (ppo_trainer.py:641-653), in which case only the collected steps count
    steps_total = trainer.num_steps_done - steps_before
    assert 0 < steps_total <= world * n_envs * n_steps * a.steps
    # frames seen by the probed call site during the timed region
    local_steps = trainer.local_steps_done - local_before
    upd_frames = local_steps * ppo.ppo_epoch
    roll_frames = local_steps + n_envs * a.steps
    frames = upd_frames + (roll_frames if a.probe.endswith("_fwd") else 0)
    if a.probe.startswith("enc_"):
        kname = f"resnet encoder {a.probe[4:]} (all kernels)"
    elif a.workload == "c2" and a.probe in PROBE_KERNELS:
        kname = f"{a.probe}: {PROBE_KERNELS[a.probe][0]}"
    else:
        kname = f"igemm contraction at call site {a.probe}"
    ach = flops_per_frame * frames / (probe_ms * 1e-3) / 1e12 if probe_ms > 0 else None
    traffic = hbm_traffic(a.workload, a.probe)
    out = {
        "metric": "env-steps/sec (SPS) PointNav RGB-D 256x256, 64 envs x 128 rollout" if a.workload != "c5" else
                  "env-steps/sec (SPS) ObjectNav RGB-D+semantic 256x256, 32 envs x 64 rollout",
        "value": round(steps_total / dt, 1), "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOADS[a.workload]["name"], "envs_per_gpu": n_envs, "rollout_steps": n_steps,
                   "ppo_epoch": ppo.ppo_epoch, "num_mini_batch": ppo.num_mini_batch, "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2) if ach else None,
                     "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4) if ach else None,
                     "traffic": traffic, "launches": probe_cnt, "avg_launch_ms": round(probe_ms / max(probe_cnt, 1), 4)},
    }
    if world == 1 and not a.no_cpu_baseline and a.workload == "c2":
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
