"""GPU parity AT THE BENCHMARK SHAPES (BASELINE.json configs[1] / configs[2] / configs[4] per GPU): the trainer built from the YAML entrypoint collects a
full 64 envs x 128 steps rollout of 256x256 RGB-D on the device, then ONE full-size PPO minibatch -- 2048 frames for SimpleCNN+GRU
(M=4), 4096 frames for ResNet18 + 2-layer LSTM (M=2), 1024 frames (32 envs x 64 steps, M=2) of rgb + depth + semantic for the ObjectNav
ResNet50, hidden 512 -- is evaluated by the HIP engine (forward, fused loss, backward)
and by the CPU oracle on the same arena columns (oracle.parity / oracle.functional.minibatch_chunked, bounded memory).  These are the
code paths the golden fixtures (T <= 32, N <= 4) never reach: tile selection and split-K plans at M = 1.8e6 ... 1.7e7 rows,
per-tile buffer re-basing of > 2 GB tensors, the patch-resident weight gradients at 4096 frames, chunked GroupNorm at full batch.

Tolerance: 1e-4 relative on values / log-probs / entropy (all frames), on the three losses and on the returns (north_star);
gradients: 1e-4 norm-wise for every parameter of the shallow SimpleCNN policy and for every parameter downstream of the 20-layer
GroupNorm encoder.  The encoder's own weight gradients sit upstream of ~1e9 ReLU decisions per minibatch, a handful of which fall
within fp32 round-off of zero and legitimately come out on the other side (two evaluations of the CPU oracle itself differ by
2e-3 when ONE such bit flips: tests/test_oracle_golden.py::test_minibatch_chunked_equals_whole).  That is accounted for, not
assumed: the mask bits on which engine and oracle disagree are counted (they must all belong to activations below 1e-4), the
oracle's sign pattern (and its max-pool arg-max, the one other discontinuity: near-tied window maxima) is written over the engine's
saved activations, the backward is repeated and EVERY gradient must then meet 1e-4 norm-wise.  The measured figures are written to gpurun_out/parity_<workload>.json."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(name, rep):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"parity_{name}.json"), "w") as f:
        json.dump({k: v for k, v in rep.items() if not k.startswith("_")}, f, indent=1)


@pytest.mark.parametrize("workload,frames", [("c2", 2048), ("c3", 4096), ("c5", 1024)])
def test_full_minibatch_at_benchmark_shape_vs_oracle(workload, frames):
    import bench
    from oracle import parity
    torch.manual_seed(1234)
    trainer, cfg = bench.make_trainer(workload, 3)
    trainer._init_train()
    ppo_cfg = cfg.habitat_baselines.rl.ppo
    agent = trainer._agent
    st = agent.rollouts
    N, T = (32, 64) if workload == "c5" else (64, 128)  # C5 = BASELINE.json configs[4] per GPU: ObjectNav ResNet50, 5 visual channels
    assert (trainer.envs.num_envs, ppo_cfg.num_steps, ppo_cfg.hidden_size) == (N, T, 512)
    agent.eval()
    assert trainer.collect_rollout() == N * T
    last = st.get_last_step()
    nv = agent.actor_critic.get_value({k: v.contiguous() for k, v in last["observations"].items()}, last["recurrent_hidden_states"],
                                      last["prev_actions"], last["masks"])
    st.compute_returns(nv, ppo_cfg.use_gae, ppo_cfg.gamma, ppo_cfg.tau)  # the GAE variant the benchmark times (scan)
    rep = {"workload": bench.WORKLOADS[workload]["name"], "gae_variant": st.gae_variant,
           "returns_max_rel": parity.returns_parity(st, nv, ppo_cfg.use_gae, ppo_cfg.gamma, ppo_cfg.tau)}
    agent.train()
    ppo = agent.updater
    adv = ppo.get_advantages(st)
    torch.manual_seed(99)
    batch = next(st.data_generator(adv, ppo_cfg.num_mini_batch))
    assert batch.T * batch.n == frames
    rep.update(parity.minibatch_parity(agent.actor_critic, ppo, st, batch, ppo_cfg, env_chunk=4, inject_masks=True))
    per = rep.pop("_grads_per_param")
    per2 = rep.pop("_grads_per_param_with_oracle_masks")
    _report(workload, rep)
    print(json.dumps(rep))
    assert rep["returns_max_rel"] <= 1e-4, rep
    for k in ("value_max_rel", "log_prob_max_rel", "entropy_max_rel", "value_loss_rel", "action_loss_rel", "dist_entropy_rel"):
        assert rep[k] <= 1e-4, (k, rep[k])
    if "rmv_max_rel" in rep:
        assert rep["rmv_max_rel"] <= 1e-5, rep["rmv_max_rel"]
    bad = []
    for k, (elem, normw) in per.items():
        # upstream of a ReLU / max-pool decision: the visual encoder and visual_fc (C3: ~1.5e9 sign bits, C2: ~3e8).  Which
        # near-zero activations flip depends on the arithmetic path (fp32 MFMA / split-bf16 MFMA / the CPU's own blocking), a
        # handful always do; their effect is accounted for below, where the oracle's decisions are injected
        deep = ("visual_encoder" in k) or ("visual_fc" in k)
        if normw > (3e-2 if deep else 1e-4):
            bad.append((k, elem, normw))
    assert not bad, bad
    # ReLU accounting: only activations within round-off of zero may disagree in sign, and with the oracle's signs in place ...
    assert all(mag < 1e-4 for _, mag in rep["relu_mask_flips_by_layer"].values()), rep["relu_mask_flips_by_layer"]
    assert rep["relu_mask_bits_differing"] <= 1e-5 * rep["relu_mask_bits_total"]
    bad2 = [(k, e, nw) for k, (e, nw) in per2.items() if nw > 1e-4]
    assert not bad2, bad2  # ... every gradient agrees
    trainer.envs.close()


def test_benchmark_shape_minibatch_agrees_on_every_matrix_path():
    """C2 at the benchmark shape (one 2048-frame minibatch of 256 x 256 RGB-D) through the engine under every kernel-selection mask of
    hab_set_matrix_path that production can end up on: fp32 MFMA everywhere (0), the plain split-bf16 contraction alone (1), the default
    minus the kernels hard-wired to this geometry (what other observation sizes run).  The default path is pinned to the oracle by
    test_full_minibatch_at_benchmark_shape_vs_oracle[c2-2048]; every other path must reproduce ITS outputs to 1e-4 and its gradients
    norm-wise (1e-4 behind the encoder, 3e-2 inside it: a few ReLU decisions within round-off of zero legitimately differ)."""
    import bench
    from habitat_amd import _lib
    L = _lib.lib()
    torch.manual_seed(1234)
    trainer, cfg = bench.make_trainer("c2", 3)
    trainer._init_train()
    ppo_cfg = cfg.habitat_baselines.rl.ppo
    agent = trainer._agent
    st = agent.rollouts
    agent.eval()
    assert trainer.collect_rollout() == 64 * 128
    last = st.get_last_step()
    nv = agent.actor_critic.get_value({k: v.contiguous() for k, v in last["observations"].items()}, last["recurrent_hidden_states"],
                                      last["prev_actions"], last["masks"])
    st.compute_returns(nv, ppo_cfg.use_gae, ppo_cfg.gamma, ppo_cfg.tau)
    agent.train()
    adv = agent.updater.get_advantages(st)
    torch.manual_seed(99)
    batch = next(st.data_generator(adv, ppo_cfg.num_mini_batch))
    eng = agent.actor_critic.engine
    Bf, obs = st.buffers, st.buffers["observations"]
    Bn = batch.T * batch.n
    g = torch.Generator(device="cuda").manual_seed(5)
    dv, dlp, dent = (torch.randn(Bn, device="cuda", generator=g) * 1e-3 for _ in range(3))

    def run():
        v, lp, ent = (torch.zeros(Bn, device="cuda") for _ in range(3))
        eng.evaluate(obs.get("rgb"), obs.get("depth"), obs.get("pointgoal_with_gps_compass"), batch.rows, Bf["recurrent_hidden_states"],
                     Bf["masks"], Bf["actions"], batch.pack, Bn, batch.n, value=v, log_prob=lp, entropy=ent, prev_actions=Bf["prev_actions"])
        eng.backward(obs.get("rgb"), obs.get("depth"), obs.get("pointgoal_with_gps_compass"), batch.rows, Bf["actions"], batch.pack, dv, dlp,
                     dent, prev_actions=Bf["prev_actions"])
        torch.cuda.synchronize()
        return (v.clone(), lp.clone(), ent.clone()), {k: t.detach().clone() for k, t in eng.grad_views.items() if k not in eng.buffer_names}

    prev = L.hab_set_matrix_path(-1)
    try:
        ref_out, ref_g = run()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
        for name, mode in (("fp32_mfma", 0), ("split_bf16_igemm_only", 1), ("no_256x256_strip_kernels", 1023 & ~(64 | 256 | 512)), ("no_dense_gemm", 1023)):
            L.hab_set_matrix_path(mode)
            out, grads = run()
            for a, b, what in zip(out, ref_out, ("value", "log_prob", "entropy")):
                assert rel(a, b) <= 1e-4, (name, what, rel(a, b))
            bad = []
            for k, gr in grads.items():
                nw = float((gr - ref_g[k]).norm() / ref_g[k].norm().clamp_min(1e-20))
                deep = ("visual_encoder" in k)
                if nw > (3e-2 if deep else 1e-4):
                    bad.append((k, nw))
            assert not bad, (name, bad)
    finally:
        L.hab_set_matrix_path(prev)
        trainer.envs.close()
