"""Bisect of tests/test_zz_gpu_cpca.py::test_cpca_through_the_engine_hook_vs_oracle (GPU; diagnostic, not a test).

Splits the engine-vs-oracle gradient difference into
  (A) forward: the engine's aux_loss_state tensors vs the oracle's,
  (B) the cpca module on the GPU vs on the CPU for IDENTICAL inputs,
  (C) the sensitivity of the module's two input gradients to (A) (CPU module on the engine's tensors vs on the oracle's),
  (D) the engine's backward with the ORACLE's two input gradients injected through hab_policy_set_extra_grads (a linear probe loss).
Usage (GPU box): python tools/diag_cpca.py > gpurun_out/diag_cpca.txt
"""
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "habitat-lab_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import functional as O  # noqa: E402
import test_zz_gpu_cpca as Z  # noqa: E402


def rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64)
    b = b.detach().cpu().numpy().astype(np.float64)
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))


class Probe(nn.Module):
    """loss = <rnn_output, Gf> + <perception_embed, Gp>: its input gradients are exactly (Gf, Gp)."""

    def __init__(self, gf, gp):
        super().__init__()
        self.gf, self.gp = gf, gp

    def forward(self, state, batch):
        return {"loss": (state["rnn_output"] * self.gf).sum() + (state["perception_embed"] * self.gp).sum()}


def run(kind, T=16, N=4, hidden=64):
    from habitat_amd.common import spaces as S
    from habitat_amd.common.rollout_storage import RolloutStorage
    from habitat_amd.rl.ppo import PPO, PointNavBaselinePolicy, PointNavResNetPolicy
    Z._register()
    AUX, GOAL = Z.AUX, Z.GOAL
    H = W = 64 if kind == "resnet18" else 44
    osp = S.Dict({"rgb": S.Box(0, 255, (H, W, 3), np.uint8), "depth": S.Box(0.0, 1.0, (H, W, 1), np.float32),
                  GOAL: S.Box(-1e9, 1e9, (2,), np.float32)})
    asp = S.Discrete(4)
    torch.manual_seed(17)
    aux_cfg = {AUX: Z.AUX_CFG}
    if kind == "baseline":
        pol = PointNavBaselinePolicy(osp, asp, hidden_size=hidden, aux_loss_config=aux_cfg, max_frames=T * N, max_envs=N)
        spec = O.NetSpec(kind="baseline", hidden=hidden)
    else:
        pol = PointNavResNetPolicy(osp, asp, hidden_size=hidden, backbone="resnet18", aux_loss_config=aux_cfg, max_frames=T * N, max_envs=N)
        spec = O.NetSpec(kind="resnet", rnn_type="GRU", num_layers=1, backbone="resnet18", baseplanes=32, visual_keys=("rgb", "depth"),
                         normalize=False, hidden=hidden)
    params = {k: v.detach().clone() for k, v in pol.state_dict().items() if not k.startswith("aux_loss_modules.")}
    host_module = copy.deepcopy(pol.aux_loss_modules[AUX])
    pol.to("cuda")
    pol.train()
    st = RolloutStorage(T, N, osp, asp, pol, device="cuda", gae_variant="scan")
    B = st.buffers
    Z._fill(B, np.random.default_rng(9), T, N, H, W, hidden)
    st.current_rollout_step_idxs = [T]
    cfg = types.SimpleNamespace(clip_param=0.2, ppo_epoch=1, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.01, lr=2.5e-4, eps=1e-5,
                                max_grad_norm=0.5, use_clipped_value_loss=True, use_normalized_advantage=False)
    ppo = PPO.from_config(pol, cfg)
    adv = ppo.get_advantages(st)
    torch.manual_seed(5)
    batch = next(st.data_generator(adv, 1))
    inds = batch.inds
    take = lambda t: t[0:T, inds].flatten(0, 1).cpu()
    obs = {k: take(v) for k, v in B["observations"].items()}
    ob = {"action_log_probs": take(B["action_log_probs"]), "advantages": take(adv), "value_preds": take(B["value_preds"]), "returns": take(B["returns"])}
    seq = batch["rnn_build_seq_info"]
    info = {k[4:]: seq[k] for k in seq.keys() if k.startswith("cpu_")}
    info["cpu_sequence_lengths"] = info["sequence_lengths"]
    actions_cpu = take(B["actions"])

    # ---- oracle ----
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    taps = {}
    v, lp, ent, _ = O.evaluate_actions(p, spec, obs, B["recurrent_hidden_states"][0, inds].cpu(), take(B["prev_actions"]), take(B["masks"]),
                                       actions_cpu, training=True, taps=taps)
    total_o, *_ = O.ppo_loss(v, lp, ent, ob, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)
    pe_o = taps["cnn_out"] if kind == "baseline" else taps["visual_fc"]
    rn_o = taps["rnn_out"]
    torch.manual_seed(23)
    aux_o = host_module({"rnn_output": rn_o, "perception_embed": pe_o}, {"action": actions_cpu, "rnn_build_seq_info": info})["loss"]
    gf_o, gp_o = torch.autograd.grad(aux_o, [rn_o, pe_o], retain_graph=True)
    # gradients of the PPO part alone and of the aux part alone wrt the parameters
    keys = list(p.keys())
    g_ppo = torch.autograd.grad(total_o, [p[k] for k in keys], retain_graph=True, allow_unused=True)
    g_aux = torch.autograd.grad(aux_o, [p[k] for k in keys], retain_graph=True, allow_unused=True)
    g_ppo = {k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(keys, g_ppo)}
    g_aux = {k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(keys, g_aux)}
    g_tot = {k: g_ppo[k] + g_aux[k] for k in keys}
    print(f"[{kind}] oracle: ppo loss {float(total_o):.6f} aux loss {float(aux_o):.6f} |gf| {float(gf_o.norm()):.4e} |gp| {float(gp_o.norm()):.4e}")

    # ---- engine forward through the bridge with the real module ----
    eng = pol.engine
    torch.manual_seed(23)
    ve, lpe, ente, _, aux = pol.evaluate_actions(batch["observations"], batch["recurrent_hidden_states"], batch["prev_actions"], batch["masks"],
                                                 batch["actions"], batch["rnn_build_seq_info"])
    Bf = T * N
    feats_e = eng.tap(4)[:Bf * hidden].view(Bf, hidden).clone()
    perc_e = eng.tap(3).view(Bf, -1)[:, :hidden].clone()
    print(f"[{kind}] (A) forward: rnn_output rel {rel(feats_e, rn_o):.3e}  perception_embed rel {rel(perc_e, pe_o):.3e}  "
          f"values rel {rel(ve, v):.3e}  aux loss rel {abs(float(aux[AUX]['loss']) - float(aux_o)) / abs(float(aux_o)):.3e}")

    # ---- (B) module GPU vs CPU on identical (oracle) inputs ----
    gm = pol.aux_loss_modules[AUX]
    a = rn_o.detach().cuda().requires_grad_(True)
    b = pe_o.detach().cuda().requires_grad_(True)
    torch.manual_seed(23)
    lg = gm({"rnn_output": a, "perception_embed": b}, {"action": batch["actions"], "rnn_build_seq_info": batch["rnn_build_seq_info"]})["loss"]
    ga, gb = torch.autograd.grad(lg, [a, b])
    print(f"[{kind}] (B) module GPU vs CPU, same inputs: loss rel {abs(float(lg) - float(aux_o)) / abs(float(aux_o)):.3e}  "
          f"d rnn_output rel {rel(ga, gf_o):.3e}  d perception_embed rel {rel(gb, gp_o):.3e}")

    # ---- (C) CPU module on the engine's tensors vs on the oracle's ----
    a2 = feats_e.cpu().requires_grad_(True)
    b2 = perc_e.cpu().requires_grad_(True)
    torch.manual_seed(23)
    l2 = host_module({"rnn_output": a2, "perception_embed": b2}, {"action": actions_cpu, "rnn_build_seq_info": info})["loss"]
    ga2, gb2 = torch.autograd.grad(l2, [a2, b2])
    print(f"[{kind}] (C) CPU module, engine's tensors vs oracle's: d rnn_output rel {rel(ga2, gf_o):.3e}  d perception_embed rel {rel(gb2, gp_o):.3e}")

    # ---- the failing comparison itself, split by parameter ----
    bt = {k: batch[k] for k in ("action_log_probs", "advantages", "value_preds", "returns")}
    total, *_ = O.ppo_loss(ve, lpe, ente, bt, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)
    for q in pol.parameters():
        q.grad = None
    (total + aux[AUX]["loss"]).backward()
    shallow = [k for k in eng.grad_views if k not in eng.buffer_names and "visual_encoder" not in k and "visual_fc" not in k]
    real = {k: eng.grad_views[k].clone() for k in eng.grad_views if k not in eng.buffer_names}
    print(f"[{kind}] real module through the hook (the failing test), shallow parameters:")
    for k in shallow:
        print(f"    {k:48s} rel {rel(real[k], g_tot[k]):.3e}   |ppo part| {float(g_ppo[k].norm()):.3e} |aux part| {float(g_aux[k].norm()):.3e}")

    # ---- (D) engine backward with the ORACLE's input gradients injected ----
    pol.aux_loss_modules[AUX] = Probe(gf_o.cuda(), gp_o.cuda())
    ve, lpe, ente, _, aux = pol.evaluate_actions(batch["observations"], batch["recurrent_hidden_states"], batch["prev_actions"], batch["masks"],
                                                 batch["actions"], batch["rnn_build_seq_info"])
    total, *_ = O.ppo_loss(ve, lpe, ente, bt, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)
    for q in pol.parameters():
        q.grad = None
    (total + aux[AUX]["loss"]).backward()
    print(f"[{kind}] (D) oracle's (d rnn_output, d perception_embed) injected through set_extra_grads:")
    worst = 0.0
    for k in shallow:
        e = rel(eng.grad_views[k], g_tot[k])
        worst = max(worst, e)
        print(f"    {k:48s} rel {e:.3e}")
    print(f"[{kind}] (D) worst shallow {worst:.3e}")
    # ---- (E) PPO loss alone (probe with zero gradients) ----
    pol.aux_loss_modules[AUX] = Probe(torch.zeros_like(gf_o).cuda(), torch.zeros_like(gp_o).cuda())
    ve, lpe, ente, _, aux = pol.evaluate_actions(batch["observations"], batch["recurrent_hidden_states"], batch["prev_actions"], batch["masks"],
                                                 batch["actions"], batch["rnn_build_seq_info"])
    total, *_ = O.ppo_loss(ve, lpe, ente, bt, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)
    (total + aux[AUX]["loss"]).backward()
    print(f"[{kind}] (E) PPO loss alone: worst shallow {max(rel(eng.grad_views[k], g_ppo[k]) for k in shallow):.3e}")


if __name__ == "__main__":
    for kind in (sys.argv[1:] or ["baseline", "resnet18"]):
        run(kind)
