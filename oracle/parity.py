"""TEST INFRASTRUCTURE ONLY -- compares ONE PPO minibatch step of the HIP engine (evaluate -> fused loss -> backward) with the CPU
oracle on the very same rollout arena, at any shape incl. BASELINE.json's (2048 SimpleCNN frames / 4096 ResNet18 frames of
256x256 RGB-D).  Used by tests/test_gpu_fullshape.py and by bench.py's `parity` leg; never by the product package.

The oracle side is `oracle.functional.minibatch_chunked` (exactly evaluate_actions + ppo_loss + backward of
rl/ppo/ppo.py:195-258, evaluated a few env columns at a time so host memory stays bounded)."""
from __future__ import annotations

import ctypes as C
import time
from typing import Dict, Optional

import numpy as np
import torch

from . import functional as O

GOAL = "pointgoal_with_gps_compass"


def spec_of(policy) -> O.NetSpec:
    """Oracle NetSpec of a habitat_amd policy (from the constructor arguments it keeps)."""
    kw = policy._engine_kwargs
    if kw["arch"] == "simple_cnn":
        return O.NetSpec(kind="baseline", rnn_type=kw["rnn_type"], num_layers=kw["rnn_layers"], hidden=kw["hidden"],
                         num_actions=kw["num_actions"])
    return O.NetSpec(kind="resnet", rnn_type=kw["rnn_type"], num_layers=kw["rnn_layers"], backbone=f"resnet{kw['backbone']}",
                     baseplanes=kw["baseplanes"], visual_keys=list(kw["visual_order"]),
                     normalize=bool(kw["normalize_visual_inputs"]), hidden=kw["hidden"], num_actions=kw["num_actions"])


def columns_to_cpu(storage, cols: torch.Tensor, T: int) -> dict:
    """CPU copy of the env columns `cols` of a device RolloutStorage (rows 0..T), shaped like RolloutStorage.buffers."""
    B = storage.buffers
    dev_cols = cols.to(storage.device)
    take = lambda v: v[0:T + 1].index_select(1, dev_cols).cpu()
    out = {k: take(B[k]) for k in ("recurrent_hidden_states", "rewards", "value_preds", "returns", "action_log_probs", "actions",
                                   "prev_actions", "masks")}
    out["observations"] = {k: take(v) for k, v in B["observations"].items()}
    return out


def rel(got, ref, floor=1e-3) -> float:
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(floor, np.abs(ref).max()))


def resnet_relu_taps(eng, backbone: str):
    """Engine views (flat, NHWC) of every post-ReLU activation of the ResNet encoder that the backward derives a mask from, in the
    order of oracle.functional.resnet_backbone's `taps["relu"]`: stem, per block its inner convs and its output, compression."""
    nblocks = [3, 4, 6, 3] if backbone == "resnet50" else [2, 2, 2, 2]
    per_block = 3 if backbone == "resnet50" else 2
    taps = [eng.tap(6)]  # HAB_TAP_STEM
    k = 0
    inplanes, exp = 32, (4 if backbone == "resnet50" else 1)
    for li, nb in enumerate(nblocks):
        planes = 32 << li
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            has_ds = bi == 0 and (stride != 1 or inplanes != planes * exp)
            for q in range(per_block):
                taps.append(eng.tap(100 + k + q))  # HAB_TAP_CONV_OUT: inner convs (GN + ReLU), last = block output (GN + residual + ReLU)
            k += per_block + (1 if has_ds else 0)
            inplanes = planes * exp
    taps.append(eng.tap(8))  # HAB_TAP_COMPRESSION
    return taps


class MaskInjector:
    """tap_hook for oracle.functional.minibatch_chunked: for every chunk of env columns, compares the sign pattern of the engine's
    saved post-ReLU activations with the oracle's, counts the bits that differ and overwrites exactly those engine activations
    with the oracle's sign (0, or a tiny positive value) -- so that a second backward uses the oracle's ReLU masks."""

    def __init__(self, policy, T: int, n: int):
        eng, kw = policy.engine, policy._engine_kwargs
        self.T, self.n, self.hidden = T, n, kw["hidden"]
        self.kind = kw["arch"]
        self.rin = eng.tap(3).view(T, n, -1)  # HAB_TAP_RNN_IN: [..., :hidden] = ReLU(visual fc)
        if self.kind == "resnet":
            self.eng_taps = resnet_relu_taps(eng, f"resnet{kw['backbone']}")
        else:
            self.eng_taps = [eng.tap(0), eng.tap(1)]  # HAB_TAP_CONV1 / CONV2 (ReLU outputs of SimpleCNN)
        self.pool_idx = eng.tap(13).view(torch.uint8) if self.kind == "resnet" else None  # HAB_TAP_POOL_IDX
        self.pool_diff = 0
        self.flips: Dict[str, list] = {}
        self.total = 0
        self.n_act = 0

    def _patch(self, name, ev, ov):
        diff = (ev > 0) != (ov > 0)
        nf = int(diff.sum())
        self.n_act += ev.numel()
        if nf:
            mag = float(torch.maximum(ev, ov)[diff].max())
            f = self.flips.setdefault(name, [0, 0.0])
            f[0] += nf
            f[1] = max(f[1], mag)
            self.total += nf
            ev.copy_(torch.where(ov > 0, torch.clamp_min(ev, 1e-20), torch.zeros_like(ev)))

    def __call__(self, c0: int, k: int, taps: dict):
        T, n = self.T, self.n
        if self.kind == "resnet":
            pairs = list(taps["relu"])
            fc = taps["visual_fc"]
        else:
            pairs = [("conv1", taps["conv1"]), ("conv2", taps["conv2"])]
            fc = taps["cnn_out"]
        assert len(pairs) == len(self.eng_taps)
        for (name, a), t in zip(pairs, self.eng_taps):
            a = a.detach()
            C_, H_, W_ = a.shape[1:]
            ov = a.view(T, k, C_, H_, W_).permute(0, 1, 3, 4, 2).to(t.device, non_blocking=True)
            ev = t.view(T, n, H_, W_, C_)[:, c0:c0 + k]
            self._patch(name, ev, ov)
        ov = fc.detach().view(T, k, -1).to(self.rin.device)
        self._patch("visual_fc", self.rin[:, c0:c0 + k, :self.hidden], ov)
        if self.pool_idx is not None:
            # the other discontinuity upstream of which weights sit (only the stem's): which element of a 3x3 window is the maximum.
            # Near-ties within round-off resolve differently; the oracle's arg-max (ATen: first maximum in scan order) is written
            # over the engine's arg-max bytes (kh * 3 + kw), the number of windows that differed is counted.
            stem = taps["stem"].detach()
            _, ind = torch.nn.functional.max_pool2d(stem, 3, 2, 1, return_indices=True)  # flat h * W + w of the input plane
            Hs, Ws = stem.shape[2:]
            Ho, Wo, C_ = ind.shape[2], ind.shape[3], ind.shape[1]
            h, w = ind // Ws, ind % Ws
            ho = torch.arange(Ho).view(1, 1, Ho, 1)
            wo = torch.arange(Wo).view(1, 1, 1, Wo)
            code = ((h - (ho * 2 - 1)) * 3 + (w - (wo * 2 - 1))).to(torch.uint8)
            code = code.view(T, k, C_, Ho, Wo).permute(0, 1, 3, 4, 2).to(self.pool_idx.device)
            ev = self.pool_idx[:T * n * Ho * Wo * C_].view(T, n, Ho, Wo, C_)[:, c0:c0 + k]
            self.pool_diff += int((ev != code).sum())
            ev.copy_(code)


def minibatch_parity(policy, ppo, storage, batch, cfg, env_chunk: int = 4, with_grads: bool = True, inject_masks: bool = False) -> Dict[str, object]:
    """Runs the engine's minibatch step on `batch` (a habitat_amd MiniBatch of `storage`) WITHOUT the optimiser step and the oracle
    on a CPU copy of the same columns.  Returns error figures (max |err| / max |ref| per tensor family, relative loss errors,
    per-parameter gradient errors elementwise and norm-wise) plus timings; the caller decides on thresholds.
    inject_masks: after the first backward the oracle's ReLU sign pattern is written over the engine's saved activations
    (MaskInjector) and the backward is repeated; the report then carries both sets of gradient errors and the number of mask bits
    that differed -- what remains after the injection is NOT attributable to ReLU's discontinuity."""
    from habitat_amd import _lib
    eng = policy.engine
    T, n = batch.T, batch.n
    Bn = T * n
    Bf = storage.buffers
    obs = Bf["observations"]
    params = {k: v.detach().cpu().clone() for k, v in policy.state_dict().items()}  # BEFORE evaluate: RunningMeanAndVar not yet updated
    spec = spec_of(policy)
    adv = batch.advantages_full
    extra = {k: obs[k] for k in ("semantic", "objectgoal", "compass", "gps", "visual_features") if k in obs}
    dev = storage.device
    v, lp, ent, dv, dlp, dent = (torch.zeros(Bn, device=dev) for _ in range(6))
    out16 = torch.zeros(16, device=dev)
    eng.evaluate(obs.get("rgb"), obs.get("depth"), obs.get(GOAL), batch.rows, Bf["recurrent_hidden_states"], Bf["masks"], Bf["actions"],
                 batch.pack, Bn, n, value=v, log_prob=lp, entropy=ent, prev_actions=Bf["prev_actions"], extra=extra)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().hab_ppo_loss(P(v), P(lp), P(ent), P(Bf["action_log_probs"]), P(adv), P(Bf["value_preds"]), P(Bf["returns"]),
                                       P(batch.rows), Bn, float(cfg.clip_param), float(cfg.value_loss_coef), float(cfg.entropy_coef),
                                       int(cfg.use_clipped_value_loss), P(dv), P(dlp), P(dent), P(out16), _lib.stream_ptr()), "hab_ppo_loss")
    def backward():
        eng.backward(obs.get("rgb"), obs.get("depth"), obs.get(GOAL), batch.rows, Bf["actions"], batch.pack, dv, dlp, dent,
                     prev_actions=Bf["prev_actions"], extra=extra)
        return {k: g.detach().cpu().clone() for k, g in eng.grad_views.items() if k not in eng.buffer_names}

    grads_first = backward() if with_grads else None
    torch.cuda.synchronize()
    injector = MaskInjector(policy, T, n) if (inject_masks and with_grads) else None
    # ---- oracle on the same columns -------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    cols = batch.inds.clone()
    buf = columns_to_cpu(storage, cols, T)
    adv_cpu = adv[0:T + 1].index_select(1, cols.to(dev)).cpu()
    trainable = [k for k, p_ in policy.named_parameters() if p_.requires_grad]
    ref = O.minibatch_chunked(params, spec, buf, adv_cpu, torch.arange(n), T, cfg, trainable, env_chunk=env_chunk, with_grads=with_grads,
                              tap_hook=injector)
    t_oracle = time.perf_counter() - t0
    rep: Dict[str, object] = {"frames": Bn, "envs": n, "steps": T, "oracle_seconds": round(t_oracle, 1)}
    rep["value_max_rel"] = rel(v.cpu().numpy(), ref["value"].view(-1).numpy())
    rep["log_prob_max_rel"] = rel(lp.cpu().numpy(), ref["log_prob"].view(-1).numpy())
    rep["entropy_max_rel"] = rel(ent.cpu().numpy(), ref["entropy"].view(-1).numpy())
    got = out16[:4].cpu().numpy().astype(np.float64)
    for i, k in enumerate(("value_loss", "action_loss", "dist_entropy", "total")):
        rep[k] = float(got[i])
        rep[k + "_ref"] = float(ref[k])
        rep[k + "_rel"] = float(abs(got[i] - ref[k]) / max(1e-6, abs(ref[k])))
    if ref["rmv"] is not None:
        pre = "net.visual_encoder.running_mean_and_var."
        sd = policy.state_dict()
        rep["rmv_max_rel"] = max(rel(sd[pre + "_" + k].cpu().numpy(), ref["rmv"][k].numpy(), floor=1e-6) for k in ("mean", "var", "count"))
    if with_grads:
        def errors(grads):
            per = {}
            for k, g in grads.items():
                if k not in ref["grads"]:
                    continue
                r = ref["grads"][k].numpy().astype(np.float64)
                gg = g.numpy().astype(np.float64)
                per[k] = (float(np.abs(gg - r).max() / max(1e-12, np.abs(r).max())),
                          float(np.linalg.norm(gg - r) / max(1e-30, np.linalg.norm(r))))
            return per

        per = errors(grads_first)
        rep["grad_max_rel_elementwise"] = max(v_[0] for v_ in per.values())
        rep["grad_max_rel_normwise"] = max(v_[1] for v_ in per.values())
        rep["grad_worst"] = sorted(((k, round(a, 7), round(b, 7)) for k, (a, b) in per.items()), key=lambda x: -x[2])[:6]
        rep["_grads_per_param"] = per
        if injector is not None:
            per2 = errors(backward())
            rep["relu_mask_bits_differing"] = injector.total
            rep["relu_mask_bits_total"] = injector.n_act
            rep["maxpool_argmax_differing"] = injector.pool_diff
            rep["relu_mask_flips_by_layer"] = {k: (v_[0], float(f"{v_[1]:.3e}")) for k, v_ in injector.flips.items()}
            rep["grad_max_rel_normwise_with_oracle_masks"] = max(v_[1] for v_ in per2.values())
            rep["grad_max_rel_elementwise_with_oracle_masks"] = max(v_[0] for v_ in per2.values())
            rep["grad_worst_with_oracle_masks"] = sorted(((k, round(a, 7), round(b, 7)) for k, (a, b) in per2.items()), key=lambda x: -x[2])[:6]
            rep["_grads_per_param_with_oracle_masks"] = per2
    return rep


def returns_parity(storage, next_value: torch.Tensor, use_gae: bool, gamma: float, tau: float) -> float:
    """max relative error of the device GAE (whatever variant the storage is configured with) against the reference loop
    (common/rollout_storage.py:174-205) on the storage's own rewards / value predictions / masks."""
    B = storage.buffers
    T = storage.current_rollout_step_idx
    r, _ = O.compute_returns(B["rewards"].cpu(), B["value_preds"].cpu(), B["masks"].cpu(), next_value.cpu().view(-1, 1), T, use_gae, gamma, tau)
    return rel(B["returns"].cpu().numpy()[:T], r.numpy()[:T])


# ---------------------------------------------------------------------------------------------------------------------------------
# Whole-update parity: teacher-forced (every minibatch step of the HIP path starts from the ORACLE's pre-step state) and free-running
# ---------------------------------------------------------------------------------------------------------------------------------
def oracle_rollout(params: dict, spec: O.NetSpec, N: int, T: int, H: int, W: int, hidden: int, hidden_layers: int, cfg, seed: int = 4242,
                   task: str = "pointnav"):
    """The oracle collects one rollout of N synthetic envs x T steps (oracle/synth.py observations, policy.act per step with a fixed
    exponential-noise stream: rl/ppo/ppo_trainer.py:343-399), bootstraps the value of step T and computes the GAE returns
    (common/rollout_storage.py:174-205).  Returns (buffers shaped like RolloutStorage.buffers, next_value, the E x M env-column
    permutations of the update, seconds spent generating observations)."""
    import time as _t
    from . import synth
    from .fixtures import synth_rollout_inputs
    t0 = _t.perf_counter()
    envs = synth.SyntheticEnvs(N, H, W, seed=seed, task=task)  # objectnav: + semantic, objectgoal, compass, gps (ddppo_objectnav.yaml)
    obs, rew, done = synth_rollout_inputs(envs, T)
    t_env = _t.perf_counter() - t0
    torch.manual_seed(7)
    noise = torch.stack([torch.empty(N, spec.num_actions).exponential_(1) for _ in range(T)])
    perms = [list(torch.randperm(N).chunk(cfg.num_mini_batch)) for _ in range(cfg.ppo_epoch)]
    buf = dict(observations={k: torch.from_numpy(np.stack([o[k] for o in obs])) for k in obs[0]})
    del obs
    buf["recurrent_hidden_states"] = torch.zeros(T + 1, N, hidden_layers, hidden)
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
                      buf["prev_actions"][t], buf["masks"][t], exp_noise=noise[t])
            buf["actions"][t], buf["action_log_probs"][t], buf["value_preds"][t] = r["actions"], r["action_log_probs"], r["values"]
            buf["recurrent_hidden_states"][t + 1], buf["prev_actions"][t + 1] = r["rnn_hidden_states"], r["actions"]
        feats, _ = O.net_forward(params, spec, {k: v[T] for k, v in buf["observations"].items()}, buf["recurrent_hidden_states"][T],
                                 buf["prev_actions"][T], buf["masks"][T])
        nv = O.heads(params, feats)[2]
    buf["returns"], buf["value_preds"] = O.compute_returns(buf["rewards"], buf["value_preds"], buf["masks"], nv, T, True, cfg.gamma, cfg.tau)
    return buf, nv, perms, t_env


def storage_from_oracle(buf: dict, next_value, T: int, N: int, observation_space, action_space, policy, device, cfg):
    """A device RolloutStorage holding the oracle's rollout `buf`; the returns are recomputed by the device GAE (scan variant, the
    one the timed cycles use) from the oracle's value predictions."""
    from habitat_amd.common.rollout_storage import RolloutStorage
    st = RolloutStorage(T, N, observation_space, action_space, policy, device=device, gae_variant="scan")
    B = st.buffers
    for k, v in buf["observations"].items():
        B["observations"][k].copy_(v)
    for k in ("actions", "prev_actions", "action_log_probs", "rewards", "masks", "recurrent_hidden_states", "value_preds"):
        B[k].copy_(buf[k])
    st.current_rollout_step_idxs = [T]
    st.compute_returns(next_value.to(device), True, cfg.gamma, cfg.tau)
    return st


def oracle_update_trace(params: dict, spec: O.NetSpec, buf: dict, T: int, cfg, trainable, perms):
    """Runs O.ppo_update (rl/ppo/ppo.py:301-332) from `params` with fresh Adam moments and keeps, for every minibatch step k, the
    state the step STARTED from (parameters + buffers, Adam moments, step count, env columns) and what it produced (loss scalars,
    gradient norm, per-frame values / log-probs).  Returns (averaged metrics, trace); trace[k]["after"] is the parameter state
    the step left (= trace[k + 1]["params"], or the final parameters for the last step)."""
    p = {k: (v.clone().requires_grad_(True) if k in trainable else v.clone()) for k, v in params.items()}
    opt = dict(step=0, m={k: torch.zeros_like(p[k]) for k in trainable}, v={k: torch.zeros_like(p[k]) for k in trainable})
    trace, record = [], []

    def pre_step(step, epoch, inds, pp, oo):
        trace.append(dict(step=int(step), epoch=int(epoch), inds=inds.clone(), params={k: v.detach().clone() for k, v in pp.items()},
                          m={k: v.clone() for k, v in oo["m"].items()}, v={k: v.clone() for k, v in oo["v"].items()}))

    metrics = O.ppo_update(p, spec, buf, T, cfg, opt, list(trainable), perms=perms, record=record, pre_step=pre_step)
    final = {k: v.detach().clone() for k, v in p.items()}
    for k, (tr, rc) in enumerate(zip(trace, record)):
        tr.update({q: float(rc[q]) for q in ("value_loss", "action_loss", "dist_entropy", "grad_norm", "total", "surrogate_abs_mean")})
        tr["values"], tr["log_probs"] = rc["values"], rc["log_probs"]
        tr["after"] = trace[k + 1]["params"] if k + 1 < len(trace) else final
    return metrics, trace, final


def _clip_decisions(values, log_probs, batch, clip):
    """Which branch of the two `min` / `where` selections of the PPO loss (rl/ppo/ppo.py:212-236) every frame takes: (the clipped
    surrogate is the active one, the value prediction is clipped)."""
    ratio = torch.exp(log_probs.view(-1, 1) - batch["action_log_probs"])
    s1 = batch["advantages"] * ratio
    s2 = batch["advantages"] * torch.clamp(ratio, 1.0 - clip, 1.0 + clip)
    return (s2 < s1).view(-1), ((values.view(-1, 1) - batch["value_preds"]).abs() >= clip).view(-1)


def _load_step_state(policy, upd, tr, lr_group=None):
    """Oracle pre-step state -> engine: parameters + buffers through load_state_dict (re-packs the kernel-side weight images),
    Adam moments scattered into the flat arenas at the engine's parameter offsets, step count."""
    policy.load_state_dict(tr["params"])
    opt = upd.optimizer
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    for nm, shp, off in policy.engine.specs:
        if nm in tr["m"]:
            n = tr["m"][nm].numel()
            opt.exp_avg[off:off + n].copy_(tr["m"][nm].reshape(-1))
            opt.exp_avg_sq[off:off + n].copy_(tr["v"][nm].reshape(-1))
    opt.step_count = tr["step"]


def summarize_free_running(fr: dict) -> dict:
    """Two figures that say at a glance why the free-running means differ from the teacher-forced steps: in how many minibatch steps a
    frame sat on a clip boundary and came out on the other side, and the largest gradient-norm difference among the steps where none did
    (a single flipped frame is worth ~1 / sqrt(frames) of a minibatch gradient)."""
    flips = [a + b for a, b in zip(fr.get("ratio_clip_flips", []), fr.get("value_clip_flips", []))]
    quiet = [g for g, f in zip(fr.get("grad_norm_rel", []), flips) if not f]
    return {"steps_with_clip_flips": sum(1 for f in flips if f), "frames_flipped": int(sum(flips)),
            "max_grad_norm_rel_in_steps_without_flips": (max(quiet) if quiet else None)}


def update_parity(policy, upd, storage, buf: dict, trace, final, T: int, cfg, trainable, bar: float = 1e-4) -> Dict[str, object]:
    """The HIP update against the oracle's trace of the SAME update (same rollout arena `storage` == `buf`, same permutations).

    teacher_forced: for every minibatch step k the engine is loaded with the oracle's pre-step parameters, buffers and Adam moments,
    runs ONE step (evaluate -> loss -> backward -> clip + Adam) on the same env columns, and is compared on: value loss, action
    loss (absolute error over the mean |clipped surrogate| -- the loss itself is a mean near zero), entropy, gradient norm (all
    relative), the parameter STEP it took (max |step_hip - step_oracle| in units of lr, and L2-relative), per-frame values /
    log-probs, and the number of frames whose ratio-clip / value-clip branch differs.  Nothing a step does can leak into the next
    comparison, so a defect that only shows after step 1 (moments, bias correction, statistics) is caught, chaos is not amplified.

    free_running: the update as the trainer runs it, from the common start; per step the same figures plus how far the parameters
    had already drifted from the oracle's before the step."""
    from habitat_amd.common.rollout_storage import MiniBatch
    from habitat_amd.rl.ppo.ppo import SLOT_WIDTH
    dev = storage.device
    N = storage._num_envs
    adv_dev = upd.get_advantages(storage)
    adv_cpu = O.get_advantages(buf["returns"], buf["value_preds"], cfg.use_normalized_advantage)
    dones = torch.logical_not(storage.buffers["masks"]).cpu().view(-1, N).numpy()
    lr = float(cfg.lr)

    def one_step(tr, slot):
        batch = MiniBatch(storage, tr["inds"], T, adv_dev, dones)
        upd._update_from_batch(batch, tr["epoch"], storage, slot)
        Bn = T * len(tr["inds"])
        return upd._wk["v"][:Bn].clone(), upd._wk["lp"][:Bn].clone()

    def compare(tr, slot, v, lp, before, after_ref):
        h = slot.cpu().double()
        take = lambda t: t[0:T, tr["inds"]].flatten(0, 1)  # (O.gather_minibatch without the observations: ~1 GB per step at C2's shape)
        ob = {"action_log_probs": take(buf["action_log_probs"]), "value_preds": take(buf["value_preds"]), "advantages": take(adv_cpu)}
        v, lp = v.cpu(), lp.cpu()
        r = {"value_loss_rel": abs(h[0].item() - tr["value_loss"]) / max(1e-6, abs(tr["value_loss"])),
             "action_loss_err_over_surrogate": abs(h[1].item() - tr["action_loss"]) / max(1e-6, tr["surrogate_abs_mean"]),
             "dist_entropy_rel": abs(h[2].item() - tr["dist_entropy"]) / max(1e-6, abs(tr["dist_entropy"])),
             "grad_norm_rel": abs(h[12].item() - tr["grad_norm"]) / max(1e-6, abs(tr["grad_norm"])),
             "value_max_rel": rel(v.numpy(), tr["values"].numpy()), "log_prob_max_rel": rel(lp.numpy(), tr["log_probs"].numpy())}
        d_hip = _clip_decisions(v, lp, ob, cfg.clip_param)
        d_ref = _clip_decisions(tr["values"], tr["log_probs"], ob, cfg.clip_param)
        r["ratio_clip_flips"] = int((d_hip[0] != d_ref[0]).sum())
        r["value_clip_flips"] = int((d_hip[1] != d_ref[1]).sum())
        if before is not None:  # the parameter step of this minibatch against the oracle's, from the same starting point
            sd = policy.state_dict()
            worst, num, den = 0.0, 0.0, 0.0
            for k in trainable:
                s_hip = sd[k].detach().cpu().double() - before[k].double()
                s_ref = after_ref[k].double() - before[k].double()
                worst = max(worst, float((s_hip - s_ref).abs().max()))
                num += float((s_hip - s_ref).pow(2).sum())
                den += float(s_ref.pow(2).sum())
            r["param_step_max_err_over_lr"] = worst / lr
            r["param_step_rel_l2"] = (num / max(den, 1e-300)) ** 0.5
        return r

    def fold(rows):
        keys = [k for k in rows[0]]
        out = {k: [float(f"{r[k]:.3e}") if isinstance(r[k], float) else r[k] for r in rows] for k in keys}
        return out

    policy.train()
    # ---- teacher-forced ------------------------------------------------------------------------------------------------------------
    rows = []
    slots = torch.zeros(len(trace), SLOT_WIDTH, device=dev)
    for k, tr in enumerate(trace):
        _load_step_state(policy, upd, tr)
        v, lp = one_step(tr, slots[k])
        rows.append(compare(tr, slots[k], v, lp, tr["params"], tr["after"]))
    tf = fold(rows)
    loss_keys = ("value_loss_rel", "action_loss_err_over_surrogate", "dist_entropy_rel")
    tf["max_rel_losses"] = max(max(tf[k]) for k in loss_keys)
    tf["max_rel_grad_norm"] = max(tf["grad_norm_rel"])
    tf["max_rel"] = max(tf["max_rel_losses"], tf["max_rel_grad_norm"])
    tf["bar"] = bar
    tf["within_bar"] = bool(tf["max_rel"] <= bar)
    # ---- free-running ---------------------------------------------------------------------------------------------------------------
    _load_step_state(policy, upd, trace[0])
    rows = []
    slots = torch.zeros(len(trace), SLOT_WIDTH, device=dev)
    for k, tr in enumerate(trace):
        sd = policy.state_dict()
        drift = max(float((sd[q].detach().cpu() - tr["params"][q]).abs().max()) for q in trainable)
        v, lp = one_step(tr, slots[k])
        r = compare(tr, slots[k], v, lp, None, None)
        r["param_max_abs_drift_before_step"] = drift
        rows.append(r)
    fr = fold(rows)
    fr.update(summarize_free_running(fr))
    sd = policy.state_dict()
    fr["post_update_param_max_abs_diff"] = float(f"{max(float((sd[q].detach().cpu() - final[q]).abs().max()) for q in trainable):.3e}")
    host = slots.cpu().double()
    for i, k in ((0, "value_loss"), (1, "action_loss"), (2, "dist_entropy"), (12, "grad_norm")):
        ref = sum(tr[k] for tr in trace) / len(trace)
        fr[k + "_rel_of_update_means"] = float(f"{abs(host[:, i].mean().item() - ref) / max(1e-6, abs(ref)):.3e}")
    return {"steps": len(trace), "teacher_forced": tf, "free_running": fr}
