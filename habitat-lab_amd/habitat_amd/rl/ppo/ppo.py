"""PPO updater with the surface of habitat_baselines/rl/ppo/ppo.py:33-384 (`PPO.from_config`,
`update(rollouts) -> Dict[str, float]`, `.optimizer`, hooks, `get_resume_state/load_state_dict`) whose
minibatch step is a fixed sequence of HIP launches on the policy engine:

    evaluate (fwd, gather through rows) -> hab_ppo_loss (loss + dL/d{value,logp,entropy} + metrics)
    -> engine backward (all parameter grads into the flat arena) -> [DD-PPO: all-reduce of the arena]
    -> hab_clip_adam_step (global-norm clip + Adam on the flat arena)

No host synchronisation happens inside `update`; the learner metrics stay on the device and are
read back once at the end (the reference reads ~15 scalars per minibatch)."""
from __future__ import annotations

import collections
from typing import Any, Dict, List, Optional

import torch
from torch import nn as nn

from habitat_amd import _lib
from habitat_amd._lib import check, ptr, stream_ptr
from habitat_amd.common.baseline_registry import baseline_registry
from habitat_amd.common.rollout_storage import MiniBatch, RolloutStorage

EPS_PPO = 1e-5
METRIC_KEYS = ["value_loss", "action_loss", "dist_entropy", "_total", "value_pred_min", "value_pred_mean", "value_pred_max",
               "prob_ratio_min", "prob_ratio_mean", "prob_ratio_max", "ppo_fraction_clipped", "_B"]
# VER learner statistics (rl/ppo/ppo.py:262-263,285-299), slots 13..19 of a minibatch's metric row
VER_METRIC_KEYS = {13: "ver_is_coeffs_min", 14: "ver_is_coeffs_mean", 15: "ver_is_coeffs_max", 16: "fraction_stale",
                   17: "policy_version_difference_min", 18: "policy_version_difference_mean", 19: "policy_version_difference_max"}
SLOT_WIDTH = 24


class Updater:
    """Interface of rl/ppo/updater.py:11-39."""

    def update(self, rollouts) -> Dict[str, float]: raise NotImplementedError
    @classmethod
    def from_config(cls, actor_critic, config): raise NotImplementedError
    @property
    def lr_scheduler(self): return None
    def after_update(self) -> None: pass
    def get_resume_state(self) -> Dict[str, Any]: raise NotImplementedError
    def load_state_dict(self, state) -> None: raise NotImplementedError


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam-compatible shell (param_groups / lr for LambdaLR, state_dict for resume) around the fused
    clip+Adam kernel that updates the policy's flat parameter arena in one pass (ppo.py:112-137,347-371)."""

    def __init__(self, policy, lr, eps, betas=(0.9, 0.999)):
        self.policy = policy
        eng = policy.engine
        own = {nm for nm, _, _ in eng.specs}  # parameters of foreign modules hung on the policy are not in the arena (never trained:
        # in the reference they receive no gradient, test_ddppo_reduce.py:60-61)
        super().__init__([p for nm, p in policy.named_parameters() if p.requires_grad and nm in own], dict(lr=lr, eps=eps, betas=betas))
        self.exp_avg = torch.zeros_like(eng.params_flat)
        self.exp_avg_sq = torch.zeros_like(eng.params_flat)
        self.step_count = 0
        self.extra: List[Any] = []  # non-policy scalars trained by the same optimiser (adaptive entropy coefficient), after the policy
        self._scratch = torch.zeros(1024, dtype=torch.float64, device=eng.params_flat.device)

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: float = 0.0, grad_scale: float = 1.0, grad_norm_out=None):
        eng = self.policy.engine
        g = self.param_groups[0]
        self.step_count += 1
        check(_lib.lib().hab_clip_adam_step(ptr(eng.params_flat), ptr(eng.grads_flat), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                            eng.params_flat.numel(), ptr(self._scratch), 1024, float(grad_scale),
                                            float(max_grad_norm or 0.0), float(g["lr"]), float(g["betas"][0]),
                                            float(g["betas"][1]), float(g["eps"]), self.step_count, ptr(grad_norm_out),
                                            stream_ptr()), "hab_clip_adam_step")
        eng.repack()

    def zero_grad(self, set_to_none: bool = False):
        self.policy.engine.grads_flat.zero_()

    # ---- resume-state wire format = torch.optim.Adam.state_dict() (rl/ppo/ppo.py:377-384) ---------------------------------
    def _slots(self):
        """(index, name, offset, numel, shape) of every optimised parameter in `self.param_groups[0]["params"]` order, which is
        the reference's `filter(requires_grad, self.parameters())` order (ppo.py:113)."""
        eng = self.policy.engine
        spec = {nm: (off, shp) for nm, shp, off in eng.specs}
        out = []
        for nm, par in self.policy.named_parameters():
            if not par.requires_grad or nm not in spec:
                continue
            off, shp = spec[nm]
            n = 1
            for d in shp:
                n *= int(d)
            out.append((len(out), nm, off, n, tuple(shp)))
        return out

    def state_dict(self):
        sd = flat_to_adam_state_dict(self._slots(), self.step_count, self.exp_avg, self.exp_avg_sq, self.param_groups[0])
        for x in self.extra:  # the reference's optimiser lists them after the policy's parameters (PPO.parameters() order)
            i = len(sd["param_groups"][0]["params"])
            sd["param_groups"][0]["params"].append(i)
            if self.step_count > 0:
                sd["state"][i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": x.exp_avg.detach().cpu().clone(),
                                  "exp_avg_sq": x.exp_avg_sq.detach().cpu().clone()}
        return sd

    def load_state_dict(self, sd):
        if "state" not in sd:  # round-1 files of this package: {step, exp_avg, exp_avg_sq} flat arenas
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            groups = sd.get("param_groups", [])
        else:
            n_pol = len(self._slots())
            ids = sd["param_groups"][0]["params"]
            for x, pid in zip(self.extra, ids[n_pol:]):
                e = sd["state"].get(pid)
                if e is not None:
                    x.exp_avg.copy_(e["exp_avg"])
                    x.exp_avg_sq.copy_(e["exp_avg_sq"])
            pol_sd = {"state": {k: v for k, v in sd["state"].items() if k in ids[:n_pol]},
                      "param_groups": [dict(sd["param_groups"][0], params=ids[:n_pol])]}
            self.step_count = adam_state_dict_to_flat(self._slots(), pol_sd, self.exp_avg, self.exp_avg_sq)
            groups = sd["param_groups"]
        for g, s in zip(self.param_groups, groups):
            g.update({k: v for k, v in s.items() if k != "params"})


_ADAM_GROUP_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=True, capturable=False, differentiable=False,
                            fused=None, decoupled_weight_decay=False)


def flat_to_adam_state_dict(slots, step: int, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, group: Dict[str, Any]) -> Dict[str, Any]:
    """The flat Adam arenas as `torch.optim.Adam(...).state_dict()`: per-parameter `step` (float scalar tensor, as Adam keeps
    it), `exp_avg`, `exp_avg_sq` in parameter shape, one param group listing indices 0..P-1.  A reference `PPO.load_state_dict`
    (ppo.py:382-384) accepts it unchanged."""
    state = {}
    if step > 0:  # Adam creates a parameter's state on its first step
        m, v = exp_avg.detach().cpu(), exp_avg_sq.detach().cpu()
        for i, _nm, off, n, shp in slots:
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": m[off:off + n].view(shp).clone(),
                        "exp_avg_sq": v[off:off + n].view(shp).clone()}
    g = {k: v for k, v in group.items() if k != "params"}
    for k, v in _ADAM_GROUP_DEFAULTS.items():
        g.setdefault(k, v)
    g["betas"] = tuple(g["betas"])
    g["params"] = [i for i, *_ in slots]
    return {"state": state, "param_groups": [g]}


def adam_state_dict_to_flat(slots, sd: Dict[str, Any], exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor) -> int:
    """Inverse of flat_to_adam_state_dict: scatters a torch.optim.Adam state_dict (ours or the reference's) into the flat
    arenas and returns the common step count.  Every parameter must have made the same number of steps (true for PPO: all
    parameters receive a gradient in every minibatch)."""
    st = sd["state"]
    ids = sd["param_groups"][0]["params"]
    if len(ids) != len(slots):
        raise _lib.HabError(f"optimizer state lists {len(ids)} parameters, the policy has {len(slots)}")
    exp_avg.zero_()
    exp_avg_sq.zero_()
    steps = set()
    for (i, nm, off, n, shp), pid in zip(slots, ids):
        e = st.get(pid, st.get(str(pid)))
        if e is None:
            continue
        if tuple(e["exp_avg"].shape) != shp:
            raise _lib.HabError(f"optimizer state of {nm}: shape {tuple(e['exp_avg'].shape)} != {shp}")
        exp_avg[off:off + n].copy_(e["exp_avg"].reshape(-1))
        exp_avg_sq[off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
        steps.add(int(float(e["step"])))
    if len(steps) > 1:
        raise _lib.HabError(f"optimizer state with per-parameter step counts {sorted(steps)} cannot be mapped to the fused Adam step")
    return steps.pop() if steps else 0


class LagrangeInequalityCoefficient(nn.Module):
    """Learnable coefficient alpha = exp(log_alpha) of the constraint `x > threshold` (utils/common.py:749-806, the `greater_than`
    form the adaptive entropy penalty uses): loss alpha * (threshold - [x]) - [alpha] * x, projected into [alpha_min, alpha_max]
    after every optimiser step.  log_alpha is a device scalar; its loss term, gradient and Adam step run in the fused kernels
    (`hab_ppo_loss_ver`, `hab_lagrange_adam_step`), so no value crosses to the host during an update."""

    def __init__(self, threshold: float, init_alpha: float = 1.0, alpha_min: float = 1e-4, alpha_max: float = 1.0,
                 greater_than: bool = False, device=None):
        super().__init__()
        import math
        if not greater_than:
            raise _lib.HabError("only the greater_than form (entropy > target) is on the accelerated path")
        self.log_alpha = nn.Parameter(torch.full((), math.log(init_alpha), device=device))
        self.threshold = float(threshold)
        self.log_alpha_min, self.log_alpha_max = math.log(alpha_min), math.log(alpha_max)
        self._greater_than = greater_than
        self.exp_avg = torch.zeros((), device=device)
        self.exp_avg_sq = torch.zeros((), device=device)

    def forward(self):
        return torch.exp(self.log_alpha)

    def project_into_bounds(self):
        with torch.no_grad():
            self.log_alpha.data.clamp_(self.log_alpha_min, self.log_alpha_max)

    def lagrangian_loss(self, x):
        alpha = self()
        return alpha * (self.threshold - x.detach()) - alpha.detach() * x

    @torch.no_grad()
    def adam_step(self, grad: torch.Tensor, grad_scale: float, group: Dict[str, Any], step: int, alpha_out=None) -> None:
        check(_lib.lib().hab_lagrange_adam_step(ptr(self.log_alpha), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(grad), float(grad_scale),
                                                float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                                                float(group["eps"]), int(step), float(self.log_alpha_min), float(self.log_alpha_max),
                                                ptr(alpha_out), stream_ptr()), "hab_lagrange_adam_step")


@baseline_registry.register_updater
class PPO(nn.Module, Updater):
    @classmethod
    def from_config(cls, actor_critic, config):
        return cls(actor_critic=actor_critic, clip_param=config.clip_param, ppo_epoch=config.ppo_epoch,
                   num_mini_batch=config.num_mini_batch, value_loss_coef=config.value_loss_coef,
                   entropy_coef=config.entropy_coef, lr=config.lr, eps=config.eps, max_grad_norm=config.max_grad_norm,
                   use_clipped_value_loss=config.use_clipped_value_loss,
                   use_normalized_advantage=config.use_normalized_advantage,
                   entropy_target_factor=getattr(config, "entropy_target_factor", 0.0),
                   use_adaptive_entropy_pen=getattr(config, "use_adaptive_entropy_pen", False))

    def __init__(self, actor_critic, clip_param: float, ppo_epoch: int, num_mini_batch: int, value_loss_coef: float,
                 entropy_coef: float, lr: Optional[float] = None, eps: Optional[float] = None,
                 max_grad_norm: Optional[float] = None, use_clipped_value_loss: bool = False,
                 use_normalized_advantage: bool = True, entropy_target_factor: float = 0.0,
                 use_adaptive_entropy_pen: bool = False) -> None:
        super().__init__()
        self.actor_critic = actor_critic
        self.clip_param = clip_param
        self.ppo_epoch = ppo_epoch
        self.num_mini_batch = num_mini_batch
        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.use_normalized_advantage = use_normalized_advantage
        self.device = next(actor_critic.parameters()).device
        if actor_critic.engine is None:
            raise _lib.HabError("move the policy to a GPU before building the updater (policy.to('cuda'))")
        self.optimizer = FlatAdam(actor_critic, lr, eps)
        self.non_ac_params: List[torch.Tensor] = []
        if (use_adaptive_entropy_pen and hasattr(actor_critic, "num_actions")
                and getattr(actor_critic, "action_distribution_type", None) == "gaussian"):  # ppo.py:85-103
            self.entropy_coef = LagrangeInequalityCoefficient(-float(entropy_target_factor) * actor_critic.num_actions,
                                                              init_alpha=entropy_coef, alpha_max=1.0, alpha_min=1e-4, greater_than=True,
                                                              device=self.device)
            self.non_ac_params = [self.entropy_coef.log_alpha]
            self.optimizer.extra.append(self.entropy_coef)
        dev = self.device
        self._stats = torch.zeros(4, device=dev)
        self._adv: Optional[torch.Tensor] = None
        self.last_minibatch_metrics: List[torch.Tensor] = []

    # ---- distributed hooks (overridden by DDPPO) ------------------------------------------------
    def _world_size(self) -> int:
        return 1

    def _all_reduce_grads(self) -> None:
        pass

    def _all_reduce_scalar_stats(self, t: torch.Tensor) -> None:
        pass

    # ---- advantages (ppo.py:139-153, ddppo.py:59-84) -----------------------------------------------
    def get_advantages(self, rollouts: RolloutStorage) -> torch.Tensor:
        L = _lib.lib()
        B = rollouts.buffers
        ret, vp = B["returns"], B["value_preds"]
        if self._adv is None or self._adv.shape != ret.shape:
            self._adv = torch.empty_like(ret)
        adv, cnt, s = self._adv, ret.numel(), stream_ptr()
        if not self.use_normalized_advantage:
            check(L.hab_advantages(ptr(ret), ptr(vp), ptr(adv), cnt, 0, None, None, s), "hab_advantages")
        elif self._world_size() == 1:
            check(L.hab_advantages(ptr(ret), ptr(vp), ptr(adv), cnt, 1, None, ptr(self._stats), s), "hab_advantages")
        else:
            w = float(self._world_size())
            check(L.hab_advantages(ptr(ret), ptr(vp), ptr(adv), cnt, 2, None, ptr(self._stats), s), "hab_advantages")
            self._all_reduce_scalar_stats(self._stats[0:1])
            self._stats[0:1].div_(w)
            check(L.hab_advantages(ptr(ret), ptr(vp), ptr(adv), cnt, 4, ptr(self._stats), ptr(self._stats), s), "hab_advantages")
            self._all_reduce_scalar_stats(self._stats[1:2])
            self._stats[1:2].div_(w)
            check(L.hab_advantages(ptr(ret), ptr(vp), ptr(adv), cnt, 3, ptr(self._stats), None, s), "hab_advantages")
        return adv

    # ---- one minibatch with auxiliary losses: the reference's own sequence (ppo.py:164-299) on the autograd bridge --------------
    def _aux_optimizer(self):
        """Adam over the auxiliary-loss modules' parameters (the reference optimises them with the policy in ONE Adam, ppo.py:112-137:
        same lr / eps / betas, lr following the policy optimiser's schedule; each loss's parameter group is clipped to max_grad_norm by
        ITSELF, not as part of the policy's norm, ppo.py:361-364)."""
        if getattr(self, "_aux_opt", None) is None:
            params = [p for ps in self.actor_critic.aux_loss_parameters().values() for p in ps if p.requires_grad]
            g = self.optimizer.param_groups[0]
            self._aux_opt = torch.optim.Adam(params, lr=g["lr"], eps=g["eps"], betas=tuple(g["betas"])) if params else False
        return self._aux_opt or None

    def _update_from_batch_with_aux_losses(self, batch: MiniBatch, epoch: int, rollouts: RolloutStorage, slot: torch.Tensor):
        """Policies with `aux_loss_modules`: evaluate_actions on the dense minibatch (autograd bridge: the engine's forward, its
        activations handed to the auxiliary modules as `aux_loss_state`), the PPO loss in torch ops exactly as rl/ppo/ppo.py:195-250
        states it, + the auxiliary losses (:248), backward through the bridge (the engine's backward receives the gradients wrt
        rnn_output / perception_embed through hab_policy_set_extra_grads), then the fused clip + Adam on the policy arena and a torch Adam
        on the auxiliary modules.  The dense minibatch is a gathered copy (what the reference's data generator makes)."""
        ac = self.actor_critic
        if "policy_version" in batch.storage.buffers or isinstance(self.entropy_coef, LagrangeInequalityCoefficient):
            raise _lib.HabError("auxiliary losses are supported by the PPO / DD-PPO updater with a fixed entropy coefficient (not VER, not the "
                                "adaptive entropy penalty)")
        for p in ac.parameters():
            p.grad = None  # (the arena's gradient views are written by the engine's backward; autograd's own copies are not used)
        values, logp, ent, _, aux = self._evaluate_actions(batch["observations"], batch["recurrent_hidden_states"], batch["prev_actions"],
                                                           batch["masks"], batch["actions"], batch["rnn_build_seq_info"])
        old_logp, adv, old_v, ret = batch["action_log_probs"], batch["advantages"], batch["value_preds"], batch["returns"]
        ratio = torch.exp(logp - old_logp)
        action_loss = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param))
        v = values.float()
        if self.use_clipped_value_loss:
            delta = v.detach() - old_v
            v = torch.where(delta.abs() < self.clip_param, v, old_v + delta.clamp(-self.clip_param, self.clip_param))
        value_loss = 0.5 * (v - ret).pow(2)
        action_loss, value_loss, entropy = action_loss.mean(), value_loss.mean(), ent.mean()
        terms = [self.value_loss_coef * value_loss, action_loss, -float(self.entropy_coef) * entropy] + [r["loss"] for r in aux.values()]
        total = self.before_backward(torch.stack(terms).sum())
        total.backward()
        self.after_backward(total)
        aux_opt = self._aux_optimizer()
        if aux_opt is not None and self._world_size() > 1:  # what DistributedDataParallel does for every parameter of the wrapped module
            for grp in aux_opt.param_groups:
                for p in grp["params"]:
                    if p.grad is not None:
                        self._all_reduce_scalar_stats(p.grad)
                        p.grad.div_(self._world_size())
        with torch.no_grad():
            r = ratio.detach()
            vals = values.detach().float()
            slot[0:4] = torch.stack([value_loss.detach(), action_loss.detach(), entropy.detach(), total.detach()])
            slot[4:7] = torch.stack([vals.min(), vals.mean(), vals.max()])
            slot[7:10] = torch.stack([r.min(), r.mean(), r.max()])
            slot[10] = (r > 1.0 + self.clip_param).float().mean() + (r < 1.0 - self.clip_param).float().mean()
            slot[11] = float(r.numel())
        if getattr(ac, "_dense_grad_sync", None) is None:
            self.before_step()
            scale = 1.0 / self._world_size()
        else:
            scale = 1.0  # DD-PPO: the bridge's backward already AVERAGED the arena over the ranks (DistributedDataParallel's semantics)
        self.optimizer.step(max_grad_norm=self.max_grad_norm, grad_scale=scale, grad_norm_out=slot[12:13])
        if aux_opt is not None:
            for grp in aux_opt.param_groups:
                grp["lr"] = self.optimizer.param_groups[0]["lr"]
            for ps in ac.aux_loss_parameters().values():  # ppo.py:361-364: one clip per auxiliary loss, after the rank average
                ps = [p for p in ps if p.grad is not None]
                if ps:
                    torch.nn.utils.clip_grad_norm_(ps, self.max_grad_norm)
            aux_opt.step()
            aux_opt.zero_grad(set_to_none=True)
        self.after_step()
        self.last_aux_losses = {k: r_["loss"].detach() for k, r_ in aux.items()}  # (device scalars: read them after the update)
        for name, res in aux.items():  # learner metrics `aux_<name>_<key>` (ppo.py:281-283): averaged over the update's minibatches
            for k_, v_ in res.items():
                self.__dict__.setdefault("_aux_metrics", {}).setdefault(f"aux_{name}_{k_}", []).append(v_.detach().float().mean())

    # ---- one minibatch (ppo.py:164-299) -------------------------------------------------------------
    def _update_from_batch(self, batch: MiniBatch, epoch: int, rollouts: RolloutStorage, slot: torch.Tensor):
        if len(getattr(self.actor_critic, "aux_loss_modules", ())) > 0:
            return self._update_from_batch_with_aux_losses(batch, epoch, rollouts, slot)
        eng = self.actor_critic.engine
        L = _lib.lib()
        st: RolloutStorage = batch.storage
        Bf = st.buffers
        obs = Bf["observations"]
        rgb, depth, goal, extra = self.actor_critic._obs_ptrs(obs)
        ver = "policy_version" in Bf  # VERRolloutStorage: slots of a linear buffer, importance weights, staleness statistics
        Bn, n = (batch.B, batch.n) if ver and hasattr(batch, "B") else (batch.T * batch.n, batch.n)
        w = self._work(Bn)
        eng.evaluate(rgb, depth, goal, batch.rows, Bf["recurrent_hidden_states"], Bf["masks"], Bf["actions"], batch.pack, Bn, n,
                     value=w["v"], log_prob=w["lp"], entropy=w["ent"], prev_actions=Bf["prev_actions"], extra=extra)
        lag = self.entropy_coef if isinstance(self.entropy_coef, LagrangeInequalityCoefficient) else None
        if ver or lag is not None:
            check(L.hab_ppo_loss_ver(ptr(w["v"]), ptr(w["lp"]), ptr(w["ent"]), ptr(Bf["action_log_probs"]), ptr(batch.advantages_full),
                                     ptr(Bf["value_preds"]), ptr(Bf["returns"]), ptr(batch.rows), Bn, float(self.clip_param),
                                     float(self.value_loss_coef), 0.0 if lag is not None else float(self.entropy_coef),
                                     int(self.use_clipped_value_loss), ptr(Bf.get("is_coeffs")), ptr(Bf.get("is_stale")),
                                     ptr(Bf.get("policy_version")), int(st.cpu_current_policy_version[0, 0]) if ver else 0,
                                     ptr(lag.log_alpha) if lag is not None else None, lag.threshold if lag is not None else 0.0,
                                     ptr(w["dv"]), ptr(w["dlp"]), ptr(w["dent"]), ptr(slot), stream_ptr()), "hab_ppo_loss_ver")
        else:
            check(L.hab_ppo_loss(ptr(w["v"]), ptr(w["lp"]), ptr(w["ent"]), ptr(Bf["action_log_probs"]), ptr(batch.advantages_full),
                                 ptr(Bf["value_preds"]), ptr(Bf["returns"]), ptr(batch.rows), Bn, float(self.clip_param),
                                 float(self.value_loss_coef), float(self.entropy_coef), int(self.use_clipped_value_loss),
                                 ptr(w["dv"]), ptr(w["dlp"]), ptr(w["dent"]), ptr(slot), stream_ptr()), "hab_ppo_loss")
        eng.backward(rgb, depth, goal, batch.rows, Bf["actions"], batch.pack, w["dv"], w["dlp"], w["dent"],
                     prev_actions=Bf["prev_actions"], extra=extra)
        self.before_step()
        self.optimizer.step(max_grad_norm=self.max_grad_norm, grad_scale=1.0 / self._world_size(), grad_norm_out=slot[12:13])
        if lag is not None:  # same optimiser, not part of the clipped norm (ppo.py:361-364 clips policy_parameters() only)
            if self._world_size() > 1:
                self._all_reduce_scalar_stats(slot[20:21])
            lag.adam_step(slot[20:21], 1.0 / self._world_size(), self.optimizer.param_groups[0], self.optimizer.step_count,
                          alpha_out=slot[21:22])
        self.after_step()

    def _work(self, B):
        if getattr(self, "_wk", None) is None or self._wk["v"].numel() < B:
            dev = self.device
            self._wk = {k: torch.empty(B, device=dev) for k in ("v", "lp", "ent", "dv", "dlp", "dent")}
        return self._wk

    def update(self, rollouts: RolloutStorage) -> Dict[str, float]:
        advantages = self.get_advantages(rollouts)
        self._aux_metrics: Dict[str, List[torch.Tensor]] = {}
        nmb = self.ppo_epoch * self.num_mini_batch
        slots = torch.zeros(nmb + 1, SLOT_WIDTH, device=self.device)
        k = 0
        last_epoch_slots = []
        for epoch in range(self.ppo_epoch):
            for batch in rollouts.data_generator(advantages, self.num_mini_batch):
                if k >= slots.shape[0]:
                    slots = torch.cat([slots, torch.zeros_like(slots)], 0)
                self._update_from_batch(batch, epoch, rollouts, slots[k])
                if epoch == self.ppo_epoch - 1:
                    last_epoch_slots.append(k)
                k += 1
        host = slots[:k].cpu()  # the single device->host read of the update
        self.last_minibatch_metrics = host
        out: Dict[str, float] = {}
        for i, name in enumerate(METRIC_KEYS):
            if name.startswith("_"):
                continue
            rows = last_epoch_slots if name == "ppo_fraction_clipped" else list(range(k))
            out[name] = float(host[rows, i].mean())
        out["grad_norm"] = float(host[:, 12].mean())
        if isinstance(self.entropy_coef, LagrangeInequalityCoefficient):
            out["entropy_coef"] = float(host[:, 21].mean())
        if "policy_version" in rollouts.buffers:
            for i, name in VER_METRIC_KEYS.items():
                if name.startswith("ver_is_coeffs") and "is_coeffs" not in rollouts.buffers:
                    continue
                out[name] = float(host[:, i].mean())
        for name, vals in self._aux_metrics.items():
            out[name] = float(torch.stack(vals).mean())
        return out

    def _evaluate_actions(self, *args, **kwargs):
        """Reference-style entry (ppo.py:155-162): dense tensors in, autograd out; under DD-PPO the gradients that reach
        `param.grad` are already averaged over ranks, as with the reference's DistributedDataParallel wrapper (ddppo.py:142-157)."""
        return self.actor_critic.evaluate_actions(*args, **kwargs)

    # ---- hooks kept for subclass compatibility (ppo.py:341-375) ---------------------------------------
    def before_backward(self, loss): return loss
    def after_backward(self, loss): pass

    def before_step(self):
        self._all_reduce_grads()

    def after_step(self): pass

    def get_resume_state(self):
        out = {"optim_state": self.optimizer.state_dict()}
        aux = getattr(self, "_aux_opt", None)
        if aux:  # (the reference keeps policy and auxiliary-loss parameters in ONE optimiser; here the modules' Adam is a second entry)
            out["aux_optim_state"] = aux.state_dict()
        return out

    def load_state_dict(self, state):
        if "optim_state" in state:
            self.optimizer.load_state_dict(state["optim_state"])
        if "aux_optim_state" in state and self._aux_optimizer() is not None:
            self._aux_opt.load_state_dict(state["aux_optim_state"])
