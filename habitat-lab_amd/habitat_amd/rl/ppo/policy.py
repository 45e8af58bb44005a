"""Actor-critic policies with the plugin surface of habitat_baselines/rl/ppo/policy.py
(PolicyActionData :47-96, Policy :99-249, NetPolicy :252-413, PointNavBaselinePolicy :427-460) whose
forward AND backward run in the HIP policy engine (habitat_amd/engine.py -> libhabitat_amd.so).

The module is an ``nn.Module`` whose parameters carry the reference's ``state_dict()`` names; once the
policy is moved to a GPU (``.to(device)``) every parameter is a view into ONE flat fp32 arena (and its
``.grad`` a view into the flat gradient arena) that the kernels and the fused Adam step operate on.
Initial values are drawn exactly like the reference does (same torch initialisers in the same order),
so a seed gives the same initial policy in both code bases.
"""
from __future__ import annotations

import abc
from dataclasses import dataclass
from typing import Any, Dict, Iterable, List, Optional

import numpy as np
import torch
from torch import nn as nn

from habitat_amd import _lib
from habitat_amd.common.baseline_registry import baseline_registry
from habitat_amd.common.spaces import get_num_actions
from habitat_amd.engine import DevicePackInfo, PolicyEngine

VISUAL_FEATURES_KEY = "visual_features"  # PointNavResNetNet.PRETRAINED_VISUAL_FEATURES_KEY (resnet_policy.py:399)
GOAL_UUID = "pointgoal_with_gps_compass"  # IntegratedPointGoalGPSAndCompassSensor.cls_uuid (tasks/nav/nav.py:309)
POINTGOAL_UUID = "pointgoal"            # PointGoalSensor.cls_uuid (tasks/nav/nav.py:127)


@dataclass
class PolicyActionData:
    """Same fields as rl/ppo/policy.py:47-96."""
    rnn_hidden_states: Optional[torch.Tensor] = None
    actions: Optional[torch.Tensor] = None
    values: Optional[torch.Tensor] = None
    action_log_probs: Optional[torch.Tensor] = None
    take_actions: Optional[torch.Tensor] = None
    policy_info: Optional[List[Dict[str, Any]]] = None
    should_inserts: Optional[torch.BoolTensor] = None

    def write_action(self, write_idx: int, write_action: torch.Tensor) -> None:
        self.actions[:, write_idx] = write_action

    @property
    def env_actions(self) -> torch.Tensor:
        return self.actions if self.take_actions is None else self.take_actions


class Policy(abc.ABC):
    """rl/ppo/policy.py:99-249."""

    def __init__(self, action_space):
        self._action_space = action_space

    @property
    def should_load_agent_state(self): return True
    @property
    def policy_action_space(self): return self._action_space
    @property
    def policy_action_space_shape_lens(self): return [self._action_space]
    @property
    def num_recurrent_layers(self) -> int: return 0
    @property
    def recurrent_hidden_size(self) -> int: return 0
    @property
    def visual_encoder(self): return None

    def _get_policy_components(self) -> List[nn.Module]: return []
    def aux_loss_parameters(self) -> Dict[str, Iterable[torch.Tensor]]: return {}

    def policy_parameters(self) -> Iterable[torch.Tensor]:
        for c in self._get_policy_components():
            yield from c.parameters()

    def all_policy_tensors(self) -> Iterable[torch.Tensor]:
        yield from self.policy_parameters()
        for c in self._get_policy_components():
            yield from c.buffers()

    def get_value(self, observations, rnn_hidden_states, prev_actions, masks) -> torch.Tensor:
        raise NotImplementedError

    def get_extra(self, action_data: PolicyActionData, infos, dones):
        return [] if action_data.policy_info is None else action_data.policy_info

    def evaluate_actions(self, observations, rnn_hidden_states, prev_actions, masks, action, rnn_build_seq_info):
        raise NotImplementedError

    @abc.abstractmethod
    def act(self, observations, rnn_hidden_states, prev_actions, masks, deterministic=False) -> PolicyActionData:
        raise NotImplementedError

    def on_envs_pause(self, envs_to_pause: List[int]) -> None:
        pass

    def update_hidden_state(self, rnn_hxs, prev_actions, action_data: PolicyActionData) -> None:
        for env_i, should_insert in enumerate(action_data.should_inserts):
            if should_insert.item():
                rnn_hxs[env_i] = action_data.rnn_hidden_states[env_i]
                prev_actions[env_i].copy_(action_data.actions[env_i])

    @classmethod
    @abc.abstractmethod
    def from_config(cls, config, observation_space, action_space, **kwargs):
        pass


def _attach(root: nn.Module, dotted: str, value, buffer: bool = False):
    """Registers a parameter (or buffer) under `dotted` ('a.b.0.weight'), creating bare container modules on the way."""
    *path, leaf = dotted.split(".")
    m = root
    for p in path:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if buffer:
        m.register_buffer(leaf, value)
    else:
        m.register_parameter(leaf, value)


class _EvaluateFn(torch.autograd.Function):
    """Autograd bridge for reference-style callers (loss built with torch ops, loss.backward()).
    The fused updater (habitat_amd PPO) bypasses this and calls the engine directly."""

    @staticmethod
    def forward(ctx, policy, call, *params):
        ctx.policy, ctx.call = policy, call
        v, lp, ent = policy._evaluate_dense(call)
        # `aux_loss_state` of the reference's nets (policy.py:565-589, resnet_policy.py:634-767): the recurrent encoder's output and the
        # visual embedding (ReLU(visual fc); absent for a blind net) -- copies of the engine's activations, differentiable through backward()
        eng, B, H = policy.engine, call["B"], policy.recurrent_hidden_size
        feats = eng.tap(4)[:B * H].view(B, H).clone()                                  # HAB_TAP_RNN_OUT
        blind = getattr(policy, "is_blind", False) or not (policy._engine_kwargs.get("has_rgb") or policy._engine_kwargs.get("has_depth")
                                                          or policy._engine_kwargs.get("has_semantic"))
        ctx.has_perc = not blind
        ctx.use_extra = len(policy.aux_loss_modules) > 0  # (decided here: no device read-back to find out whether a gradient is zero)
        perc = eng.tap(3).view(B, -1)[:, :H].clone() if not blind else v.new_zeros(B, 0)  # HAB_TAP_RNN_IN[:, :hidden]
        return v, lp, ent, feats, perc

    @staticmethod
    def backward(ctx, dv, dlp, dent, dfeat, dperc):
        pol, call = ctx.policy, ctx.call
        if ctx.use_extra:  # gradients that reached the two aux_loss_state tensors from the auxiliary losses
            xf = dfeat.contiguous() if dfeat is not None else None
            xp = dperc.contiguous() if (ctx.has_perc and dperc is not None) else None
            pol.engine.set_extra_grads(xf, xp)
            ctx.keep = (xf, xp)  # alive until the backward below has been enqueued
        pol._backward_dense(call, dv.contiguous().view(-1), dlp.contiguous().view(-1), dent.contiguous().view(-1))
        if pol._dense_grad_sync is not None:  # DD-PPO: what DistributedDataParallel does inside backward (ddppo.py:110-157)
            pol._dense_grad_sync()
        grads = tuple(g.clone() for k, g in pol.engine.grad_views.items() if k not in pol.engine.buffer_names)
        return (None, None) + grads


class NetPolicy(nn.Module, Policy):
    """Engine-backed counterpart of rl/ppo/policy.py:252-413 (categorical action distribution)."""

    action_distribution_type = "categorical"

    def __init__(self, action_space, engine_kwargs: dict, init_fn, buffer_names=()):
        Policy.__init__(self, action_space)
        nn.Module.__init__(self)
        self.dim_actions = get_num_actions(action_space)
        self._engine_kwargs = dict(engine_kwargs, num_actions=self.dim_actions)
        self.action_distribution_type = engine_kwargs.get("action_dist", "categorical")
        self.engine: Optional[PolicyEngine] = None
        self._dense_grad_sync = None  # set by the DD-PPO updater: averages the gradient arena over ranks (autograd bridge only)
        self.device = torch.device("cpu")
        self._hidden = engine_kwargs["hidden"]
        self._rnn_type = engine_kwargs["rnn_type"].upper()
        self._rnn_layers = engine_kwargs["rnn_layers"]
        # CPU staging of the parameters in reference state_dict order, initialised like the reference.
        for name, value in init_fn().items():
            if name in buffer_names:
                _attach(self, name, value, buffer=True)
            else:
                _attach(self, name, nn.Parameter(value))
        self.aux_loss_modules = nn.ModuleDict()  # filled by _build_aux_modules (subclass ctors): rl/ppo/policy.py:287-289,592-608
        self.train()

    # ---- reference properties -----------------------------------------------------------------
    @property
    def hidden_state_shape(self): return (self.num_recurrent_layers, self.recurrent_hidden_size)
    @property
    def hidden_state_shape_lens(self): return [self.recurrent_hidden_size]
    @property
    def recurrent_hidden_size(self) -> int: return self._hidden
    @property
    def num_recurrent_layers(self) -> int: return self._rnn_layers * (2 if self._rnn_type == "LSTM" else 1)
    @property
    def visual_encoder(self): return self._modules["net"]._modules.get("visual_encoder")

    def _get_policy_components(self) -> List[nn.Module]:
        return [self._modules["net"], self._modules["critic"], self._modules["action_distribution"]]

    def aux_loss_parameters(self): return {k: v.parameters() for k, v in self.aux_loss_modules.items()}

    def _build_aux_modules(self, aux_loss_config, action_space) -> None:
        """get_aux_modules (rl/ppo/policy.py:592-608): one module per entry of habitat_baselines.rl.auxiliary_losses, looked up in the
        registry (`baseline_registry.register_auxiliary_loss`) and built as `cls(action_space, net, **cfg)`.  `net` is the policy's
        parameter container with the attributes the reference's losses read off a Net (output_size, perception_embedding_size,
        num_recurrent_layers, recurrent_hidden_size, is_blind).  The modules are ordinary torch modules: they live outside the flat
        arena, are called from evaluate_actions on the autograd bridge and optimised by the updater's second Adam (ppo.py)."""
        if not aux_loss_config:
            return
        net = self._modules["net"]
        net.output_size = net.recurrent_hidden_size = net.perception_embedding_size = self._hidden
        net.num_recurrent_layers = self.num_recurrent_layers
        net.is_blind = not (self._engine_kwargs.get("has_rgb") or self._engine_kwargs.get("has_depth") or self._engine_kwargs.get("has_semantic"))
        items = aux_loss_config.items() if hasattr(aux_loss_config, "items") else aux_loss_config
        for name, cfg in items:
            cls = baseline_registry.get_auxiliary_loss(str(name))
            if cls is None:
                raise _lib.HabError(f"auxiliary loss '{name}' is not registered (baseline_registry.register_auxiliary_loss)")
            self.aux_loss_modules[str(name)] = cls(action_space, net, **dict(cfg))

    def forward(self, *x):
        raise NotImplementedError

    # ---- device placement: builds the engine and re-homes parameters into its flat arena ---------
    def to(self, *args, **kwargs):
        device = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (str, torch.device)):
                device = torch.device(a)
        if device is None:
            raise _lib.HabError("habitat_amd policies are fp32-only; .to() accepts a device")
        return self._materialize(device)

    def cuda(self, device=None):
        return self._materialize(torch.device("cuda" if device is None else device))

    def _materialize(self, device: torch.device):
        if device.type != "cuda":
            if self.engine is None:
                return self  # still staged on the CPU; nothing can be *computed* there
            raise _lib.HabError("habitat_amd policies cannot be moved off the GPU (no CPU execution path)")
        if self.engine is not None and self.device == device:
            return self
        with torch.cuda.device(device):
            eng = PolicyEngine(device=device, **self._engine_kwargs)
        names = [s[0] for s in eng.specs]
        # modules a caller hung on the policy afterwards (test_ddppo_reduce.py:60-61 `actor_critic.unused = nn.Linear(64, 64)`) are not
        # the engine's: they move like any nn.Module and stay outside the arena, the fused optimiser and the gradient exchange
        own = set(names)
        state = {k: v.detach() for k, v in self.state_dict().items() if k in own}
        for nm, m in self._modules.items():
            if nm not in ("net", "critic", "action_distribution") and m is not None:
                m.to(device)
        assert names == list(state.keys()), "engine parameter table does not match the module's parameters"
        for nm, shp, _ in eng.specs:
            assert tuple(state[nm].shape) == shp, (nm, state[nm].shape, shp)
        with torch.cuda.device(device):
            eng.load({k: v.to(device) for k, v in state.items()})
        for nm in names:  # swap the staged CPU parameters for views into the flat arena
            *path, leaf = nm.split(".")
            m = self
            for p in path:
                m = m._modules[p]
            if nm in eng.buffer_names:
                m._buffers[leaf] = eng.views[nm]
                continue
            par = nn.Parameter(eng.views[nm], requires_grad=m._parameters[leaf].requires_grad)
            if par.requires_grad:
                par.grad = eng.grad_views[nm]
            m._parameters[leaf] = par
        self.engine, self.device = eng, device
        eng.set_training(self.training)
        ve = self.visual_encoder
        if ve is not None and self._engine_kwargs.get("arch") == "resnet" and (self._engine_kwargs.get("has_rgb") or
                                                                               self._engine_kwargs.get("has_depth") or
                                                                               self._engine_kwargs.get("has_semantic")):
            # the reference calls `actor_critic.visual_encoder(batch)` and reads `.output_shape` (ppo_trainer.py:271-279)
            ve.output_shape = eng.visual_feature_shape()
            ve.forward = self.encode_visual
        return self

    @torch.no_grad()
    def encode_visual(self, observations, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ResNetEncoder.forward alone (resnet_policy.py:255-276): (n, C, Hf, Wf), what the rollout stores under
        `visual_features` when the encoder is frozen.  RunningMeanAndVar follows the module's training flag."""
        eng = self._require_engine()
        obs = {k: v for k, v in observations.items() if k != VISUAL_FEATURES_KEY}
        rgb, depth, _, extra = self._obs_ptrs(obs)
        ref = rgb if rgb is not None else depth
        n = ref.shape[0]
        if out is None:
            out = torch.empty((n,) + tuple(eng.visual_feature_shape()), device=self.device)
        assert out.is_contiguous() and out.shape[0] == n
        eng.encode(rgb, depth, n, out, extra=extra)
        return out

    def train(self, mode: bool = True):
        """nn.Module.train / eval: the engine's RunningMeanAndVar only updates in training mode."""
        super().train(mode)
        if getattr(self, "engine", None) is not None:
            self.engine.set_training(mode)
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        out = super().load_state_dict(state_dict, strict=strict)
        if self.engine is not None:
            self.engine.repack()
        return out

    def _require_engine(self) -> PolicyEngine:
        if self.engine is None:
            raise _lib.HabError("policy is not on a GPU yet: call .to('cuda') first (habitat_amd has no CPU execution path)")
        return self.engine

    # ---- helpers --------------------------------------------------------------------------------
    def _obs_ptrs(self, observations):
        """(rgb, depth, goal, extra) tensors of an observation dict, checked for the dtypes / layout the kernels read in place."""
        kw = self._engine_kwargs
        rgb = observations["rgb"] if kw["has_rgb"] else None
        depth = observations["depth"] if kw["has_depth"] else None
        goal = observations[getattr(self, "goal_key", GOAL_UUID)] if kw.get("goal_dim", 2) > 0 else None
        extra = {}
        if kw.get("has_semantic"):
            extra["semantic"] = observations["semantic"]
        if kw.get("num_object_categories", 0) > 0:
            extra["objectgoal"] = observations["objectgoal"]
        if kw.get("has_compass"):
            extra["compass"] = observations["compass"]
        if kw.get("has_gps"):
            extra["gps"] = observations["gps"]
        if kw.get("arch") == "resnet" and kw.get("pointgoal_dim", 0) > 0:
            extra["pointgoal"] = observations[POINTGOAL_UUID]
        if kw.get("proximity_dim", 0) > 0:
            extra["proximity"] = observations["proximity"]
        if VISUAL_FEATURES_KEY in observations:  # frozen encoder: the rollout holds its output (resnet_policy.py:636-646)
            extra["visual_features"] = observations[VISUAL_FEATURES_KEY]
        want = {"rgb": torch.uint8, "depth": torch.float32, "goal": torch.float32, "semantic": torch.int32, "objectgoal": torch.int64,
                "compass": torch.float32, "gps": torch.float32, "visual_features": torch.float32, "pointgoal": torch.float32,
                "proximity": torch.float32}
        for name, t in dict(rgb=rgb, depth=depth, goal=goal, **extra).items():
            if t is None:
                continue
            if t.dtype != want[name]:
                raise _lib.HabError(f"observation '{name}' must be {want[name]} (got {t.dtype})")
            if not t.is_contiguous():
                raise _lib.HabError("observation tensors must be contiguous NHWC")
        return rgb, depth, goal, extra

    def draw_noise(self, n: int) -> torch.Tensor:
        """The draw the reference's sampling makes from the CPU generator: Exp(1) noise for torch.multinomial
        (utils/common.py:64-68), N(0, 1) for CustomNormal.rsample (:99-103, torch.distributions.Normal.rsample)."""
        if self.action_distribution_type == "gaussian":
            q = torch.empty(n, self.dim_actions).normal_()
        else:
            q = torch.empty(n, self.dim_actions).exponential_(1)
        return q.pin_memory().to(self.device, non_blocking=True) if self.device.type == "cuda" else q

    # ---- reference API --------------------------------------------------------------------------
    @torch.no_grad()
    def act(self, observations, rnn_hidden_states, prev_actions, masks, deterministic=False, exp_noise=None, out=None):
        eng = self._require_engine()
        rgb, depth, goal, extra = self._obs_ptrs(observations)
        n = rnn_hidden_states.shape[0]
        dev = self.device
        if out is None:
            acts = (torch.empty(n, self.dim_actions, device=dev) if self.action_distribution_type == "gaussian"
                    else torch.empty(n, 1, dtype=torch.long, device=dev))
            out = dict(values=torch.empty(n, 1, device=dev), actions=acts,
                       action_log_probs=torch.empty(n, 1, device=dev),
                       rnn_hidden_states=torch.empty(n, self.num_recurrent_layers, self._hidden, device=dev))
        if not deterministic and exp_noise is None:
            exp_noise = self.draw_noise(n)
        eng.act(rgb, depth, goal, rnn_hidden_states.contiguous(), masks.contiguous(), n, exp_noise=exp_noise,
                deterministic=deterministic, values=out["values"], actions=out["actions"],
                action_log_probs=out["action_log_probs"], hidden_out=out["rnn_hidden_states"],
                prev_actions=prev_actions, extra=extra)
        return PolicyActionData(values=out["values"], actions=out["actions"], action_log_probs=out["action_log_probs"],
                                rnn_hidden_states=out["rnn_hidden_states"])

    @torch.no_grad()
    def get_value(self, observations, rnn_hidden_states, prev_actions, masks):
        eng = self._require_engine()
        rgb, depth, goal, extra = self._obs_ptrs(observations)
        n = rnn_hidden_states.shape[0]
        values = torch.empty(n, 1, device=self.device)
        eng.act(rgb, depth, goal, rnn_hidden_states.contiguous(), masks.contiguous(), n, values=values, prev_actions=prev_actions,
                extra=extra)
        return values

    def evaluate_actions(self, observations, rnn_hidden_states, prev_actions, masks, action, rnn_build_seq_info):
        """Dense-tensor entry (reference calling convention).  Differentiable through _EvaluateFn."""
        self._require_engine()
        rgb, depth, goal, extra = self._obs_ptrs(observations)
        B, n = masks.shape[0], rnn_hidden_states.shape[0]
        pack = _pack_from_seq_info(rnn_build_seq_info, B, n, self.device)
        call = dict(rgb=rgb, depth=depth, goal=goal, extra=extra, hidden0=rnn_hidden_states.contiguous(), masks=masks.contiguous(),
                    actions=action.contiguous(), prev_actions=prev_actions, pack=pack, B=B, n=n)
        aux_loss_res = {}
        if torch.is_grad_enabled():
            own = self.engine.grad_views
            v, lp, ent, feats, perc = _EvaluateFn.apply(self, call, *(p_ for k, p_ in self.named_parameters() if k in own))
            if len(self.aux_loss_modules) > 0:  # rl/ppo/policy.py:386-394
                aux_loss_state = {"rnn_output": feats}
                if perc.shape[1] > 0:
                    aux_loss_state["perception_embed"] = perc
                batch = dict(observations=observations, rnn_hidden_states=rnn_hidden_states, prev_actions=prev_actions, masks=masks,
                             action=action, rnn_build_seq_info=rnn_build_seq_info)
                aux_loss_res = {k: m(aux_loss_state, batch) for k, m in self.aux_loss_modules.items()}
        else:
            v, lp, ent = self._evaluate_dense(call)
        hidden = torch.empty(n, self.num_recurrent_layers, self._hidden, device=self.device)
        self.engine.final_hidden(hidden)
        return v, lp, ent, hidden, aux_loss_res

    def _evaluate_dense(self, c):
        dev, B = self.device, c["B"]
        v, lp, ent = (torch.empty(B, 1, device=dev) for _ in range(3))
        self.engine.evaluate(c["rgb"], c["depth"], c["goal"], None, c["hidden0"], c["masks"], c["actions"], c["pack"], B, c["n"],
                             value=v, log_prob=lp, entropy=ent, prev_actions=c["prev_actions"], extra=c["extra"])
        return v, lp, ent

    def _backward_dense(self, c, dv, dlp, dent):
        self.engine.backward(c["rgb"], c["depth"], c["goal"], None, c["actions"], c["pack"], dv, dlp, dent,
                             prev_actions=c["prev_actions"], extra=c["extra"])


def _pack_from_seq_info(info, B, n, device) -> DevicePackInfo:
    """Accepts either our DevicePackInfo or a reference-style rnn_build_seq_info dict
    (rl/models/rnn_state_encoder.py:171-184) and returns the engine's view of it."""
    if isinstance(info, DevicePackInfo):
        return info
    if hasattr(info, "pack"):
        return info.pack
    g = lambda k: (info["cpu_" + k] if ("cpu_" + k) in info else info[k].cpu()).numpy()
    T = B // n
    # rebuild through the dones implied by the fragment starts (tie order of the caller's arrays is irrelevant)
    starts = g("sequence_starts")
    dones = np.zeros((T, n), dtype=np.uint8)
    for s in starts:
        t, e = divmod(int(s), n)
        if t > 0:
            dones[t, e] = 1
    return DevicePackInfo(dones, device)


def _baseline_init(cin, H, W, hidden, num_actions, goal_dim):
    """Parameter values exactly as PointNavBaselinePolicy.__init__ produces them (same initialisers, same order of
    RNG consumption): SimpleCNN.layer_init simple_cnn.py:126-133, RNNStateEncoder.layer_init
    rnn_state_encoder.py:288-293, CategoricalNet utils/common.py:90-91, CriticHead policy.py:420-421."""
    def co(x, k, s): return (x - k) // s + 1
    h, w = co(co(co(H, 8, 4), 4, 2), 3, 1), co(co(co(W, 8, 4), 4, 2), 3, 1)
    out = {}
    # PointNavBaselineNet.__init__ builds SimpleCNN first (policy.py:530), then the RNN (:532-535)
    # (blind -- no visual sensor -- : SimpleCNN.cnn is an empty nn.Sequential, simple_cnn.py:95-97: no parameters, no RNG consumed)
    layers = [] if cin == 0 else [("0", nn.Conv2d(cin, 32, 8, 4)), ("2", nn.Conv2d(32, 64, 4, 2)), ("4", nn.Conv2d(64, 32, 3, 1)),
                                  ("6", nn.Linear(32 * h * w, hidden))]
    for _, layer in layers:
        nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
        nn.init.constant_(layer.bias, val=0)
    for idx, layer in layers:
        out[f"net.visual_encoder.cnn.{idx}.weight"] = layer.weight.detach()
        out[f"net.visual_encoder.cnn.{idx}.bias"] = layer.bias.detach()
    rnn = nn.GRU(input_size=(hidden if cin else 0) + goal_dim, hidden_size=hidden, num_layers=1)  # policy.py:532-535
    for name, param in rnn.named_parameters():
        if "weight" in name:
            nn.init.orthogonal_(param)
        elif "bias" in name:
            nn.init.constant_(param, 0)
    for name, param in rnn.named_parameters():
        out[f"net.state_encoder.rnn.{name}"] = param.detach()
    # NetPolicy.__init__: CategoricalNet (policy.py:273-276) then CriticHead (:291)
    lin = nn.Linear(hidden, num_actions)
    nn.init.orthogonal_(lin.weight, gain=0.01)
    nn.init.constant_(lin.bias, 0)
    out["action_distribution.linear.weight"], out["action_distribution.linear.bias"] = lin.weight.detach(), lin.bias.detach()
    fc = nn.Linear(hidden, 1)
    nn.init.orthogonal_(fc.weight)
    nn.init.constant_(fc.bias, 0)
    out["critic.fc.weight"], out["critic.fc.bias"] = fc.weight.detach(), fc.bias.detach()
    return out


@baseline_registry.register_policy
class PointNavBaselinePolicy(NetPolicy):
    """SimpleCNN + GRU policy (rl/ppo/policy.py:427-589) on the HIP engine."""

    def __init__(self, observation_space, action_space, hidden_size: int = 512, aux_loss_config=None, max_frames: int = 4096,
                 max_envs: int = 64, **kwargs):
        sp = observation_space.spaces
        has_rgb, has_depth = "rgb" in sp, "depth" in sp
        # goal sensor: IntegratedPointGoalGPSAndCompassSensor first, then PointGoalSensor (policy.py:504-514); the ImageGoalSensor variant
        # builds a second SimpleCNN for the goal image (:515-522) and is outside the accelerated path
        goal_key = GOAL_UUID if GOAL_UUID in sp else (POINTGOAL_UUID if POINTGOAL_UUID in sp else None)
        if goal_key is None:
            raise _lib.HabError(f"PointNavBaselinePolicy on habitat_amd needs the '{GOAL_UUID}' or the '{POINTGOAL_UUID}' sensor"
                                + (" (an 'imagegoal' goal encoder is outside the accelerated path)" if "imagegoal" in sp else ""))
        # blind (no rgb / depth, SimpleCNN.is_blind simple_cnn.py:54): the net is goal -> GRU -> heads -- the configuration of the
        # reference's own DD-PPO test (test/test_ddppo_reduce.py:43-56)
        vis = sp["rgb"] if has_rgb else (sp["depth"] if has_depth else None)
        H, W = (int(vis.shape[0]), int(vis.shape[1])) if vis is not None else (0, 0)
        cin = (3 if has_rgb else 0) + (1 if has_depth else 0)
        goal_dim = int(sp[goal_key].shape[0])
        na = get_num_actions(action_space)
        super().__init__(action_space,
                         dict(arch="simple_cnn", rnn_type="GRU", rnn_layers=1, hidden=hidden_size, H=H, W=W, has_rgb=has_rgb,
                              has_depth=has_depth, goal_dim=goal_dim, max_frames=max_frames, max_envs=max_envs),
                         lambda: _baseline_init(cin, H, W, hidden_size, na, goal_dim))
        self.goal_key = goal_key
        self._build_aux_modules(aux_loss_config, action_space)

    @classmethod
    def from_config(cls, config, observation_space, action_space, **kwargs):
        hb = config.habitat_baselines
        ppo = hb.rl.ppo
        n_envs = int(hb.num_environments)
        return cls(observation_space=observation_space, action_space=action_space, hidden_size=ppo.hidden_size,
                   aux_loss_config=hb.rl.auxiliary_losses,
                   max_frames=int(ppo.num_steps) * max(1, -(-n_envs // int(ppo.num_mini_batch))), max_envs=n_envs)


# rl/ddppo/policy/resnet.py:296-345 -> (HAB_BACKBONE_* code, block kind, stage depths, ResNeXt?, SE?)
BACKBONES = {"resnet18": (18, "basic", [2, 2, 2, 2], False, False), "resnet50": (50, "bottleneck", [3, 4, 6, 3], False, False),
             "resneXt50": (51, "bottleneck", [3, 4, 6, 3], True, False), "se_resnet50": (52, "bottleneck", [3, 4, 6, 3], False, True),
             "se_resneXt50": (53, "bottleneck", [3, 4, 6, 3], True, True), "se_resneXt101": (101, "bottleneck", [3, 4, 23, 3], True, True)}


def _resnet_init(n_in, hidden, num_actions, rnn_type, rnn_layers, backbone, baseplanes, H, W, normalize, has_goal=True, n_obj=0,
                 has_gps=False, has_compass=False, gauss=None, pointgoal_dim=0, proximity_dim=0, blind=False):
    """Parameter / buffer values exactly as PointNavResNetPolicy.__init__ produces them: the torch modules are created in
    the reference's order (resnet_policy.py:389-396 embedding, :454-456 tgt_embeding, :578-585 ResNetEncoder [default
    Conv2d / GroupNorm initialisers -- ResNetEncoder.layer_init is never called], :588-595 visual_fc, :597-602 state encoder
    with orthogonal / zero init rnn_state_encoder.py:288-293), then CategoricalNet and CriticHead (policy.py:273-291)."""
    out = {}
    if gauss is not None:  # continuous actions: nn.Linear(num_actions, 32) (resnet_policy.py:424-428)
        emb = nn.Linear(num_actions, 32)
        out["net.prev_action_embedding.weight"], out["net.prev_action_embedding.bias"] = emb.weight.detach(), emb.bias.detach()
    else:
        emb = nn.Embedding(num_actions + 1, 32)
        out["net.prev_action_embedding.weight"] = emb.weight.detach()
    n_slots = 1
    if has_goal:  # module creation order of PointNavResNetNet.__init__ (resnet_policy.py:441-528)
        tgt = nn.Linear(3, 32)
        out["net.tgt_embeding.weight"], out["net.tgt_embeding.bias"] = tgt.weight.detach(), tgt.bias.detach()
        n_slots += 1
    if n_obj > 0:
        obj = nn.Embedding(n_obj, 32)
        out["net.obj_categories_embedding.weight"] = obj.weight.detach()
        n_slots += 1
    if has_gps:
        gps = nn.Linear(2, 32)
        out["net.gps_embedding.weight"], out["net.gps_embedding.bias"] = gps.weight.detach(), gps.bias.detach()
        n_slots += 1
    if pointgoal_dim > 0:  # resnet_policy.py:489-494
        pg = nn.Linear(pointgoal_dim, 32)
        out["net.pointgoal_embedding.weight"], out["net.pointgoal_embedding.bias"] = pg.weight.detach(), pg.bias.detach()
        n_slots += 1
    if proximity_dim > 0:  # :510-515
        px = nn.Linear(proximity_dim, 32)
        out["net.proximity_embedding.weight"], out["net.proximity_embedding.bias"] = px.weight.detach(), px.bias.detach()
        n_slots += 1
    if has_compass:
        cmp_ = nn.Linear(2, 32)
        out["net.compass_embedding.weight"], out["net.compass_embedding.bias"] = cmp_.weight.detach(), cmp_.bias.detach()
        n_slots += 1
    ve = "net.visual_encoder."
    if normalize:
        out[ve + "running_mean_and_var._mean"] = torch.zeros(1, n_in, 1, 1)
        out[ve + "running_mean_and_var._var"] = torch.zeros(1, n_in, 1, 1)
        out[ve + "running_mean_and_var._count"] = torch.zeros(())
    ng = baseplanes // 2
    _, kind, layers, resnext, se = BACKBONES[backbone]
    bottleneck = kind == "bottleneck"
    expansion = (2 if resnext else 4) if bottleneck else 1
    cardinality = baseplanes // 2 if resnext else 1

    def conv_gn(prefix_w, prefix_g, cin, cout, k, groups, conv_groups=1):
        conv = nn.Conv2d(cin, cout, kernel_size=k, bias=False, groups=conv_groups)
        gn = nn.GroupNorm(groups, cout)
        out[prefix_w + ".weight"] = conv.weight.detach()
        out[prefix_g + ".weight"], out[prefix_g + ".bias"] = gn.weight.detach(), gn.bias.detach()

    bb = ve + "backbone."
    if not blind:
        conv_gn(bb + "conv1.0", bb + "conv1.1", n_in, baseplanes, 7, ng)
    inplanes = baseplanes
    for li, nblocks in enumerate(layers if not blind else ()):
        planes = (2 * baseplanes if resnext else baseplanes) * (2 ** li)  # resnet.py:223-225
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and li > 0) else 1
            bp = f"{bb}layer{li + 1}.{bi}."
            has_ds = bi == 0 and (stride != 1 or inplanes != planes * expansion)
            # resnet.py:_make_layer builds the downsample modules BEFORE the block's own convs
            ds = None
            if has_ds:
                ds = (nn.Conv2d(inplanes, planes * expansion, kernel_size=1, bias=False), nn.GroupNorm(ng, planes * expansion))
            if not bottleneck:
                conv_gn(bp + "convs.0", bp + "convs.1", inplanes, planes, 3, ng)
                conv_gn(bp + "convs.3", bp + "convs.4", planes, planes, 3, ng)
            else:
                conv_gn(bp + "convs.0", bp + "convs.1", inplanes, planes, 1, ng)
                # only the first block of a stage receives the cardinality (resnet.py:257-268)
                conv_gn(bp + "convs.3", bp + "convs.4", planes, planes, 3, ng, conv_groups=cardinality if bi == 0 else 1)
                conv_gn(bp + "convs.6", bp + "convs.7", planes, planes * expansion, 1, ng)
            se_mods = None
            if se and bottleneck:  # SE is created after the block's convs (resnet.py:166-176), registered after downsample
                c_se = planes * expansion
                se_mods = (nn.Linear(c_se, int(c_se / 16)), nn.Linear(int(c_se / 16), c_se))
            if ds is not None:
                out[bp + "downsample.0.weight"] = ds[0].weight.detach()
                out[bp + "downsample.1.weight"], out[bp + "downsample.1.bias"] = ds[1].weight.detach(), ds[1].bias.detach()
            if se_mods is not None:
                out[bp + "se.excite.0.weight"], out[bp + "se.excite.0.bias"] = se_mods[0].weight.detach(), se_mods[0].bias.detach()
                out[bp + "se.excite.2.weight"], out[bp + "se.excite.2.bias"] = se_mods[1].weight.detach(), se_mods[1].bias.detach()
            inplanes = planes * expansion
    if not blind:  # (is_blind: no backbone, no compression, no visual_fc -- resnet_policy.py:200-246,586-595)
        fh, fw = int(np.ceil((H // 2) / 32.0)), int(np.ceil((W // 2) / 32.0))
        ncomp = int(round(2048 / (fh * fw)))
        conv_gn(ve + "compression.0", ve + "compression.1", inplanes, ncomp, 3, 1)
        fc = nn.Linear(ncomp * fh * fw, hidden)
        out["net.visual_fc.1.weight"], out["net.visual_fc.1.bias"] = fc.weight.detach(), fc.bias.detach()
    rnn_cls = nn.LSTM if rnn_type == "LSTM" else nn.GRU
    rnn = rnn_cls(input_size=(0 if blind else hidden) + 32 * n_slots, hidden_size=hidden, num_layers=rnn_layers)
    for name, param in rnn.named_parameters():
        if "weight" in name:
            nn.init.orthogonal_(param)
        elif "bias" in name:
            nn.init.constant_(param, 0)
    for name, param in rnn.named_parameters():
        out[f"net.state_encoder.rnn.{name}"] = param.detach()
    if gauss is not None:  # GaussianNet.__init__ (utils/common.py:112-149)
        if gauss["use_std_param"]:
            out["action_distribution.std"] = (torch.randn(num_actions) * 0.01 + gauss["std_init"]).detach()
        k = num_actions if gauss["use_std_param"] else 2 * num_actions
        lin = nn.Linear(hidden, k)
        nn.init.orthogonal_(lin.weight, gain=0.01)
        nn.init.constant_(lin.bias, 0)
        if not gauss["use_std_param"]:
            with torch.no_grad():
                lin.bias[num_actions:].fill_(gauss["std_init"])
        out["action_distribution.mu_maybe_std.weight"], out["action_distribution.mu_maybe_std.bias"] = lin.weight.detach(), lin.bias.detach()
    else:
        lin = nn.Linear(hidden, num_actions)
        nn.init.orthogonal_(lin.weight, gain=0.01)
        nn.init.constant_(lin.bias, 0)
        out["action_distribution.linear.weight"], out["action_distribution.linear.bias"] = lin.weight.detach(), lin.bias.detach()
    fcv = nn.Linear(hidden, 1)
    nn.init.orthogonal_(fcv.weight)
    nn.init.constant_(fcv.bias, 0)
    out["critic.fc.weight"], out["critic.fc.bias"] = fcv.weight.detach(), fcv.bias.detach()
    return out


def _gaussian_options(ad):
    """ActionDistributionConfig (default_structured_configs.py:70-85) -> (init options, engine kwargs), GaussianNet.__init__'s
    choices (utils/common.py:118-140): which raw quantity the linear layer / parameter produces and its clamp range."""
    import math
    g = lambda k, d: (ad.get(k, d) if isinstance(ad, dict) else getattr(ad, k, d))
    use_log_std, use_softplus = bool(g("use_log_std", True)), bool(g("use_softplus", False))
    use_std_param, clamp_std = bool(g("use_std_param", False)), bool(g("clamp_std", True))
    if use_log_std:
        lo, hi, std_init = float(g("min_log_std", -5)), float(g("max_log_std", 2)), float(g("log_std_init", 0.0))
    elif use_softplus:
        inv = lambda x: math.log(math.exp(x) - 1)
        lo, hi, std_init = inv(float(g("min_std", 1e-6))), inv(float(g("max_std", 1))), inv(1.0)
    else:
        lo, hi, std_init = float(g("min_std", 1e-6)), float(g("max_std", 1)), 1.0
    flags = ((_lib.GAUSS_TANH_MU if g("action_activation", "tanh") == "tanh" else 0) | (_lib.GAUSS_USE_LOG_STD if use_log_std else 0)
             | (_lib.GAUSS_USE_SOFTPLUS if use_softplus else 0) | (_lib.GAUSS_USE_STD_PARAM if use_std_param else 0)
             | (_lib.GAUSS_CLAMP_STD if clamp_std else 0))
    return (dict(use_std_param=use_std_param, std_init=std_init),
            dict(action_dist="gaussian", gauss_flags=flags, gauss_min_std=lo, gauss_max_std=hi))


@baseline_registry.register_policy
class PointNavResNetPolicy(NetPolicy):
    """GroupNorm-ResNet + GRU/LSTM policy (rl/ddppo/policy/resnet_policy.py:50-162,391-767) on the HIP engine.
    Supported on the accelerated path: backbones resnet18 / resnet50, rgb and/or depth visual sensors, the
    pointgoal_with_gps_compass goal (2-D polar), discrete actions."""

    def __init__(self, observation_space, action_space, hidden_size: int = 512, num_recurrent_layers: int = 1,
                 rnn_type: str = "GRU", resnet_baseplanes: int = 32, backbone: str = "resnet18",
                 normalize_visual_inputs: bool = False, force_blind_policy: bool = False, policy_config=None,
                 aux_loss_config=None, fuse_keys=None, max_frames: int = 4096, max_envs: int = 64, **kwargs):
        sp = observation_space.spaces
        if backbone not in BACKBONES:
            raise _lib.HabError(f"backbone {backbone!r} is not one of {sorted(BACKBONES)} (rl/ddppo/policy/resnet.py:296-345)")
        gauss = gauss_kw = None
        dist = getattr(policy_config, "action_distribution_type", "categorical") if policy_config is not None else "categorical"
        if dist == "gaussian":
            gauss, gauss_kw = _gaussian_options(policy_config.action_dist)
        elif dist != "categorical":
            raise ValueError(f"Action distribution {dist} not supported.")
        visual_keys = [k for k, v in sp.items() if len(v.shape) > 1]  # observation-space order (resnet_policy.py:178-182)
        # force_blind_policy (resnet_policy.py:553-554): the visual encoder is built on an EMPTY observation space -- the images stay in
        # the observation dict (the rollout still stores them) but no backbone, compression or visual_fc exists and the recurrent
        # encoder's input is the embeddings alone.  An observation space without images gives the same net.
        blind = bool(force_blind_policy) or not visual_keys
        image_keys, visual_keys = visual_keys, ([] if blind else visual_keys)
        if blind and normalize_visual_inputs:
            # the reference builds RunningMeanAndVar(0) there and fails its `assert n_channels > 0` (running_mean_and_var.py:16): a blind
            # policy exists only without input normalisation (from_config turns it on when the observation space has "rgb")
            raise AssertionError("normalize_visual_inputs needs at least one visual channel (blind policy)")
        # 1-D sensors with an embedding on the accelerated path (resnet_policy.py:454-515,662-734).  `heading` is refused: the reference's
        # forward embeds sensor_observations[0] -- the FIRST ROW of the batch (:705-713) -- which only type-checks for one environment;
        # `imagegoal` / `instance_imagegoal` build a second ResNetEncoder for the goal image (:517-545)
        known_1d = {GOAL_UUID, POINTGOAL_UUID, "proximity", "objectgoal", "compass", "gps"}
        other = [k for k in sp.keys() if k not in image_keys and k not in known_1d]
        if any(k not in ("rgb", "depth", "semantic") for k in image_keys) or other:
            raise _lib.HabError("PointNavResNetPolicy on habitat_amd supports the rgb / depth / semantic visual sensors and the "
                                f"pointgoal_with_gps_compass, pointgoal, proximity, objectgoal, compass, gps 1-D sensors (got {list(sp.keys())})")
        has_rgb, has_depth, has_sem = "rgb" in visual_keys, "depth" in visual_keys, "semantic" in visual_keys
        H, W = (int(sp[visual_keys[0]].shape[0]), int(sp[visual_keys[0]].shape[1])) if visual_keys else (0, 0)
        n_in = (3 if has_rgb else 0) + (1 if has_depth else 0) + (1 if has_sem else 0)
        na = get_num_actions(action_space)
        rnn_type = rnn_type.upper()
        has_goal = GOAL_UUID in sp
        if has_goal and int(sp[GOAL_UUID].shape[0]) != 2:
            raise _lib.HabError("only the 2-D polar pointgoal_with_gps_compass is on the accelerated path")
        n_obj = int(sp["objectgoal"].high[0]) + 1 if "objectgoal" in sp else 0  # resnet_policy.py:468-476
        has_gps, has_compass = "gps" in sp, "compass" in sp
        if has_gps and int(sp["gps"].shape[0]) != 2:
            raise _lib.HabError("gps sensor must be 2-D")
        pg_dim = int(sp[POINTGOAL_UUID].shape[0]) if POINTGOAL_UUID in sp else 0
        px_dim = int(sp["proximity"].shape[0]) if "proximity" in sp else 0
        if pg_dim > 4 or px_dim > 4:
            raise _lib.HabError("pointgoal / proximity sensors of more than 4 dimensions are outside the accelerated path")
        bufs = tuple("net.visual_encoder.running_mean_and_var." + k for k in ("_mean", "_var", "_count")) if normalize_visual_inputs else ()
        super().__init__(action_space,
                         dict(arch="resnet", backbone=BACKBONES[backbone][0], baseplanes=resnet_baseplanes,
                              normalize_visual_inputs=bool(normalize_visual_inputs), rnn_type=rnn_type,
                              rnn_layers=num_recurrent_layers, hidden=hidden_size, H=H, W=W, has_rgb=has_rgb, has_depth=has_depth,
                              goal_dim=2 if has_goal else 0, max_frames=max_frames, max_envs=max_envs,
                              visual_order=tuple(visual_keys), has_semantic=has_sem, num_object_categories=n_obj,
                              has_compass=has_compass, has_gps=has_gps, pointgoal_dim=pg_dim, proximity_dim=px_dim, **(gauss_kw or {})),
                         lambda: _resnet_init(n_in, hidden_size, na, rnn_type, num_recurrent_layers, backbone, resnet_baseplanes,
                                              H, W, normalize_visual_inputs, has_goal, n_obj, has_gps, has_compass, gauss,
                                              pointgoal_dim=pg_dim, proximity_dim=px_dim, blind=blind),
                         buffer_names=bufs)
        self.is_blind = blind
        self._build_aux_modules(aux_loss_config, action_space)

    @classmethod
    def from_config(cls, config, observation_space, action_space, **kwargs):
        hb = config.habitat_baselines
        ppo, dd = hb.rl.ppo, hb.rl.ddppo
        n_envs = int(hb.num_environments)
        return cls(observation_space=observation_space, action_space=action_space, hidden_size=ppo.hidden_size,
                   rnn_type=dd.rnn_type, num_recurrent_layers=dd.num_recurrent_layers, backbone=dd.backbone,
                   normalize_visual_inputs="rgb" in observation_space.spaces,
                   force_blind_policy=getattr(hb, "force_blind_policy", False),
                   policy_config=hb.rl.policy[kwargs.get("agent_name") or "main_agent"],
                   aux_loss_config=getattr(hb.rl, "auxiliary_losses", None),
                   max_frames=int(ppo.num_steps) * max(1, -(-n_envs // int(ppo.num_mini_batch))), max_envs=n_envs)
