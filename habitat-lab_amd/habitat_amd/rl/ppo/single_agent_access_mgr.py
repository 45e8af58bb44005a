"""SingleAgentAccessMgr: builds {policy, rollout storage, updater, LR schedule} from registry names in the config.
Surface of habitat_baselines/rl/ppo/single_agent_access_mgr.py:40-298 (+ agent_access_mgr.py:17-131)."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from habitat_amd import _lib
from torch.optim.lr_scheduler import LambdaLR

from habitat_amd.common.baseline_registry import baseline_registry


def linear_lr_schedule(percent_done: float) -> float:
    return 1 - percent_done


class EnvironmentSpec:
    def __init__(self, observation_space, action_space, orig_action_space=None):
        self.observation_space, self.action_space = observation_space, action_space
        self.orig_action_space = orig_action_space if orig_action_space is not None else action_space


@baseline_registry.register_agent_access_mgr
class SingleAgentAccessMgr:
    def __init__(self, config, env_spec: EnvironmentSpec, is_distrib: bool, device, resume_state: Optional[Dict[str, Any]],
                 num_envs: int, percent_done_fn: Callable[[], float], lr_schedule_fn: Optional[Callable[[float], float]] = None,
                 agent_name=None):
        self._env_spec, self._config, self._num_envs, self._device = env_spec, config, num_envs, device
        self._ppo_cfg = config.habitat_baselines.rl.ppo
        self._is_distributed = is_distrib
        self._is_static_encoder = not config.habitat_baselines.rl.ddppo.train_encoder
        self.agent_name = agent_name or config.habitat.simulator.agents_order[0]
        self.nbuffers = 2 if self._ppo_cfg.use_double_buffered_sampler else 1
        self._percent_done_fn = percent_done_fn
        lr_schedule_fn = lr_schedule_fn or linear_lr_schedule
        self._actor_critic = self._create_policy()
        self._updater = self._create_updater(self._actor_critic)
        if self._updater.optimizer is None:
            self._lr_scheduler = None
        else:
            self._lr_scheduler = LambdaLR(optimizer=self._updater.optimizer, lr_lambda=lambda _: lr_schedule_fn(self._percent_done_fn()))
        if resume_state is not None:
            self.load_state_dict(resume_state)
        self._rollouts = None

    # ---- construction ---------------------------------------------------------------------------------
    def _create_policy(self):
        cfg = self._config
        name = cfg.habitat_baselines.rl.policy[self.agent_name].name
        policy_cls = baseline_registry.get_policy(name)
        if policy_cls is None:
            raise ValueError(f"Couldn't find policy {name}")
        actor_critic = policy_cls.from_config(cfg, self._env_spec.observation_space, self._env_spec.action_space,
                                              orig_action_space=self._env_spec.orig_action_space, agent_name=self.agent_name)
        dd = cfg.habitat_baselines.rl.ddppo
        if dd.pretrained_encoder or dd.pretrained:
            ckpt = torch.load(dd.pretrained_weights, map_location="cpu", weights_only=False)
            sd = ckpt["state_dict"]
            if dd.pretrained:
                actor_critic.load_state_dict({k[len("actor_critic."):]: v for k, v in sd.items()})
            else:
                pre = "actor_critic.net.visual_encoder."
                own = actor_critic.state_dict()
                own.update({"net.visual_encoder." + k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
                actor_critic.load_state_dict(own)
        if self._is_static_encoder:
            # single_agent_access_mgr.py:230-232.  The frozen parameters keep zero gradients in the flat arena (the engine's
            # backward stops at visual_fc when the rollout supplies `visual_features`), so the fused Adam leaves them untouched.
            if actor_critic.visual_encoder is None or not hasattr(actor_critic, "encode_visual") or \
                    actor_critic._engine_kwargs.get("arch") != "resnet":
                raise _lib.HabError("rl.ddppo.train_encoder=False needs a policy whose net consumes `visual_features` "
                                    "(PointNavResNetPolicy; the baseline net has no such input, rl/ppo/policy.py:557-589)")
            for param in actor_critic.visual_encoder.parameters():
                param.requires_grad_(False)
        if dd.reset_critic and (dd.pretrained or dd.pretrained_encoder):
            torch.nn.init.orthogonal_(actor_critic._modules["critic"]._modules["fc"].weight)
            torch.nn.init.constant_(actor_critic._modules["critic"]._modules["fc"].bias, 0)
        actor_critic.to(self._device)
        return actor_critic

    def _create_updater(self, actor_critic):
        hb = self._config.habitat_baselines
        name = hb.distrib_updater_name if self._is_distributed else hb.updater_name
        updater_cls = baseline_registry.get_updater(name)
        if updater_cls is None:
            raise ValueError(f"Couldn't find updater {name}")
        return updater_cls.from_config(actor_critic, self._ppo_cfg)

    def _create_storage(self, num_envs, env_spec, actor_critic, policy_action_space, config, device):
        cls = baseline_registry.get_storage(config.habitat_baselines.rollout_storage_name)
        ppo = config.habitat_baselines.rl.ppo
        obs_space = self.rollout_obs_space(env_spec, actor_critic)
        st = cls(numsteps=ppo.num_steps, num_envs=num_envs, observation_space=obs_space,
                 action_space=policy_action_space, actor_critic=actor_critic, is_double_buffered=ppo.use_double_buffered_sampler)
        st.to(device)
        return st

    def rollout_obs_space(self, env_spec, actor_critic):
        """get_rollout_obs_space (single_agent_access_mgr.py:300-319): a frozen encoder's output is stored next to the sensors."""
        obs_space = env_spec.observation_space
        if self._is_static_encoder:
            from habitat_amd.common import spaces
            from habitat_amd.rl.ppo.policy import VISUAL_FEATURES_KEY
            lim = float(np.finfo(np.float32).max)
            obs_space = spaces.Dict({VISUAL_FEATURES_KEY: spaces.Box(-lim, lim, tuple(actor_critic.visual_encoder.output_shape), np.float32),
                                     **obs_space.spaces})
        return obs_space

    def post_init(self, create_rollouts_fn: Optional[Callable] = None) -> None:
        create = create_rollouts_fn or self._create_storage
        self._rollouts = create(num_envs=self._num_envs, env_spec=self._env_spec, actor_critic=self._actor_critic,
                                policy_action_space=self._actor_critic.policy_action_space, config=self._config, device=self._device)

    def init_distributed(self, find_unused_params: bool = False) -> None:
        if hasattr(self._updater, "init_distributed"):
            self._updater.init_distributed(find_unused_params=find_unused_params)

    # ---- accessors ----------------------------------------------------------------------------------------
    @property
    def masks_shape(self): return (1,)
    @property
    def policy_action_space(self): return self._actor_critic.policy_action_space
    @property
    def rollouts(self): return self._rollouts
    @property
    def actor_critic(self): return self._actor_critic
    @property
    def updater(self): return self._updater

    def train(self):
        self._actor_critic.train()
        self._updater.train()

    def eval(self):
        self._actor_critic.eval()

    # ---- state ----------------------------------------------------------------------------------------------
    def get_resume_state(self) -> Dict[str, Any]:
        # single_agent_access_mgr.py:253-263: the bare actor_critic.state_dict() (no prefix) + {"optim_state": Adam.state_dict()}
        ret = {"state_dict": {k: v.cpu() for k, v in self._actor_critic.state_dict().items()}, **self._updater.get_resume_state()}
        if self._lr_scheduler is not None:
            ret["lr_sched_state"] = self._lr_scheduler.state_dict()
        return ret

    def get_save_state(self):
        return {"state_dict": {k: v.cpu() for k, v in self._actor_critic.state_dict().items()}}

    def load_ckpt_state_dict(self, ckpt: Dict) -> None:
        self._actor_critic.load_state_dict(ckpt["state_dict"])

    def load_state_dict(self, state: Dict) -> None:
        sd = state["state_dict"]
        if any(k.startswith("actor_critic.") for k in sd):
            sd = {k[len("actor_critic."):]: v for k, v in sd.items() if k.startswith("actor_critic.")}
        self._actor_critic.load_state_dict(sd)
        if self._updater is not None:
            self._updater.load_state_dict(state)
            if "lr_sched_state" in state and self._lr_scheduler is not None:
                self._lr_scheduler.load_state_dict(state["lr_sched_state"])

    # ---- schedule hooks (:285-297) ------------------------------------------------------------------------
    def after_update(self):
        if self._ppo_cfg.use_linear_lr_decay and self._lr_scheduler is not None:
            self._lr_scheduler.step()
        self._updater.after_update()

    def pre_rollout(self):
        if self._ppo_cfg.use_linear_clip_decay:
            self._updater.clip_param = self._ppo_cfg.clip_param * (1 - self._percent_done_fn())
