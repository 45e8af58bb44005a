"""Environment source boundary.

`VectorEnvFactory.construct_envs(...) -> VectorEnv` is the reference's hook
(habitat_baselines/common/env_factory.py:14-38, selected through
``habitat_baselines.vector_env_factory._target_``).  The benchmark of this path uses synthetic
observations (BASELINE.json), so this module provides `SyntheticVectorEnvFactory` / `SyntheticVectorEnv`:
a VectorEnv-API object (habitat/core/vector_env.py:229-232,380-410,451-484: num_envs,
observation_spaces, action_spaces, orig_action_spaces, reset, async_step_at, wait_step_at, post_step,
close) whose observations are produced ON THE DEVICE by `hab_synth_step` -- bit-identical to
oracle/synth.py -- and which additionally offers `step_into(...)`, writing the next observations,
rewards and masks straight into rollout-arena rows with no host round trip.  Any other object with the
VectorEnv API (e.g. habitat's own process-per-env VectorEnv) can be returned by a user factory; the
trainer then uses the generic host path."""
from __future__ import annotations

import abc
import importlib
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from habitat_amd import _lib
from habitat_amd._lib import check, ptr, stream_ptr
from habitat_amd.common import spaces

GOAL_UUID = "pointgoal_with_gps_compass"
NUM_SEMANTIC_IDS, NUM_OBJECT_CATEGORIES = 40, 21  # synthetic ObjectNav sensor ranges


class VectorEnvFactory(abc.ABC):
    @abc.abstractmethod
    def construct_envs(self, config, workers_ignore_signals: bool = False, enforce_scenes_greater_eq_environments: bool = False,
                       is_first_rank: bool = True):
        ...


def instantiate(target_cfg):
    """hydra.utils.instantiate for a `{_target_: 'pkg.mod.Class', **kwargs}` node."""
    target = target_cfg["_target_"]
    mod, _, name = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), name)
    return cls(**{k: v for k, v in target_cfg.items() if k != "_target_"})


class SyntheticVectorEnv:
    def __init__(self, num_envs: int, height: int, width: int, seed: int = 100, env_offset: int = 0, use_rgb: bool = True,
                 use_depth: bool = True, num_actions: int = 4, device="cuda", task: str = "pointnav"):
        self.num_envs, self.H, self.W = num_envs, height, width
        self.seed, self.env_offset = int(seed) & 0xFFFFFFFF, int(env_offset)
        self.use_rgb, self.use_depth, self.task = use_rgb, use_depth, task
        self.device = torch.device(device)
        d = {}
        if use_rgb:
            d["rgb"] = spaces.Box(0, 255, (height, width, 3), np.uint8)
        if use_depth:
            d["depth"] = spaces.Box(0.0, 1.0, (height, width, 1), np.float32)
        fmin, fmax = np.finfo(np.float32).min, np.finfo(np.float32).max
        if task == "objectnav":  # sensor set of the ObjectNav experiments (rgb, depth, semantic + objectgoal, compass, gps)
            d["semantic"] = spaces.Box(0, NUM_SEMANTIC_IDS - 1, (height, width, 1), np.int32)
            d["objectgoal"] = spaces.Box(0, NUM_OBJECT_CATEGORIES - 1, (1,), np.int64)
            d["compass"] = spaces.Box(-np.pi, np.pi, (1,), np.float32)
            d["gps"] = spaces.Box(fmin, fmax, (2,), np.float32)
        else:
            d[GOAL_UUID] = spaces.Box(fmin, fmax, (2,), np.float32)
        self.observation_spaces = [spaces.Dict(d) for _ in range(num_envs)]
        self.action_spaces = [spaces.Discrete(num_actions) for _ in range(num_envs)]
        self.orig_action_spaces = self.action_spaces
        self.number_of_episodes = [None] * num_envs
        if self.device.type != "cuda":
            raise _lib.HabError("SyntheticVectorEnv generates observations on the GPU; no GPU device given")
        dev = self.device
        self._t = torch.zeros(num_envs, dtype=torch.int64, device=dev)
        self._since = torch.zeros(num_envs, dtype=torch.int64, device=dev)
        self._rgb = torch.zeros(num_envs, height, width, 3, dtype=torch.uint8, device=dev) if use_rgb else None
        self._depth = torch.zeros(num_envs, height, width, 1, device=dev) if use_depth else None
        self._goal = torch.zeros(num_envs, 2, device=dev) if task != "objectnav" else None
        self._obj = {}
        if task == "objectnav":
            self._obj = dict(semantic=torch.zeros(num_envs, height, width, 1, dtype=torch.int32, device=dev),
                             objectgoal=torch.zeros(num_envs, 1, dtype=torch.int64, device=dev),
                             compass=torch.zeros(num_envs, 1, device=dev), gps=torch.zeros(num_envs, 2, device=dev))
        self._rew = torch.zeros(num_envs, device=dev)
        self._nd = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self._pending: set = set()      # envs with an outstanding async_step_at (host path)
        self._host_cache = None
        self._tmp = None

    # ---- device fast path ---------------------------------------------------------------------
    def _emit(self, rgb, depth, goal, reward, not_done, advance: int):
        check(_lib.lib().hab_synth_step(ptr(rgb), ptr(depth), ptr(goal), ptr(reward), ptr(not_done), ptr(self._t), ptr(self._since),
                                        self.seed, self.env_offset, self.num_envs, self.H, self.W, advance, stream_ptr()),
              "hab_synth_step")

    def _emit_objectnav(self, obs):
        if self.task != "objectnav":
            return
        check(_lib.lib().hab_synth_objectnav_sensors(ptr(obs.get("semantic")), ptr(obs.get("objectgoal")), ptr(obs.get("compass")),
                                                     ptr(obs.get("gps")), ptr(self._t), self.seed, self.env_offset, self.num_envs,
                                                     self.H, self.W, stream_ptr()), "hab_synth_objectnav_sensors")

    def reset_into(self, rgb, depth, goal):
        """Resets all envs and writes their first observations into the given (N, ...) device tensors."""
        self._t.zero_()
        self._since.zero_()
        self._emit(rgb, depth, goal, None, None, 0)

    def step_into(self, rgb, depth, goal, reward, not_done):
        """Advances all envs one step (actions do not influence synthetic observations) and writes the new
        observations / rewards (N,) / not-done masks (N,) bytes straight into the given device tensors."""
        self._emit(rgb, depth, goal, reward, not_done, 1)

    def reset_into_obs(self, obs):
        """Dict form: `obs` maps sensor uuid -> (N, ...) device tensor (a rollout-arena row)."""
        self.reset_into(obs.get("rgb"), obs.get("depth"), obs.get(GOAL_UUID))
        self._emit_objectnav(obs)

    def step_into_obs(self, obs, reward, not_done):
        self.step_into(obs.get("rgb"), obs.get("depth"), obs.get(GOAL_UUID), reward, not_done)
        self._emit_objectnav(obs)

    # ---- VectorEnv API (host path) ----------------------------------------------------------------
    def _host_obs(self) -> List[Dict[str, np.ndarray]]:
        rgb = self._rgb.cpu().numpy() if self.use_rgb else None
        depth = self._depth.cpu().numpy() if self.use_depth else None
        goal = self._goal.cpu().numpy() if self._goal is not None else None
        extra = {k: v.cpu().numpy() for k, v in self._obj.items()}
        out = []
        for i in range(self.num_envs):
            o = {}
            if self.use_rgb:
                o["rgb"] = rgb[i]
            if self.use_depth:
                o["depth"] = depth[i]
            for k, v in extra.items():
                o[k] = v[i]
            if goal is not None:
                o[GOAL_UUID] = goal[i]
            out.append(o)
        return out

    def _own_obs(self):
        o = dict(self._obj)
        if self.use_rgb:
            o["rgb"] = self._rgb
        if self.use_depth:
            o["depth"] = self._depth
        if self._goal is not None:
            o[GOAL_UUID] = self._goal
        return o

    def reset(self):
        self.reset_into_obs(self._own_obs())
        return self._host_obs()

    def async_step_at(self, index_env: int, action) -> None:
        if index_env in self._pending:
            raise _lib.HabError(f"env {index_env}: async_step_at called twice without wait_step_at")
        self._pending.add(int(index_env))

    def advance_on_device(self) -> List[int]:
        """Advances exactly the envs whose step was requested, entirely on the device (no host copy of any observation): their new
        observations / rewards / not-done bytes are in the env's own tensors afterwards.  All envs (the usual case): one generator
        launch into the env's own tensors.  A subset (VER batches, the double-buffered sampler's halves, ppo_trainer.py:743-768): the
        generator runs on scratch copies of the per-env clocks / tensors and only the requested rows are taken over.  Returns the ids."""
        ids = sorted(self._pending)
        self._pending.clear()
        if len(ids) == self.num_envs:
            self.step_into_obs(self._own_obs(), self._rew, self._nd)
        elif ids:
            own = self._own_obs()
            if self._tmp is None:
                self._tmp = ({k: torch.empty_like(v) for k, v in own.items()}, torch.empty_like(self._rew), torch.empty_like(self._nd))
            tobs, trew, tnd = self._tmp
            t0, s0 = self._t.clone(), self._since.clone()
            self.step_into_obs(tobs, trew, tnd)
            sel = torch.tensor(ids, device=self.device)
            keep = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
            keep[sel] = False
            self._t[keep], self._since[keep] = t0[keep], s0[keep]
            for k, v in own.items():
                v[sel] = tobs[k][sel]
            self._rew[sel], self._nd[sel] = trew[sel], tnd[sel]
        return ids

    def step_results_host(self):
        """(rewards (N,), not-done (N,)) of the last steps as host arrays: the two small per-env vectors the host-side episode
        accounting needs (one device->host copy each; observations stay in HBM)."""
        return self._rew.cpu().numpy(), self._nd.cpu().numpy()

    def _advance_pending(self):
        """Host path (VectorEnv.wait_step_at): device advance + the host copies of the observations the caller asked for."""
        ids = self.advance_on_device()
        obs = self._host_obs()
        rew, nd = self.step_results_host()
        if self._host_cache is None or len(ids) == self.num_envs:
            self._host_cache = (obs, rew.copy(), nd.copy())
        else:
            c_obs, c_rew, c_nd = self._host_cache
            for i in ids:
                c_obs[i], c_rew[i], c_nd[i] = obs[i], rew[i], nd[i]

    def wait_step_at(self, index_env: int):
        if index_env in self._pending:  # every env requested so far advances on the first wait
            self._advance_pending()
        obs, rew, nd = self._host_cache
        return obs[index_env], float(rew[index_env]), not bool(nd[index_env]), {}

    def post_step(self, observations):
        return observations

    def close(self):
        pass


class SyntheticVectorEnvFactory(VectorEnvFactory):
    """Default `_target_`: N synthetic PointNav envs sized from habitat.simulator.sensors.*; per-rank env ids are
    offset by rank * num_environments exactly like the reference offsets the seed (ppo_trainer.py:208-211)."""

    def __init__(self, use_rgb: bool = True, use_depth: bool = True):
        self.use_rgb, self.use_depth = use_rgb, use_depth

    def construct_envs(self, config, workers_ignore_signals: bool = False, enforce_scenes_greater_eq_environments: bool = False,
                       is_first_rank: bool = True, device="cuda", env_offset: int = 0):
        hb, hab = config.habitat_baselines, config.habitat
        sens = hab.simulator.sensors
        use_rgb = self.use_rgb and "rgb" in sens
        use_depth = self.use_depth and "depth" in sens
        ref = sens["rgb"] if use_rgb else sens["depth"]
        task = "objectnav" if str(hab.task.type).lower().startswith("objectnav") else "pointnav"
        return SyntheticVectorEnv(int(hb.num_environments), int(ref.height), int(ref.width), seed=int(hab.seed),
                                  env_offset=env_offset, use_rgb=use_rgb, use_depth=use_depth,
                                  num_actions=len(hab.task.actions), device=device, task=task)


class ProcessVectorEnvFactory(VectorEnvFactory):
    """`_target_: habitat_amd.common.env_factory.ProcessVectorEnvFactory`: one worker PROCESS per environment behind the
    reference's VectorEnv API (core/vector_env.py), observations through the shared-memory plane.  The worker env is
    `core.host_env.HostSyntheticNavEnv` unless `make_env_fn` names another constructor ('pkg.mod.fn'); with habitat-sim the
    reference's `make_gym_from_config` goes here (habitat_baselines/common/habitat_env_factory.py:80-119)."""

    def __init__(self, make_env_fn: Optional[str] = None, shared_obs: bool = True, start_method: str = "forkserver", work_us: int = 0):
        self.make_env_fn, self.shared_obs, self.start_method, self.work_us = make_env_fn, shared_obs, start_method, work_us

    def construct_envs(self, config, workers_ignore_signals: bool = False, enforce_scenes_greater_eq_environments: bool = False,
                       is_first_rank: bool = True, env_offset: int = 0):
        from habitat_amd.core.host_env import make_host_env
        from habitat_amd.core.vector_env import VectorEnv
        hb, hab = config.habitat_baselines, config.habitat
        sens = hab.simulator.sensors
        use_rgb, use_depth = "rgb" in sens, "depth" in sens
        ref = sens["rgb"] if use_rgb else sens["depth"]
        fn = make_host_env
        if self.make_env_fn:
            mod, _, name = self.make_env_fn.rpartition(".")
            fn = getattr(importlib.import_module(mod), name)
        n = int(hb.num_environments)
        args = [(int(hab.seed) + env_offset + i, int(ref.height), int(ref.width), use_rgb, use_depth, len(hab.task.actions),
                 int(hab.environment.max_episode_steps), self.work_us) for i in range(n)]
        return VectorEnv(fn, args, auto_reset_done=True, multiprocessing_start_method=self.start_method,
                         workers_ignore_signals=workers_ignore_signals, shared_obs=self.shared_obs)
