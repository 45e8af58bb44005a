"""Environment-side transports for the VER inference worker (the reference's transfer buffers + per-environment task queues,
ver_trainer.py:279-296, environment_worker.py:148-203): who steps the environments and where their results land."""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from habitat_amd.rl.ver.inference_worker import EnvironmentTransport


class _Records(EnvironmentTransport):
    def __init__(self, num_envs: int, report=None):
        self.num_envs = num_envs
        self.rewards = np.zeros(num_envs, np.float32)
        self.masks = np.zeros(num_envs, bool)
        self.episode_ids = np.zeros(num_envs, np.int64)
        self.step_ids = np.zeros(num_envs, np.int64)
        self._ep_reward = np.zeros(num_envs, np.float64)
        self._ep_len = np.zeros(num_envs, np.int64)
        self.report = report

    def _record(self, e: int, reward: float, done: bool, info) -> None:
        """environment_worker.py:163-215: ids advance with the step, an episode end bumps the episode id and reports."""
        if not np.isfinite(reward):
            reward = -1.0
        self.step_ids[e] += 1
        if done:
            self.episode_ids[e] += 1
            self.step_ids[e] = 0
        self.rewards[e], self.masks[e] = reward, not done
        self._ep_reward[e] += reward
        self._ep_len[e] += 1
        if done:
            if self.report is not None:
                self.report.episode_end(dict(env_idx=e, length=int(self._ep_len[e]), reward=float(self._ep_reward[e]), info=info or {}))
            self._ep_reward[e], self._ep_len[e] = 0.0, 0


class VectorEnvTransport(_Records):
    """Process-per-environment workers (core/vector_env.py): results arrive through the pipes whenever an environment finishes,
    observations sit in the shared-memory slabs."""

    def __init__(self, envs, report=None):
        super().__init__(envs.num_envs, report)
        self.envs = envs
        self._fresh: List[int] = []

    def start_experience_collection(self) -> List[int]:
        self.envs.reset()  # first observations land in the slabs; masks False, ids 0
        return list(range(self.num_envs))

    def observations(self, env_ids, device):
        return self.envs.gather_obs(env_ids, device)

    def send_action(self, env_idx, action):
        a = np.asarray(action)
        self.envs.async_step_at(env_idx, a.item() if a.size == 1 else a)

    def poll(self, timeout, max_messages):
        out = []
        for i, (_obs, reward, done, info) in self.envs.poll_steps(timeout, max_messages):
            self._record(i, float(reward), bool(done), info)
            out.append(i)
        return out


class DeviceEnvTransport(_Records):
    """The device-resident synthetic environment source (common/env_factory.py::SyntheticVectorEnv): observations never leave HBM.
    Environments finish their steps at different simulated rates (`speeds`: probability per poll that an outstanding step has
    arrived), which is what produces variable experience; speeds of 1 make every step arrive at the next poll."""

    def __init__(self, envs, report=None, speeds: Optional[np.ndarray] = None, seed: int = 0):
        super().__init__(envs.num_envs, report)
        self.envs = envs
        self._outstanding = np.zeros(self.num_envs, bool)
        self._speeds = np.ones(self.num_envs) if speeds is None else np.asarray(speeds, dtype=np.float64)
        self._rng = np.random.RandomState(seed)

    def start_experience_collection(self) -> List[int]:
        self.envs.reset_into_obs(self.envs._own_obs())
        return list(range(self.num_envs))

    def observations(self, env_ids, device):
        idx = torch.as_tensor(env_ids, device=self.envs.device)
        return {k: v.index_select(0, idx) for k, v in self.envs._own_obs().items()}

    def send_action(self, env_idx, action):
        self._outstanding[env_idx] = True

    def poll(self, timeout, max_messages):
        cand = np.nonzero(self._outstanding)[0]
        if len(cand) == 0:
            return []
        arrived = [int(e) for e in cand if self._rng.random_sample() < self._speeds[e]][:max_messages]
        if not arrived:
            return []
        for e in arrived:
            self.envs.async_step_at(e, 0)
        self.envs.advance_on_device()               # observations are generated in HBM and stay there
        rew, nd = self.envs.step_results_host()     # two N-element vectors for the host-side episode accounting
        for e in arrived:
            self._outstanding[e] = False
            self._record(e, float(rew[e]), not bool(nd[e]), {})
        return arrived
