"""VER inference worker (habitat_baselines/rl/ver/inference_worker.py:52-520): batches the environments whose step has arrived,
runs ONE policy forward for them on the HIP engine, hands the actions back, and writes the step into the VER arena.

This implementation runs in the trainer's process (the reference's `main_is_iw` arrangement, ver_trainer.py:269-337, i.e.
`rl.ver.overlap_rollouts_and_learn=False`, one inference worker): the policy lives in device memory once, the arena is written by
device-side index copies, and the only host<->device traffic per batch is the observation upload (from the environment workers'
shared-memory slabs) and the sampled actions coming back.

`EnvironmentTransport` is what the worker needs from the environment side: per-environment transfer records (reward, not-done
mask, episode id, step id written by the environment after each step: environment_worker.py:186-203), the observations of a list of
environments as device tensors, and a way to send an action.  `core.vector_env.VectorEnv` provides it through `VectorEnvTransport`;
tests use an in-process transport."""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from habitat_amd.common.obs_transformers import apply_obs_transforms_batch
from habitat_amd.common.windowed_running_mean import WindowedRunningMean
from habitat_amd.rl.ppo.policy import VISUAL_FEATURES_KEY
from habitat_amd.rl.ver.ver_rollout_storage import VERRolloutStorage


class EnvironmentTransport:
    """Transfer buffers + per-environment task queues of the reference (ver_trainer.py:279-296, worker_common.py:24-38)."""
    num_envs: int
    rewards: np.ndarray       # (N,) float32: reward earned by the environment's previous action
    masks: np.ndarray         # (N,) bool: False at the first observation of an episode
    episode_ids: np.ndarray   # (N,) int64
    step_ids: np.ndarray      # (N,) int64

    def observations(self, env_ids: List[int], device) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def send_action(self, env_idx: int, action) -> None:
        """EnvironmentWorkerTasks.step for environment env_idx with this action."""
        raise NotImplementedError

    def poll(self, timeout: float, max_messages: int) -> List[int]:
        """Environments whose step has arrived since the last call (queues.inference.get_many)."""
        raise NotImplementedError


class InferenceWorker:
    def __init__(self, config, actor_critic, rollouts: VERRolloutStorage, transport: EnvironmentTransport, device, obs_transforms=(),
                 num_inference_workers: int = 1, report=None):
        self.config, self.actor_critic, self.rollouts, self.transport = config, actor_critic, rollouts, transport
        self.device = torch.device(device)
        self.obs_transforms = list(obs_transforms)
        self.report = report
        self.new_reqs: List[int] = []
        self.replay_reqs: List[int] = []
        self._n_replay_steps = 0
        self._variable_experience = bool(rollouts.variable_experience)
        self._static_encoder = not config.habitat_baselines.rl.ddppo.train_encoder
        n = transport.num_envs
        # request batching thresholds (inference_worker.py:108-127)
        self.min_reqs = int(max(n / num_inference_workers / 1.5, 1))
        self.max_reqs = int(max(n / num_inference_workers * 1.5, 1))
        self.min_wait_time = 0.01
        self.last_step_time = time.perf_counter()
        self._avg_step_time = WindowedRunningMean(128)
        self._current_policy_version = int(rollouts.cpu_current_policy_version[0, 0])

    # ---- one batch of requests (inference_worker.py:238-420) ---------------------------------------------------------------------
    @torch.no_grad()
    def step(self, exp_noise: Optional[torch.Tensor] = None) -> Tuple[bool, List[Tuple[int, int]]]:
        steps_finished: List[Tuple[int, int]] = []
        if not self.new_reqs:
            return False, steps_finished
        ro, tr = self.rollouts, self.transport
        self._current_policy_version = int(ro.cpu_current_policy_version[0, 0])
        current_steps = ro.current_steps.copy()
        final_batch = False
        slots = None
        if self._variable_experience:
            # environments that have contributed the fewest steps go first: they are the ones cut off when the rollout fills up
            self.new_reqs.sort(key=lambda a: (ro.actor_steps_collected[a], a))
            slots, n_proc, final_batch = ro.reserve_slots(len(self.new_reqs), self._n_replay_steps)
            self.replay_reqs += self.new_reqs[n_proc:]
            self.new_reqs = self.new_reqs[:n_proc]
        else:
            for r in self.new_reqs:
                if current_steps[r] > ro.num_steps:
                    raise RuntimeError(f"Got a step from actor {r} after collecting {current_steps[r]} steps. This shouldn't be possible.")
            ro.num_steps_collected += len(self.new_reqs) - self._n_replay_steps
            if ro.num_steps_collected[0] == ro.num_steps_to_collect:
                final_batch = True
                ro.rollout_done[:] = True
        if not self.new_reqs:
            return False, steps_finished
        if self._n_replay_steps > 0 and self.replay_reqs:
            raise RuntimeError(f"Added to replay reqs before reqs from the last rollout were replayed. {self.replay_reqs}")
        reqs = list(self.new_reqs)
        dev = self.device
        env_ids = torch.as_tensor(reqs, dtype=torch.int64, device=dev)
        obs = apply_obs_transforms_batch(tr.observations(reqs, dev), self.obs_transforms)
        masks = torch.from_numpy(np.ascontiguousarray(tr.masks[reqs])).view(-1, 1).to(dev)
        rewards = torch.from_numpy(np.ascontiguousarray(tr.rewards[reqs], dtype=np.float32)).view(-1, 1).to(dev)
        episode_ids = torch.from_numpy(np.ascontiguousarray(tr.episode_ids[reqs])).view(-1, 1).to(dev)
        step_ids = torch.from_numpy(np.ascontiguousarray(tr.step_ids[reqs])).view(-1, 1).to(dev)
        hidden = ro.next_hidden_states[env_ids]
        prev_actions = ro.next_prev_actions[env_ids]
        if self._static_encoder:
            obs[VISUAL_FEATURES_KEY] = self.actor_critic.encode_visual(obs)
        action_data = self.actor_critic.act(obs, hidden, prev_actions, masks, exp_noise=exp_noise)
        if not final_batch:
            ro.next_hidden_states.index_copy_(0, env_ids, action_data.rnn_hidden_states)
            ro.next_prev_actions.index_copy_(0, env_ids, action_data.actions)
        if self._variable_experience:
            ro.remember_slots(reqs, slots)
        cpu_actions = action_data.env_actions.cpu().numpy()  # the one device->host read of the batch
        for i, env_idx in enumerate(reqs):
            steps_finished.append((int(ro.current_steps[env_idx]), int(env_idx)))
            ro.actor_steps_collected[env_idx] += 1
            ro.current_steps[env_idx] += 1
            final_step = final_batch if self._variable_experience else ro.current_steps[env_idx] == ro.num_steps + 1
            if not final_step:
                tr.send_action(env_idx, cpu_actions[i])
            else:
                # the last batch of a rollout is REPLAYED at the start of the next one: the (new) policy recomputes its value
                # estimate, which bootstraps the returns (inference_worker.py:375-381)
                self.replay_reqs.append(env_idx)
        n = len(reqs)
        current_step = dict(masks=masks, observations=obs, actions=action_data.actions, action_log_probs=action_data.action_log_probs,
                            recurrent_hidden_states=hidden, prev_actions=prev_actions,
                            policy_version=ro.current_policy_version.expand(n, 1), episode_ids=episode_ids,
                            environment_ids=env_ids.view(-1, 1), step_ids=step_ids, value_preds=action_data.values,
                            returns=torch.full((n, 1), float("nan"), device=dev))
        ro.write_step(reqs, slots, current_step, rewards, current_steps)
        self.new_reqs = []
        return True, steps_finished

    # ---- end of a rollout (inference_worker.py:422-456, single worker) -----------------------------------------------------------------
    def finish_rollout(self) -> None:
        """Requests that arrived but were not processed, and the requests of the final batch, are replayed first in the next
        rollout; environments with an action still in flight are not (their step lands in the next rollout)."""
        self.new_reqs = self.replay_reqs + self.new_reqs
        self.replay_reqs = []
        self._n_replay_steps = len(self.new_reqs)
        self.rollouts.will_replay_step[self.new_reqs] = True

    # ---- request batching (inference_worker.py:458-505) ------------------------------------------------------------------------------
    def try_one_step(self) -> bool:
        if len(self.new_reqs) < self.max_reqs:
            self.new_reqs += self.transport.poll(0.005, self.max_reqs - len(self.new_reqs))
        should = len(self.new_reqs) > 0 and (len(self.new_reqs) >= self.min_reqs
                                            or (time.perf_counter() - self.last_step_time) > self.min_wait_time)
        if not should:
            return False
        t0 = time.perf_counter()
        stepped, steps_finished = self.step()
        t1 = time.perf_counter()
        if stepped:
            self._avg_step_time.add(t1 - t0)
            self.last_step_time = t1
            self.min_wait_time = self._avg_step_time.mean / 2
            self._n_replay_steps = 0
            if self.report is not None:
                self.report.policy_step(steps_finished, t1)
        return stepped
