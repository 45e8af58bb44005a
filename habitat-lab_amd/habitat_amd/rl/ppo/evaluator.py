"""Checkpoint evaluation loop: `Evaluator` / `pause_envs` (habitat_baselines/rl/ppo/evaluator.py:17-105) and
`HabitatEvaluator.evaluate_agent` (habitat_baselines/rl/ppo/habitat_evaluator.py:39-339) for the engine-backed policies.

Same control flow as the reference: reset, then act -> step all (un-paused) envs -> on every episode end record
`{"reward": episode return, **scalar infos}` under (scene_id, episode_id, eval count) -> pause an env once its NEXT episode has
already been evaluated `evals_per_ep` times -> stop after `test_episode_count * evals_per_ep` episodes; finally average every
statistic over the episodes and write `eval_reward/average_reward`, `eval_metrics/<k>` to the writer.  Sampling follows the
reference (`deterministic=False`, :134-141).  Video / gfx-replay output is simulator-side and not reproduced: a non-empty
`eval.video_option` is refused."""
from __future__ import annotations

import abc
from collections import defaultdict
from typing import Any, Dict, List

import numpy as np
import torch

from habitat_amd import _lib
from habitat_amd.common.obs_transformers import apply_obs_transforms_batch
from habitat_amd.common.spaces import get_action_space_info
from habitat_amd.utils.logging import logger


def extract_scalars_from_info(info: Dict[str, Any], prefix: str = "") -> Dict[str, float]:
    """habitat_baselines/utils/info_dict.py:31-63: numeric leaves, nested dicts flattened with '.'."""
    out = {}
    for k, v in (info or {}).items():
        if isinstance(v, dict):
            out.update(extract_scalars_from_info(v, prefix + str(k) + "."))
        elif isinstance(v, (bool, int, float, np.integer, np.floating)) or (isinstance(v, np.ndarray) and v.size == 1):
            out[prefix + str(k)] = float(v)
    return out


def _episode_key(ep) -> tuple:
    get = (lambda k: ep[k]) if isinstance(ep, dict) else (lambda k: getattr(ep, k))
    return (get("scene_id"), get("episode_id"))


def pause_envs(envs_to_pause: List[int], envs, test_recurrent_hidden_states, not_done_masks, current_episode_reward, prev_actions,
               batch, rgb_frames=None):
    """evaluator.py:57-105: drop the paused envs from the vector env and from every per-env tensor."""
    if len(envs_to_pause) > 0:
        state_index = list(range(envs.num_envs))
        for idx in reversed(envs_to_pause):
            state_index.pop(idx)
            envs.pause_at(idx)
        test_recurrent_hidden_states = test_recurrent_hidden_states[state_index]
        not_done_masks = not_done_masks[state_index]
        current_episode_reward = current_episode_reward[state_index]
        prev_actions = prev_actions[state_index]
        for k, v in batch.items():
            batch[k] = v[state_index]
        if rgb_frames is not None:
            rgb_frames = [rgb_frames[i] for i in state_index]
    return envs, test_recurrent_hidden_states, not_done_masks, current_episode_reward, prev_actions, batch, rgb_frames


class Evaluator(abc.ABC):
    @abc.abstractmethod
    def evaluate_agent(self, agent, envs, config, checkpoint_index, step_id, writer, device, obs_transforms, env_spec, rank0_keys):
        pass


class HabitatEvaluator(Evaluator):
    def evaluate_agent(self, agent, envs, config, checkpoint_index, step_id, writer, device, obs_transforms, env_spec, rank0_keys):
        from habitat_amd.rl.ppo.ppo_trainer import batch_obs
        hb = config.habitat_baselines
        if len(hb.eval.video_option) > 0:
            raise _lib.HabError("eval.video_option: video generation needs the simulator's renderer and is not part of this path")
        observations = envs.post_step(envs.reset())
        batch = apply_obs_transforms_batch(batch_obs(observations, device), obs_transforms)
        ac = agent.actor_critic
        action_shape, discrete_actions = get_action_space_info(ac.policy_action_space)
        n0 = envs.num_envs
        current_episode_reward = torch.zeros(n0, 1, device="cpu")
        test_recurrent_hidden_states = torch.zeros((n0, *ac.hidden_state_shape), device=device)
        prev_actions = torch.zeros(n0, *action_shape, device=device, dtype=torch.long if discrete_actions else torch.float)
        not_done_masks = torch.zeros(n0, *agent.masks_shape, device=device, dtype=torch.bool)
        stats_episodes: Dict[Any, Any] = {}
        ep_eval_count: Dict[Any, int] = defaultdict(lambda: 0)
        number_of_eval_episodes = hb.test_episode_count
        evals_per_ep = hb.eval.evals_per_ep
        if number_of_eval_episodes == -1:
            number_of_eval_episodes = sum(envs.number_of_episodes)
        else:
            total_num_eps = sum(envs.number_of_episodes)
            if total_num_eps < number_of_eval_episodes and total_num_eps > 1:
                logger.warn(f"Config specified {number_of_eval_episodes} eval episodes, dataset only has {total_num_eps}.")
                number_of_eval_episodes = total_num_eps
            else:
                assert evals_per_ep == 1
        assert number_of_eval_episodes > 0, "You must specify a number of evaluation episodes with test_episode_count"
        agent.eval()
        while len(stats_episodes) < (number_of_eval_episodes * evals_per_ep) and envs.num_envs > 0:
            current_episodes_info = envs.current_episodes()
            with torch.no_grad():
                action_data = ac.act({k: v.contiguous() for k, v in batch.items()}, test_recurrent_hidden_states.contiguous(),
                                     prev_actions, not_done_masks, deterministic=False)
                if action_data.should_inserts is None:
                    test_recurrent_hidden_states = action_data.rnn_hidden_states
                    prev_actions.copy_(action_data.actions)
                else:
                    ac.update_hidden_state(test_recurrent_hidden_states, prev_actions, action_data)
            step_data = [a.item() for a in action_data.env_actions.cpu()]  # host copies: workers must never see device tensors
            outputs = envs.step(step_data)
            observations, rewards_l, dones, infos = [list(x) for x in zip(*outputs)]
            policy_infos = ac.get_extra(action_data, infos, dones)
            for i in range(len(policy_infos)):
                infos[i].update(policy_infos[i])
            observations = envs.post_step(observations)
            batch = apply_obs_transforms_batch(batch_obs(observations, device), obs_transforms)
            not_done_masks = torch.tensor([[not done] for done in dones], dtype=torch.bool, device="cpu").repeat(1, *agent.masks_shape)
            current_episode_reward += torch.tensor(rewards_l, dtype=torch.float, device="cpu").unsqueeze(1)
            next_episodes_info = envs.current_episodes()
            envs_to_pause = []
            for i in range(envs.num_envs):
                if ep_eval_count[_episode_key(next_episodes_info[i])] == evals_per_ep:
                    envs_to_pause.append(i)
                if not not_done_masks[i].any().item():  # episode ended
                    episode_stats = {"reward": current_episode_reward[i].item()}
                    episode_stats.update(extract_scalars_from_info({k: v for k, v in infos[i].items() if k not in rank0_keys}))
                    current_episode_reward[i] = 0
                    k = _episode_key(current_episodes_info[i])
                    ep_eval_count[k] += 1
                    stats_episodes[(k, ep_eval_count[k])] = episode_stats
            not_done_masks = not_done_masks.to(device=device)
            (envs, test_recurrent_hidden_states, not_done_masks, current_episode_reward, prev_actions, batch, _) = pause_envs(
                envs_to_pause, envs, test_recurrent_hidden_states, not_done_masks, current_episode_reward, prev_actions, batch)
            if any(envs_to_pause):
                ac.on_envs_pause(envs_to_pause)
        assert len(ep_eval_count) >= number_of_eval_episodes, f"Expected {number_of_eval_episodes} episodes, got {len(ep_eval_count)}."
        aggregated_stats = {}
        all_ks = set()
        for ep in stats_episodes.values():
            all_ks.update(ep.keys())
        for stat_key in all_ks:
            aggregated_stats[stat_key] = float(np.mean([v[stat_key] for v in stats_episodes.values() if stat_key in v]))
        for k, v in aggregated_stats.items():
            logger.info(f"Average episode {k}: {v:.4f}")
        writer.add_scalar("eval_reward/average_reward", aggregated_stats["reward"], step_id)
        for k, v in aggregated_stats.items():
            if k != "reward":
                writer.add_scalar(f"eval_metrics/{k}", v, step_id)
        self.last_stats_episodes, self.last_aggregated_stats = stats_episodes, aggregated_stats
        return aggregated_stats
