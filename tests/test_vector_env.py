"""CPU tests of the process-per-env VectorEnv with the shared-memory observation plane.  Mirrors what the
reference checks for its VectorEnv in test/test_habitat_env.py (test_vectorized_envs, test_with_scope, pause/resume): the vector
env must behave exactly like the same envs stepped in-process, here additionally through the slab transport."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))

from habitat_amd.core.host_env import GOAL_UUID, HostSyntheticNavEnv, make_host_env  # noqa: E402
from habitat_amd.core.vector_env import VectorEnv  # noqa: E402

H, W = 12, 20


def _args(n, max_steps=7):
    return [(100 + i, H, W, True, True, 4, max_steps, 0) for i in range(n)]


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


@pytest.mark.parametrize("shared", [True, False])
def test_vector_env_matches_inprocess_envs(shared):
    n = 3
    local = [make_host_env(*a) for a in _args(n)]
    with VectorEnv(make_host_env, _args(n), shared_obs=shared, multiprocessing_start_method="forkserver") as envs:
        assert envs.num_envs == n and len(envs.observation_spaces) == n
        assert envs.action_spaces[0].n == 4 and envs.orig_action_spaces[0].n == 4
        assert set(envs.shared_obs_keys) == ({"rgb", "depth", GOAL_UUID} if shared else set())
        for o, e in zip(envs.reset(), local):
            _same(o, e.reset())
        rng = np.random.default_rng(0)
        n_done = 0
        for step in range(25):
            actions = rng.integers(0, 4, n)
            outs = envs.step(list(actions))
            for i, (obs, reward, done, info) in enumerate(outs):
                lo, lr, ld, li = local[i].step(int(actions[i]))
                if ld:
                    lo = local[i].reset()  # auto_reset_done
                    n_done += 1
                _same(obs, lo)
                assert reward == lr and done == ld and info == li
        assert n_done > 0  # the 7-step budget guarantees episode ends -> auto reset exercised
        assert envs.call_at(1, "get_metrics") == local[1].get_metrics()
        assert envs.episode_over() == [e.episode_over for e in local]
        assert envs.current_episodes() == [e.current_episode for e in local]
        assert envs.count_episodes() == [0] * n


def test_async_step_order_and_batched_obs():
    n = 4
    local = [make_host_env(*a) for a in _args(n, 500)]
    with VectorEnv(make_host_env, _args(n, 500)) as envs:
        envs.reset()
        for e in local:
            e.reset()
        # the double-buffered sampler steps the two halves out of phase (ppo_trainer.py:743-768)
        for i in (2, 3):
            envs.async_step_at(i, 1)
        for i in (0, 1):
            envs.async_step_at(i, 2)
        res = {i: envs.wait_step_at(i) for i in (0, 1, 2, 3)}
        for i in range(n):
            _same(res[i][0], local[i].step(2 if i < 2 else 1)[0])
        b = envs.batched_obs(slice(2, 4), "cpu")
        assert b["rgb"].shape == (2, H, W, 3) and b["rgb"].dtype == torch.uint8 and b["depth"].dtype == torch.float32
        for j, i in enumerate((2, 3)):
            for k in ("rgb", "depth", GOAL_UUID):
                assert np.array_equal(b[k][j].numpy(), res[i][0][k])
        # batched copies are snapshots: a later step of the same env must not change them
        snap = b["rgb"].clone()
        envs.step_at(2, 0)
        assert torch.equal(b["rgb"], snap)


def test_pause_resume_and_errors():
    n = 3
    with VectorEnv(make_host_env, _args(n, 500)) as envs:
        envs.reset()
        envs.pause_at(1)
        assert envs.num_envs == 2
        outs = envs.step([0, 0])
        ref0, ref2 = HostSyntheticNavEnv(100, H, W), HostSyntheticNavEnv(102, H, W)
        ref0.reset(), ref2.reset()
        _same(outs[0][0], ref0.step(0)[0])
        _same(outs[1][0], ref2.step(0)[0])  # index 1 is now the third env and reads ITS slab row
        envs.resume_all()
        assert envs.num_envs == 3
        assert len(envs.step([0, 0, 0])) == 3
    with pytest.raises(AssertionError):
        VectorEnv(make_host_env, [])


def test_trainer_host_path_with_process_envs_builds_batches():
    """The trainer-side fast path: slab rows -> batched tensors equal to stacking the per-env observation dicts."""
    n = 3
    with VectorEnv(make_host_env, _args(n, 500)) as envs:
        envs.reset()
        outs = envs.step([1, 2, 3])
        obs = [o[0] for o in outs]
        batch = envs.batched_obs(slice(0, n), torch.device("cpu"))
        for k in obs[0]:
            assert np.array_equal(batch[k].numpy(), np.stack([o[k] for o in obs]))


def test_evaluator_loop_over_process_envs():
    """HabitatEvaluator (habitat_evaluator.py:39-339) on CPU with a stand-in agent: every recorded episode must carry the return /
    measures of exactly that (scene, episode) of the in-process replay, each episode is counted once, at least
    `test_episode_count` are evaluated, envs pause when their next episode is already covered, aggregates are plain means."""
    import types
    from habitat_amd.common import spaces
    from habitat_amd.config.default import get_config
    from habitat_amd.rl.ppo.evaluator import HabitatEvaluator, extract_scalars_from_info
    from habitat_amd.rl.ppo.policy import PolicyActionData

    n, K = 3, 8
    cfg = get_config("pointnav/ppo_pointnav_example.yaml", [f"habitat_baselines.num_environments={n}",
                                                             f"habitat_baselines.test_episode_count={K}"])

    class AC:
        policy_action_space = spaces.Discrete(4)
        hidden_state_shape = (1, 8)
        paused = []

        def act(self, obs, h, prev_actions, masks, deterministic=False):
            b = h.shape[0]
            assert obs["rgb"].shape[0] == b == prev_actions.shape[0] == masks.shape[0]
            return PolicyActionData(rnn_hidden_states=h + 1, actions=torch.randint(0, 4, (b, 1)))

        def get_extra(self, action_data, infos, dones):
            return []

        def on_envs_pause(self, idx):
            self.paused.append(list(idx))

    agent = types.SimpleNamespace(actor_critic=AC(), masks_shape=(1,), eval=lambda: None)
    scalars = {}
    writer = types.SimpleNamespace(add_scalar=lambda k, v, step: scalars.__setitem__(k, (v, step)))
    ev = HabitatEvaluator()
    with VectorEnv(make_host_env, _args(n, 6)) as envs:
        agg = ev.evaluate_agent(agent, envs, cfg, 0, 123, writer, torch.device("cpu"), [], None, set())
    rec = ev.last_stats_episodes
    assert len(rec) >= K and all(cnt == 1 for (_, cnt) in rec)
    # in-process replay of every env for long enough; rewards do not depend on the (valid) action taken
    expect = {}
    for a in _args(n, 6):
        e = make_host_env(*a)
        e.reset()
        for _ in range(400):
            ep = e.current_episode
            _, r, d, info = e.step(0)
            if d:
                expect[(ep.scene_id, ep.episode_id)] = dict(reward=info["episode_return"], **extract_scalars_from_info(info))
                e.reset()
    for (key, _), st in rec.items():
        assert key in expect, key
        for k, v in st.items():
            assert abs(v - expect[key][k]) < 1e-5, (key, k)
    for k in ("reward", "episode_return", "num_steps"):
        assert abs(agg[k] - np.mean([st[k] for st in rec.values()])) < 1e-6
    assert scalars["eval_reward/average_reward"] == (agg["reward"], 123) and "eval_metrics/num_steps" in scalars
    # finite episode lists (2 per env), test_episode_count = -1 -> every episode exactly once; an env is paused as soon as its
    # next episode is one that has already been evaluated, the tensors shrink with it and the policy is told
    cfg.habitat_baselines.test_episode_count = -1
    agent.actor_critic.paused.clear()
    with VectorEnv(make_host_env, [a + (2,) for a in _args(n, 6)]) as envs:
        ev.evaluate_agent(agent, envs, cfg, 0, 5, writer, torch.device("cpu"), [], None, set())
        assert envs.num_envs == 0  # all paused
    keys = sorted(k for (k, _) in ev.last_stats_episodes)
    assert keys == sorted((f"synthetic-{100 + i}", str(e)) for i in range(n) for e in (0, 1))
    assert sum(len(p) for p in agent.actor_critic.paused) == n
