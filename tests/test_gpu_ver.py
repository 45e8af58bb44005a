"""GPU: Variable Experience Rollouts against a fixture produced by the REFERENCE's own VER classes
(tests/golden/make_golden.py::ver_case drives rl/ver/inference_worker.py's step() over rl/ver/ver_rollout_storage.py).

The request batches the reference worker was handed are replayed on habitat_amd's InferenceWorker + VERRolloutStorage with the
policy on the HIP engine; after every phase of two consecutive rollouts the arena and the bookkeeping arrays must equal the
reference's: slot assignment, ids, masks, sampled actions and all integer state exactly; values / log-probs / hidden states to
1e-4; the packed GAE bit-for-bit when fed the reference's value predictions; minibatch composition exactly (same numpy draws); the
importance-weighted PPO update (metrics, parameters after all Adam steps) to 1e-4; the post-update reordering of the buffer."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import synth
from oracle.fixtures import baseline_param_shapes, det_params

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
GOAL = "pointgoal_with_gps_compass"


class ScriptedTransport:
    """In-process stand-in for the environment workers: per-environment synthetic envs stepped when the script says so."""

    def __init__(self, c, env_step):
        N = c["N"]
        self.c, self.env_step, self.num_envs = c, env_step, N
        self.rewards = np.zeros(N, np.float32)
        self.masks = np.zeros(N, bool)
        self.episode_ids = np.zeros(N, np.int64)
        self.step_ids = np.zeros(N, np.int64)
        self.env_t = np.zeros(N, np.int64)
        self.stepping = np.zeros(N, bool)
        self.obs = [self.env_step(c, e, 0)[0] for e in range(N)]
        self.sent = []

    def arrive(self, e):
        self.env_t[e] += 1
        obs, rew, done = self.env_step(self.c, e, int(self.env_t[e]))
        self.step_ids[e] += 1
        if done:
            self.episode_ids[e] += 1
            self.step_ids[e] = 0
        self.obs[e], self.rewards[e], self.masks[e] = obs, rew, not done
        self.stepping[e] = False

    def observations(self, env_ids, device):
        return {k: torch.from_numpy(np.stack([self.obs[e][k] for e in env_ids])).to(device) for k in self.obs[0]}

    def send_action(self, env_idx, action):
        self.stepping[env_idx] = True
        self.sent.append((int(env_idx), int(np.asarray(action).reshape(-1)[0])))


def _cmp_snapshot(z, tag, st, float_tol=1e-4):
    B = st.buffers
    for k in ("policy_version", "environment_ids", "episode_ids", "step_ids", "masks", "actions", "prev_actions", "is_stale"):
        assert np.array_equal(B[k].cpu().numpy(), z[f"{tag}/buf/{k}"]), (tag, k)
    assert np.array_equal(B["rewards"].cpu().numpy(), z[f"{tag}/buf/rewards"]), (tag, "rewards")
    for k in ("value_preds", "action_log_probs", "recurrent_hidden_states", "is_coeffs"):
        ref, got = z[f"{tag}/buf/{k}"], B[k].cpu().numpy()
        assert np.abs(got - ref).max() <= float_tol * max(1.0, np.abs(ref).max()), (tag, k, np.abs(got - ref).max())
    ref, got = z[f"{tag}/buf/returns"], B["returns"].cpu().numpy()
    assert np.array_equal(np.isnan(ref), np.isnan(got)), (tag, "returns nan pattern")
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    if fin.any():
        assert np.abs(got[fin] - ref[fin]).max() <= float_tol * max(1.0, np.abs(ref[fin]).max()), (tag, "returns")
    ds = B["observations"]["depth"].flatten(1).sum(1).cpu().numpy()
    assert np.allclose(ds, z[f"{tag}/buf/obs_depth_sum"], rtol=1e-4, atol=1e-2), (tag, "observations")
    for k in ("ptr", "prev_inds", "num_steps_collected", "rollout_done", "current_steps", "actor_steps_collected", "will_replay_step",
              "_first_rollout", "cpu_current_policy_version"):
        assert np.array_equal(np.asarray(getattr(st, k)).reshape(-1), z[f"{tag}/aux/{k}"].reshape(-1)), (tag, k)
    assert np.array_equal(st.next_prev_actions.cpu().numpy(), z[f"{tag}/aux/next_prev_actions"]), tag
    assert np.abs(st.next_hidden_states.cpu().numpy() - z[f"{tag}/aux/next_hidden_states"]).max() <= float_tol, tag


def test_ver_replay_of_reference_inference_worker_golden():
    from make_golden import VER_CASE as c, ver_env_step
    from habitat_amd.common import spaces as S
    from habitat_amd.rl.ppo import PPO, PointNavBaselinePolicy
    from habitat_amd.rl.ver.inference_worker import InferenceWorker
    from habitat_amd.rl.ver.ver_rollout_storage import VERRolloutStorage, generate_ver_mini_batches
    z = np.load(os.path.join(HERE, "golden", "ver_baseline_rgbd44.npz"))
    N, T, H, W = c["N"], c["T"], c["H"], c["W"]
    osp = S.Dict({"rgb": S.Box(0, 255, (H, W, 3), np.uint8), "depth": S.Box(0.0, 1.0, (H, W, 1), np.float32),
                  GOAL: S.Box(-1e9, 1e9, (2,), np.float32)})
    asp = S.Discrete(4)
    pol = PointNavBaselinePolicy(osp, asp, hidden_size=c["hidden"], max_frames=(T + 1) * N, max_envs=N)
    pol.load_state_dict(det_params(baseline_param_shapes(4, H, W, c["hidden"]), c["seed"]))
    pol.to("cuda")
    pol.eval()
    st = VERRolloutStorage(T, N, osp, asp, pol, variable_experience=True, device="cuda")
    assert st.buffers["returns"].shape == ((T + 1) * N, 1) and st.num_steps_to_collect == (T + 1) * N
    tr = ScriptedTransport(c, ver_env_step)
    cfg_full = types.SimpleNamespace(habitat_baselines=types.SimpleNamespace(rl=types.SimpleNamespace(ddppo=types.SimpleNamespace(train_encoder=True))))
    iw = InferenceWorker(cfg_full, pol, st, tr, "cuda")
    cfg = types.SimpleNamespace(use_gae=True, gamma=0.99, tau=0.95, **c["cfg"])
    ppo = PPO.from_config(pol, cfg)
    for r in range(2):
        nb = int(z[f"r{r}/num_batches"])
        for i in range(nb):
            batch = z[f"r{r}/batch{i}"].tolist()
            assert iw.new_reqs == batch[:len(iw.new_reqs)], "replayed requests must lead the first batch"
            for e in batch:
                if tr.stepping[e]:
                    tr.arrive(e)
            iw.new_reqs = list(batch)
            noise = torch.from_numpy(z[f"r{r}/noise{i}"]).cuda()
            stepped, _ = iw.step(exp_noise=noise)
            assert stepped
            iw._n_replay_steps = 0
        assert bool(st.rollout_done)
        # steps that had arrived but had not been picked up when the rollout filled (still in the inference queue of the reference)
        for e in z[f"r{r}/replay_after"].tolist():
            if e not in iw.replay_reqs and e not in iw.new_reqs:
                assert tr.stepping[e]
                tr.arrive(e)
                iw.new_reqs.append(e)
        iw.finish_rollout()
        assert iw.new_reqs == z[f"r{r}/replay_after"].tolist()
        assert np.array_equal(tr.stepping, z[f"r{r}/in_flight_after"])
        _cmp_snapshot(z, f"r{r}/collected", st)
        st.after_rollout()
        st.compute_returns(cfg.use_gae, cfg.gamma, cfg.tau)
        _cmp_snapshot(z, f"r{r}/returns", st)
        for k in ("select_inds", "num_seqs_at_step", "sequence_lengths", "sequence_starts", "last_sequence_in_batch_mask"):
            assert np.array_equal(np.asarray(getattr(st, k)), z[f"r{r}/pack/{k}"]), (r, k)
        # the packed GAE itself, bit for bit: same inputs as the reference's numpy float64 loop
        B = st.buffers
        keep = {k: B[k].clone() for k in ("returns", "value_preds")}
        B["returns"].copy_(torch.from_numpy(z[f"r{r}/collected/buf/returns"]))
        B["value_preds"].copy_(torch.from_numpy(z[f"r{r}/returns/buf/value_preds"]))
        st.compute_returns(cfg.use_gae, cfg.gamma, cfg.tau)
        got, ref = B["returns"].cpu().numpy(), z[f"r{r}/returns/buf/returns"]
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)]), "VER GAE not bit-exact"
        for k, v in keep.items():
            B[k].copy_(v)
        # minibatch composition (two draws from numpy's global generator, ver_rollout_storage.py:92,116)
        np.random.seed(c["seed"] + 10 + r)
        state = np.random.get_state()
        adv = ppo.get_advantages(st)
        mbs = list(st.data_generator(adv, cfg.num_mini_batch))
        for i, mb in enumerate(mbs):
            assert np.array_equal(mb.inds_cpu.numpy(), z[f"r{r}/mb{i}"]), (r, i)
            assert np.array_equal(mb.pack.arrays["first_step_for_env"], z[f"r{r}/mb{i}_first_step_for_env"])
            assert np.array_equal(mb.pack.arrays["sequence_lengths"], z[f"r{r}/mb{i}_sequence_lengths"])
        np.random.set_state(state)
        pol.train()
        metrics = ppo.update(st)
        pol.eval()
        for k in z.files:
            if k.startswith(f"r{r}/metric/"):
                name, ref = k.split("/")[-1], float(z[k])
                assert name in metrics, name
                assert abs(metrics[name] - ref) <= 1e-4 * max(1.0, abs(ref)), (r, name, metrics[name], ref)
        for k, v in pol.state_dict().items():
            ref = z[f"r{r}/post/{k}"]
            assert np.abs(v.cpu().numpy() - ref).max() <= 1e-4 * max(1e-2, np.abs(ref).max()), (r, k)
        st.after_update()
        st.increment_policy_version()
        _cmp_snapshot(z, f"r{r}/after_update", st)


@pytest.mark.parametrize("env_source,n_iw,overlap", [("device", 1, False), ("process", 1, False), ("device", 3, False), ("device", 2, True),
                                                      ("process", 2, True)])
def test_ver_trainer_update_cycles(env_source, n_iw, overlap, tmp_path):
    """The registered "ver" trainer from its YAML entrypoint: rollouts of a fixed number of steps collected from environments that
    finish at different rates (device-resident synthetic source with per-environment arrival rates / worker processes), PPO on the
    linear buffer with importance weights, policy versions advancing, finite losses, parameters moving, step accounting.  With one
    inference worker inside the trainer's thread (the reference's default VER configuration), with several worker threads on their
    own streams and private engines, and with collection overlapped with learning (ver_trainer.py:261-337,493-530)."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    import habitat_amd.rl.ver.ver_trainer  # noqa: F401
    N, T, size = 4, 8, 64
    ov = [f"habitat_baselines.num_environments={N}", f"habitat_baselines.rl.ppo.num_steps={T}", "habitat_baselines.num_updates=3",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", f"habitat_baselines.checkpoint_folder={tmp_path}", "habitat_baselines.log_interval=1",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000", "habitat_baselines.rl.ddppo.backbone=resnet18",
          f"habitat_baselines.rl.ver.num_inference_workers={n_iw}", f"habitat_baselines.rl.ver.overlap_rollouts_and_learn={overlap}"]
    for s in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{s}.height={size}", f"habitat.simulator.sensors.{s}.width={size}"]
    if env_source == "process":
        ov += ["habitat_baselines.vector_env_factory._target_=habitat_amd.common.env_factory.ProcessVectorEnvFactory"]
    cfg = get_config("pointnav/ver_pointnav.yaml", ov)
    cfg.habitat.simulator.sensors.pop("semantic", None)
    if env_source == "device":
        cfg.habitat.synthetic["ver_speeds"] = [1.0, 0.7, 0.3, 0.9]
    trainer = baseline_registry.get_trainer("ver")(cfg)
    trainer._init_train()
    assert len(trainer.inference_workers) == n_iw and len(trainer._iw_pool.threads) == (n_iw if overlap else n_iw - 1)
    assert (trainer.learning_rollouts is not trainer._agent.rollouts) == overlap
    st = trainer.learning_rollouts
    pol = trainer._agent.actor_critic
    before = pol.engine.params_flat.clone()
    collected = []
    for u in range(3):
        losses = trainer.run_update_cycle()
        assert all(np.isfinite(v) for v in losses.values()), losses
        collected.append(trainer.num_steps_done)
        assert int(trainer._agent.rollouts.cpu_current_policy_version[0, 0]) == u + 2
        # every slot of the learner's arena holds a step: ids consistent with the observations' source, no slot left from an
        # earlier layout (each (environment, episode, step) triple at most once)
        ids = torch.stack([st.buffers[k].view(-1) for k in ("environment_ids", "episode_ids", "step_ids")], 1).cpu().numpy()
        assert len({tuple(r) for r in ids}) == len(ids)
        assert torch.isfinite(st.buffers["value_preds"]).all() and torch.isfinite(st.buffers["action_log_probs"]).all()
    # the first rollout fills the whole buffer, the following ones collect num_envs * num_steps
    assert collected == [(T + 1) * N, (T + 1) * N + N * T, (T + 1) * N + 2 * N * T], collected
    assert {"value_loss", "action_loss", "dist_entropy", "grad_norm", "ver_is_coeffs_mean", "fraction_stale",
            "policy_version_difference_mean"} <= set(losses)
    assert float((pol.engine.params_flat - before).abs().max()) > 0
    if env_source == "device" and n_iw == 1:  # uneven arrival rates -> uneven contributions -> importance coefficients away from 1
        counts = torch.bincount(st.buffers["environment_ids"].view(-1), minlength=N).cpu().numpy()
        assert counts.max() > counts.min(), counts
    if n_iw > 1 or overlap:  # the private engines follow the learner: after the last update every worker that acted since holds
        for iw in trainer.inference_workers:  # the published parameters
            if iw.published is not None and iw._current_policy_version == 4:
                assert torch.equal(iw.actor_critic.engine.params_flat, trainer._published.flat)
        assert torch.equal(trainer._published.flat, pol.engine.params_flat)
    trainer.shutdown()
    assert all(not t.is_alive() for t in trainer._iw_pool.threads)
    trainer.envs.close()


def test_ver_overlapped_run_hands_the_learner_the_same_rollouts_as_the_sequential_run(tmp_path, monkeypatch):
    """Fixed arrival order (device source, every step arrives at the next poll, one inference worker) and a learning rate of zero
    (the acting parameters are the same whichever policy version an engine holds): the overlapped run (worker thread on its own
    stream with a private engine, learner on its copy of the arena) must hand the learner bit-identical rollouts to the sequential
    run -- observations, actions, log-probs, values, rewards, masks, ids, hidden states -- cycle after cycle; only the policy-version
    stamps differ (a rollout of the overlapped run starts before the previous update has finished)."""
    monkeypatch.setenv("HAB_VER_PREEMPTION", "0")  # a preemption deadline depends on wall-clock step times: two runs would legitimately differ
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    import habitat_amd.rl.ver.ver_trainer  # noqa: F401
    N, T, size = 4, 8, 64
    snaps = {}
    for overlap in (False, True):
        ov = [f"habitat_baselines.num_environments={N}", f"habitat_baselines.rl.ppo.num_steps={T}", "habitat_baselines.num_updates=4",
              "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
              "habitat_baselines.rl.ppo.hidden_size=64", f"habitat_baselines.checkpoint_folder={tmp_path}", "habitat_baselines.log_interval=100",
              "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000", "habitat_baselines.rl.ddppo.backbone=resnet18",
              "habitat_baselines.rl.ppo.lr=0.0", f"habitat_baselines.rl.ver.overlap_rollouts_and_learn={overlap}",
              "habitat_baselines.rl.ver.num_inference_workers=1",
              # SimpleCNN policy: no RunningMeanAndVar buffers, which a training-mode forward moves even at a learning rate of zero
              "habitat_baselines.rl.policy.main_agent.name=PointNavBaselinePolicy"]
        for s_ in ("rgb", "depth"):
            ov += [f"habitat.simulator.sensors.{s_}.height={size}", f"habitat.simulator.sensors.{s_}.width={size}"]
        cfg = get_config("pointnav/ver_pointnav.yaml", ov)
        cfg.habitat.simulator.sensors.pop("semantic", None)
        torch.manual_seed(3)
        np.random.seed(3)
        trainer = baseline_registry.get_trainer("ver")(cfg)
        trainer._init_train()
        out = []
        orig = trainer._update_agent

        def snap_then_update():
            st = trainer.learning_rollouts
            torch.cuda.synchronize()
            rec = {}
            for k, v in st.buffers.items():
                rec[k] = {kk: vv.cpu().clone() for kk, vv in v.items()} if isinstance(v, dict) else v.cpu().clone()
            out.append(rec)
            return orig()
        trainer._update_agent = snap_then_update
        for _ in range(3):
            trainer.run_update_cycle()
        trainer.shutdown()
        trainer.envs.close()
        snaps[overlap] = out
    for k, (a, b) in enumerate(zip(snaps[False], snaps[True])):
        for key in a:
            if key in ("policy_version", "is_stale", "returns", "is_coeffs"):
                continue
            if isinstance(a[key], dict):
                for kk in a[key]:
                    assert torch.equal(a[key][kk], b[key][kk]), (k, key, kk)
            elif key == "rewards":
                # the reward of a step arrives with the environment's NEXT step: the last step of every environment (the bootstrap
                # step, whose reward is never used) still holds whatever the slot held before
                env, ep, step = (a[q].view(-1).numpy() for q in ("environment_ids", "episode_ids", "step_ids"))
                order = ep * (step.max() + 1) + step
                keep = np.ones(len(env), bool)
                for e in range(N):
                    idx = np.nonzero(env == e)[0]
                    keep[idx[np.argmax(order[idx])]] = False
                keep = torch.from_numpy(keep)
                assert torch.equal(a[key].view(-1)[keep], b[key].view(-1)[keep]), (k, key)
            else:
                assert torch.equal(a[key], b[key]), (k, key)
        assert torch.equal(a["is_coeffs"], b["is_coeffs"]), k


def test_ver_trainer_with_the_preemption_decider_enabled(tmp_path, monkeypatch):
    """The default (HAB_VER_PREEMPTION unset = on) on one rank: the decider is fed by the workers' step reports, becomes ready after five learner times, sets
    deadlines, and the trainer keeps producing finite updates with rollouts of at most the step quota (on one rank with variable
    experience the optimum is normally the full quota: fast environments fill it)."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    import habitat_amd.rl.ver.ver_trainer  # noqa: F401
    monkeypatch.delenv("HAB_VER_PREEMPTION", raising=False)
    N, T, size = 4, 8, 64
    ov = [f"habitat_baselines.num_environments={N}", f"habitat_baselines.rl.ppo.num_steps={T}", "habitat_baselines.num_updates=8",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", f"habitat_baselines.checkpoint_folder={tmp_path}", "habitat_baselines.log_interval=100",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000",
          "habitat_baselines.rl.policy.main_agent.name=PointNavBaselinePolicy", "habitat_baselines.rl.ver.num_inference_workers=2"]
    for s in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{s}.height={size}", f"habitat.simulator.sensors.{s}.width={size}"]
    cfg = get_config("pointnav/ver_pointnav.yaml", ov)
    cfg.habitat.simulator.sensors.pop("semantic", None)
    cfg.habitat.synthetic["ver_speeds"] = [1.0, 0.8, 0.5, 0.9]
    trainer = baseline_registry.get_trainer("ver")(cfg)
    trainer._init_train()
    d = trainer._decider
    assert d is not None and all(iw.decider is d for iw in trainer.inference_workers)
    steps = []
    for u in range(8):
        before = trainer.num_steps_done
        losses = trainer.run_update_cycle()
        assert all(np.isfinite(v) for v in losses.values()), losses
        steps.append(trainer.num_steps_done - before)
    assert steps[0] == (T + 1) * N and all(0 < s <= N * T for s in steps[1:]), steps
    # 8 finished rollouts + the next one, armed in _update_agent BEFORE the parked workers are released (its first batches count)
    assert d.learner_time_avg.count == 5 and d.n_rollouts_started == 9
    assert sum(v.count for v in d.step_averages) > 0  # the workers' step reports arrived
    trainer.shutdown()
    trainer.envs.close()
