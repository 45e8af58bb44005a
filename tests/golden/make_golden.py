#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by running the REAL reference code
(/root/reference, loaded in place through oracle/ref_loader.py) on deterministic inputs.

  python tests/golden/make_golden.py          # needs /root/reference; writes tests/golden/*.npz

Inputs are NOT stored: parameters come from `det_params(seed)` and observations from the
counter-based synthetic generator (oracle/synth.py), both reproducible anywhere.  Only the
reference's outputs are stored.  The tests (which must run where /root/reference does not exist)
re-create the inputs and compare against these files.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402
from oracle.fixtures import det_params, golden_sample, synth_rollout_inputs  # noqa: E402
from oracle.ref_loader import load_reference, make_config  # noqa: E402

GOAL = "pointgoal_with_gps_compass"


def obs_space(ns, H, W, rgb=True, depth=True, task="pointnav"):
    sp = ns.spaces
    d = {}
    if rgb:
        d["rgb"] = sp.Box(0, 255, (H, W, 3), np.uint8)
    if depth:
        d["depth"] = sp.Box(0.0, 1.0, (H, W, 1), np.float32)
    if task == "objectnav":
        d["semantic"] = sp.Box(0, synth.NUM_SEMANTIC_IDS - 1, (H, W, 1), np.int32)
        d["objectgoal"] = sp.Box(0, synth.NUM_OBJECT_CATEGORIES - 1, (1,), np.int64)
        d["compass"] = sp.Box(-np.pi, np.pi, (1,), np.float32)
        d["gps"] = sp.Box(-1e9, 1e9, (2,), np.float32)
    else:
        d[GOAL] = sp.Box(-1e9, 1e9, (2,), np.float32)
    return sp.Dict(d)


class ReluMargin:
    """Smallest |pre-ReLU| value seen while active (nn.ReLU.forward -> F.relu): a ReLU input inside fp32 round-off of zero (~1e-6
    after 20 GroupNorm layers) lands on either side in two correct implementations -- or in two runs of THIS script at different
    thread counts -- and flips a mask bit of the backward.  Fixtures of deep encoders record the margin of the minibatch whose
    gradients they store."""

    def __init__(self):
        self.min = float("inf")

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._orig = F, F.relu

        def probe(x, inplace=False):
            self.min = min(self.min, float(x.detach().abs().min()))
            return self._orig(x, inplace=inplace)

        F.relu = probe
        return self

    def __exit__(self, *a):
        self._F.relu = self._orig


def run_case(ns, name, policy, space, cfg, T, N, seed, H, W, rgb=True, depth=True, sampled=False, task="pointnav", num_actions=4,
             action_space=None, margin_only=False):
    """Reference rollout (policy.act through RolloutStorage) + compute_returns + PPO.update.  action_space: a Box for the Gaussian
    head (the stored noise is then the N(0, 1) draw of CustomNormal.rsample instead of multinomial's Exp(1))."""
    gaussian = action_space is not None
    sd = policy.state_dict()
    newp = det_params([(k, v.shape) for k, v in sd.items() if v.dtype == torch.float32 and "running_mean_and_var" not in k], seed)
    sd.update(newp)
    policy.load_state_dict(sd)
    RolloutStorage = ns.rollout_storage.RolloutStorage
    rollouts = RolloutStorage(T, N, space, action_space if gaussian else ns.spaces.Discrete(num_actions), policy)
    envs = synth.SyntheticEnvs(N, H, W, seed=seed, use_rgb=rgb, use_depth=depth, task=task)
    obs, rew, done = synth_rollout_inputs(envs, T)
    to_t = lambda o: {k: torch.from_numpy(v) for k, v in o.items()}
    rollouts.insert_first_observations(to_t(obs[0]))
    out = {}
    policy.eval()
    torch.manual_seed(seed)
    noises = []
    for t in range(T):
        step = rollouts.get_current_step(slice(0, N), 0)
        with torch.no_grad():
            rng_state = torch.get_rng_state()
            if gaussian:  # torch.distributions.Normal.rsample -> _standard_normal(shape) = torch.normal(zeros, ones)
                noises.append(torch.normal(torch.zeros(N, num_actions), torch.ones(N, num_actions)))
            else:
                noises.append(torch.empty(N, num_actions).exponential_(1))  # what multinomial is about to draw
            torch.set_rng_state(rng_state)
            ad = policy.act(step["observations"], step["recurrent_hidden_states"], step["prev_actions"], step["masks"])
        rollouts.insert(next_recurrent_hidden_states=ad.rnn_hidden_states, actions=ad.actions,
                        action_log_probs=ad.action_log_probs, value_preds=ad.values)
        rollouts.insert(next_observations=to_t(obs[t + 1]), rewards=torch.from_numpy(rew[t]).unsqueeze(1),
                        next_masks=torch.from_numpy(~done[t]).unsqueeze(1))
        rollouts.advance_rollout()
    out["exp_noise"] = torch.stack(noises).numpy()
    with torch.no_grad():
        last = rollouts.get_last_step()
        next_value = policy.get_value(last["observations"], last["recurrent_hidden_states"], last["prev_actions"], last["masks"])
    rollouts.compute_returns(next_value, cfg.use_gae, cfg.gamma, cfg.tau)
    b = rollouts.buffers
    for k in ("actions", "prev_actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "recurrent_hidden_states"):
        out["roll_" + k] = b[k].numpy().copy()
    out["next_value"] = next_value.numpy()

    policy.train()
    ppo = ns.ppo.PPO.from_config(policy, cfg)
    adaptive = isinstance(ppo.entropy_coef, torch.nn.Module)
    out["advantages"] = ppo.get_advantages(rollouts).numpy().copy()
    # first minibatch of a fixed permutation: evaluate_actions + grads (no optimiser step)
    torch.manual_seed(seed + 1)
    gen = rollouts.data_generator(ppo.get_advantages(rollouts), cfg.num_mini_batch)
    batch = next(gen)
    with ReluMargin() as rm:
        v, lp, ent, hfin, _ = policy.evaluate_actions(batch["observations"], batch["recurrent_hidden_states"], batch["prev_actions"],
                                                      batch["masks"], batch["actions"], batch["rnn_build_seq_info"])
    out["mb0_relu_margin"] = np.float64(rm.min)
    if margin_only:
        return rm.min
    out["mb0_value"], out["mb0_logp"], out["mb0_entropy"], out["mb0_hidden"] = (
        v.detach().numpy(), lp.detach().numpy(), ent.detach().numpy(), hfin.detach().numpy())
    ratio = torch.exp(lp - batch["action_log_probs"])
    s1 = batch["advantages"] * ratio
    s2 = batch["advantages"] * torch.clamp(ratio, 1 - cfg.clip_param, 1 + cfg.clip_param)
    al = -torch.min(s1, s2).mean()
    delta = v.detach() - batch["value_preds"]
    vc = batch["value_preds"] + delta.clamp(-cfg.clip_param, cfg.clip_param)
    vv = torch.where(delta.abs() < cfg.clip_param, v, vc)
    vl = (0.5 * (vv - batch["returns"]) ** 2).mean()
    total = cfg.value_loss_coef * vl + al + (ppo.entropy_coef.lagrangian_loss(ent.mean()) if adaptive else -cfg.entropy_coef * ent.mean())
    policy.zero_grad()
    ppo.zero_grad()
    total.backward()
    out["mb0_losses"] = np.array([vl.item(), al.item(), ent.mean().item(), total.item()], dtype=np.float32)
    keep = (lambda a: golden_sample(a).copy()) if sampled else (lambda a: a.copy())
    for k, p_ in policy.named_parameters():
        if p_.grad is not None:
            out["grad/" + k] = keep(p_.grad.numpy())
            if sampled:
                out["gradnorm/" + k] = np.float64(np.linalg.norm(p_.grad.numpy().astype(np.float64)))
    if adaptive:
        out["mb0_grad_log_alpha"] = ppo.entropy_coef.log_alpha.grad.numpy().copy()
    policy.zero_grad()
    ppo.zero_grad()
    # the full update (fresh generator state so the permutations are reproducible: seed + 2)
    torch.manual_seed(seed + 2)
    perm_state = torch.get_rng_state()
    perms = [torch.randperm(N) for _ in range(cfg.ppo_epoch)]
    torch.set_rng_state(perm_state)
    out["perms"] = torch.stack(perms).numpy()
    metrics = ppo.update(rollouts)
    for k, val in metrics.items():
        out["metric/" + k] = np.float32(val)
    for k, p_ in policy.state_dict().items():
        out["post/" + k] = keep(p_.numpy())
    if adaptive:
        out["post_log_alpha"] = ppo.entropy_coef.log_alpha.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "->", len(out), "arrays;", {k: float(v) for k, v in metrics.items()})


def pack_cases(ns):
    rng = np.random.default_rng(7)
    out = {}
    cases = [(1, 1), (4, 3), (8, 5), (16, 4), (32, 8), (13, 3), (64, 6), (128, 4)]
    for i, (T, N) in enumerate(cases):
        dones = rng.random((T, N)) < (1.0 / 6.0)
        out[f"c{i}_dones"] = dones
        info = ns.rnn_state_encoder.build_pack_info_from_dones(dones)
        for k, v in info.items():
            out[f"c{i}_{k}"] = np.asarray(v)
    out["num_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "pack_info.npz"), **out)
    print("pack_info ->", len(cases), "cases")


def rnn_case(ns):
    """test/test_rnn_state_encoder.py semantics: packed seq_forward on random masks, GRU and LSTM."""
    out = {}
    for kind, layers in (("GRU", 2), ("LSTM", 2)):
        torch.manual_seed(11)
        enc = ns.rnn_state_encoder.build_rnn_state_encoder(16, 32, rnn_type=kind, num_layers=layers)
        sd = enc.state_dict()
        sd.update(det_params([(k, v.shape) for k, v in sd.items()], 5))
        enc.load_state_dict(sd)
        T, N = 12, 5
        rng = np.random.default_rng(3)
        x = torch.from_numpy(rng.standard_normal((T * N, 16)).astype(np.float32))
        masks = torch.from_numpy(rng.random((T, N)) > (1.0 / 5.0)).view(T * N, 1)
        h0 = torch.from_numpy(rng.standard_normal((N, enc.num_recurrent_layers, 32)).astype(np.float32))
        info = ns.rnn_state_encoder.build_pack_info_from_dones(np.logical_not(masks.view(T, N).numpy()))
        seq = ns.rnn_state_encoder.build_rnn_build_seq_info(torch.device("cpu"), info)
        with torch.no_grad():
            o, h = enc(x, h0, masks, seq)
        out[kind + "_out"], out[kind + "_hidden"] = o.numpy(), h.numpy()
    np.savez_compressed(os.path.join(HERE, "rnn_encoder.npz"), **out)
    print("rnn_encoder -> GRU, LSTM")


def c1_case(ns):
    """BASELINE.json configs[0] EXACTLY: PointNavBaselinePolicy (SimpleCNN + GRU 512), 4 envs x 32 steps, 84x84 depth-only,
    the ppo section of config/pointnav/ppo_pointnav_example.yaml (clip 0.1, E=1, M=1, lr 2.5e-4, max_grad_norm 0.5, T=32; advantage
    normalisation off and clipped value loss on = default_structured_configs.py:303-316).  Multi-hundred-thousand-element tensors
    are kept as strided samples + norms."""
    space = obs_space(ns, 84, 84, rgb=False)
    torch.manual_seed(0)
    pol = ns.policy.PointNavBaselinePolicy(space, ns.spaces.Discrete(4), hidden_size=512)
    cfg = make_config(clip_param=0.1, ppo_epoch=1, num_mini_batch=1, max_grad_norm=0.5, num_steps=32, value_loss_coef=0.5,
                      entropy_coef=0.01, use_normalized_advantage=False, use_clipped_value_loss=True, hidden_size=512, lr=2.5e-4,
                      eps=1e-5)
    run_case(ns, "c1_depth84_h512_4x32", pol, space, cfg, T=32, N=4, seed=41, H=84, W=84, rgb=False, sampled=True)


VER_CASE = dict(H=44, W=44, N=4, T=6, hidden=64, seed=77, speeds=(0.9, 0.6, 0.35, 0.75), p_done=0.2,
                cfg=dict(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, max_grad_norm=0.5, use_normalized_advantage=False,
                         use_clipped_value_loss=True, entropy_coef=0.01, value_loss_coef=0.5, lr=2.5e-4, eps=1e-5))


def ver_env_step(c, env, t):
    """Synthetic env `env` at its own step counter t: (obs dict, reward, done).  done: a dedicated hash draw at rate p_done."""
    H, W, seed = c["H"], c["W"], c["seed"]
    obs = {"rgb": synth.rgb(seed, env, t, H, W), "depth": synth.depth(seed, env, t, H, W), GOAL: synth.goal(seed, env, t)}
    done = bool(synth.words(seed, synth.SENSOR_DONE, env, t, 1)[0] < np.uint32(int(c["p_done"] * (1 << 32))))
    return obs, float(synth.reward(seed, env, t)), done


def ver_case():
    """Variable Experience Rollouts, driven through the REFERENCE's own classes: VERRolloutStorage, the
    InferenceWorkerProcess.step() that fills it (rl/ver/inference_worker.py:238-420), after_rollout / compute_returns, PPO.update
    with the importance-weighted loss (rl/ppo/ppo.py:226-231), after_update, and a second rollout on top of the reordered buffer.
    Environments finish their steps at different rates (a scripted scheduler with its own RandomState), so they contribute different
    numbers of steps.  Stored: the request batches the scheduler handed to step() (the test replays them), the Exp(1) noise of every
    sampling call, and snapshots of every buffer / bookkeeping array after each phase."""
    from oracle.ref_loader import load_reference_ver
    ns = load_reference_ver()
    c = VER_CASE
    N, T, H, W = c["N"], c["T"], c["H"], c["W"]
    VS, IW = ns.ver_rollout_storage.VERRolloutStorage, ns.inference_worker.InferenceWorkerProcess
    space = obs_space(ns, H, W)
    aspace = ns.spaces.Discrete(4)
    torch.manual_seed(0)
    policy = ns.policy.PointNavBaselinePolicy(space, aspace, hidden_size=c["hidden"])
    sd = policy.state_dict()
    sd.update(det_params([(k, v.shape) for k, v in sd.items()], c["seed"]))
    policy.load_state_dict(sd)
    with torch.inference_mode():  # as ver_trainer.py:255-262 creates them
        rollouts = VS(T, N, space, aspace, policy, variable_experience=True)
        tb = VS(1, N, space, aspace, policy, variable_experience=True).buffers.slice_keys(
            "rewards", "masks", "observations", "episode_ids", "environment_ids", "actions", "step_ids")[slice(0, N)]
        tb["environment_ids"][:] = torch.arange(N).view(N, 1)
    iw = object.__new__(IW)
    iw.__dict__.update(new_reqs=[], replay_reqs=[], rollouts=rollouts, _current_policy_version=int(rollouts.cpu_current_policy_version),
                       _overlapped=False, _variable_experience=True, num_inference_workers=1, _n_replay_steps=0,
                       timer=ns.timing.Timing(), device=torch.device("cpu"), obs_transforms=[], _static_encoder=False,
                       actor_critic=policy, visual_encoder=None, inference_worker_idx=0)
    iw.transfer_buffers = tb.numpy()
    iw.incoming_transfer_buffers = iw.transfer_buffers.slice_keys(set(iw.transfer_buffers.keys()) - {"actions"})
    send = iw.transfer_buffers.slice_keys("observations", "rewards", "masks", "episode_ids", "step_ids")  # environment_worker.py:223-229
    iw.queues = type("Q", (), {})()
    iw.queues.environments = [ns.BatchedQueue() for _ in range(N)]
    out = {}
    sched = np.random.RandomState(c["seed"])
    env_t = np.zeros(N, dtype=np.int64)
    ep_id = np.zeros(N, dtype=np.int64)
    step_id = np.zeros(N, dtype=np.int64)
    stepping = np.zeros(N, dtype=bool)
    policy.eval()
    torch.manual_seed(c["seed"] + 5)

    def snapshot(tag):
        for k, v in rollouts.buffers.items():
            if k == "observations":
                continue
            out[f"{tag}/buf/{k}"] = v.numpy().copy()
        out[f"{tag}/buf/obs_depth_sum"] = rollouts.buffers["observations"]["depth"].flatten(1).sum(1).numpy().copy()
        for k in ("ptr", "prev_inds", "num_steps_collected", "rollout_done", "current_steps", "actor_steps_collected",
                  "will_replay_step", "_first_rollout", "cpu_current_policy_version"):
            out[f"{tag}/aux/{k}"] = np.asarray(getattr(rollouts, k)).copy()
        out[f"{tag}/aux/next_hidden_states"] = rollouts.next_hidden_states.numpy().copy()
        out[f"{tag}/aux/next_prev_actions"] = rollouts.next_prev_actions.numpy().copy()

    def collect(rollout_idx, ready):
        batches, noises = [], []
        while not bool(rollouts.rollout_done):
            # environments with an outstanding step finish it with their own probability per scheduler tick
            for e in range(N):
                if stepping[e] and sched.random_sample() < c["speeds"][e]:
                    env_t[e] += 1
                    obs, rew, done = ver_env_step(c, e, int(env_t[e]))
                    step_id[e] += 1
                    if done:
                        ep_id[e] += 1
                        step_id[e] = 0
                    send[e] = dict(observations=obs, rewards=rew, masks=not done, episode_ids=int(ep_id[e]),
                                                           step_ids=int(step_id[e]))
                    stepping[e] = False
                    ready.append(e)
            if not ready and not iw.new_reqs:
                continue
            # the worker picks up what has arrived so far (possibly not everything); replayed requests of the last rollout are waiting
            k = int(sched.randint(0 if iw.new_reqs else 1, len(ready) + 1))
            iw.new_reqs = iw.new_reqs + ready[:k]
            del ready[:k]
            reqs = list(iw.new_reqs)
            n_proc = int(min(int(rollouts.num_steps_to_collect - rollouts.num_steps_collected), len(reqs)))  # inference_worker.py:262-270
            state = torch.get_rng_state()
            noises.append(torch.empty(n_proc, 4).exponential_(1).numpy())  # what multinomial is about to draw for the processed requests
            torch.set_rng_state(state)
            with torch.inference_mode():  # InferenceWorkerProcess.run is decorated with it
                stepped, _ = iw.step()
            assert stepped
            iw._n_replay_steps = 0
            batches.append(np.array(reqs, dtype=np.int64))
            for e in range(N):
                q = iw.queues.environments[e]
                if q.items:
                    q.items.clear()
                    stepping[e] = True
        # finish_rollout (inference_worker.py:422-456) for a single worker: outstanding + replay requests are replayed first
        iw.new_reqs = iw.replay_reqs + iw.new_reqs + ready
        del ready[:]
        iw.replay_reqs = []
        iw._n_replay_steps = len(iw.new_reqs)
        rollouts.will_replay_step[iw.new_reqs] = True
        out[f"r{rollout_idx}/num_batches"] = np.int64(len(batches))
        for i, (b, q) in enumerate(zip(batches, noises)):
            out[f"r{rollout_idx}/batch{i}"] = b
            out[f"r{rollout_idx}/noise{i}"] = q
        out[f"r{rollout_idx}/replay_after"] = np.array(iw.new_reqs, dtype=np.int64)
        out[f"r{rollout_idx}/in_flight_after"] = stepping.copy()

    # every env starts with its first observation on the table (start_experience_collection, environment_worker.py:148-160)
    ready0 = []
    for e in range(N):
        obs, _, _ = ver_env_step(c, e, 0)
        send.slice_keys("observations", "episode_ids", "step_ids")[e] = dict(observations=obs, episode_ids=0,
                                                                                                    step_ids=0)
        ready0.append(e)
    cfg = make_config(num_steps=T, hidden_size=c["hidden"], **c["cfg"])
    ppo = ns.ppo.PPO.from_config(policy, cfg)
    for r in range(2):
        collect(r, ready0 if r == 0 else [])
        snapshot(f"r{r}/collected")
        with torch.inference_mode():
            rollouts.after_rollout()
            rollouts.compute_returns(cfg.use_gae, cfg.gamma, cfg.tau)
        snapshot(f"r{r}/returns")
        for k in ("select_inds", "num_seqs_at_step", "sequence_lengths", "sequence_starts", "last_sequence_in_batch_mask"):
            out[f"r{r}/pack/{k}"] = np.asarray(getattr(rollouts, k)).copy()
        np.random.seed(c["seed"] + 10 + r)
        st = np.random.get_state()
        mbs = list(ns.ver_rollout_storage.generate_ver_mini_batches(cfg.num_mini_batch, rollouts.sequence_lengths, rollouts.num_seqs_at_step,
                                                                   rollouts.select_inds, rollouts.last_sequence_in_batch_mask,
                                                                   rollouts.episode_ids_cpu))
        for i, mb in enumerate(mbs):
            out[f"r{r}/mb{i}"] = mb
            info = ns.rnn_state_encoder.build_pack_info_from_episode_ids(rollouts.episode_ids_cpu[mb], rollouts.environment_ids_cpu[mb],
                                                                         rollouts.step_ids_cpu[mb])
            out[f"r{r}/mb{i}_first_step_for_env"] = info["first_step_for_env"]
            out[f"r{r}/mb{i}_sequence_lengths"] = info["sequence_lengths"]
        np.random.set_state(st)
        policy.train()
        metrics = ppo.update(rollouts)
        policy.eval()
        for k, v in metrics.items():
            out[f"r{r}/metric/{k}"] = np.float32(v)
        for k, v in policy.state_dict().items():
            out[f"r{r}/post/{k}"] = v.numpy().copy()
        with torch.inference_mode():
            rollouts.after_update()
            rollouts.increment_policy_version()
        iw._current_policy_version = int(rollouts.cpu_current_policy_version)
        snapshot(f"r{r}/after_update")
    np.savez_compressed(os.path.join(HERE, "ver_baseline_rgbd44.npz"), **out)
    print("ver_baseline_rgbd44 ->", len(out), "arrays;", {k: float(v) for k, v in metrics.items()})


def main():
    torch.set_num_threads(int(os.environ.get("HAB_GOLDEN_THREADS", "4")))  # EVERY branch: the fixtures must not depend on how the script was invoked (summation order of ATen's CPU kernels)
    if len(sys.argv) > 1 and sys.argv[1] == "se-seeds":  # margin scan used to choose SE_SEED below
        ns = load_reference()
        for seed in range(int(sys.argv[2]), int(sys.argv[3])):
            print(seed, se_resnext_case(ns, seed=seed, margin_only=True), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ver":
        ver_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "se":
        ns = load_reference()
        se_resnext_case(ns)
        se_resnext_case(ns, seed=SE_SEED_UNSELECTED, name="se_resnext50_rgbd128_seed7")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "nonsquare":
        baseline_nonsquare_case(load_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gaussian":
        gaussian_case(load_reference())
        return
    ns = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "c1":
        c1_case(ns)
        return
    pack_cases(ns)
    rnn_case(ns)
    c1_case(ns)
    # A: PointNavBaselinePolicy (SimpleCNN + GRU), 44x44 RGB-D, hidden 64, T=6, N=4, iccv19 hyper-parameters
    H = W = 44
    space = obs_space(ns, H, W)
    torch.manual_seed(0)
    pol = ns.policy.PointNavBaselinePolicy(space, ns.spaces.Discrete(4), hidden_size=64)
    cfg = make_config(clip_param=0.1, ppo_epoch=2, num_mini_batch=2, max_grad_norm=0.5, num_steps=6,
                      use_normalized_advantage=True, hidden_size=64, lr=2.5e-4, eps=1e-5)
    run_case(ns, "baseline_rgbd44", pol, space, cfg, T=6, N=4, seed=100, H=H, W=W)
    # A2: depth-only 84x84 (BASELINE.json configs[0] shape), hidden 32, no advantage normalisation, unclipped value loss
    space2 = obs_space(ns, 84, 84, rgb=False)
    torch.manual_seed(0)
    pol2 = ns.policy.PointNavBaselinePolicy(space2, ns.spaces.Discrete(4), hidden_size=64)
    cfg2 = make_config(clip_param=0.2, ppo_epoch=1, num_mini_batch=1, max_grad_norm=0.2, num_steps=5,
                       use_normalized_advantage=False, hidden_size=64, use_clipped_value_loss=False)
    run_case(ns, "baseline_depth84", pol2, space2, cfg2, T=5, N=3, seed=7, H=84, W=84, rgb=False)
    baseline_nonsquare_case(ns)
    # B: PointNavResNetPolicy (resnet18 GroupNorm encoder + 2-layer LSTM, RunningMeanAndVar on), 256x256 RGB-D (BASELINE.json
    # configs[2] geometry), hidden 64, ddppo_pointnav hyper-parameters.  Large tensors are stored as strided samples + norms.
    space3 = obs_space(ns, 256, 256)
    torch.manual_seed(0)
    pol3 = ns.resnet_policy.PointNavResNetPolicy(space3, ns.spaces.Discrete(4), hidden_size=64, num_recurrent_layers=2,
                                                 rnn_type="LSTM", backbone="resnet18", normalize_visual_inputs=True)
    cfg3 = make_config(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, max_grad_norm=0.2, num_steps=4,
                       use_normalized_advantage=False, hidden_size=64, lr=2.5e-4, eps=1e-5)
    run_case(ns, "resnet18_rgbd256", pol3, space3, cfg3, T=4, N=2, seed=21, H=256, W=256, sampled=True)
    # C: ObjectNav inputs (BASELINE.json configs[4] geometry): ResNet50 on rgb + depth + semantic (5 channels), objectgoal /
    # compass / gps embeddings, Discrete(6), 2-layer LSTM, ddppo_objectnav hyper-parameters (E=4 shortened to 2, M=2).
    space4 = obs_space(ns, 256, 256, task="objectnav")
    torch.manual_seed(0)
    pol4 = ns.resnet_policy.PointNavResNetPolicy(space4, ns.spaces.Discrete(6), hidden_size=64, num_recurrent_layers=2,
                                                 rnn_type="LSTM", backbone="resnet50", normalize_visual_inputs=True)
    cfg4 = make_config(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, max_grad_norm=0.2, num_steps=3,
                       use_normalized_advantage=False, hidden_size=64, lr=2.5e-4, eps=1e-5)
    run_case(ns, "objectnav_resnet50_256", pol4, space4, cfg4, T=3, N=2, seed=33, H=256, W=256, sampled=True, task="objectnav",
             num_actions=6)
    ver_case()
    gaussian_case(ns)
    se_resnext_case(ns)
    se_resnext_case(ns, seed=SE_SEED_UNSELECTED, name="se_resnext50_rgbd128_seed7")


def baseline_nonsquare_case(ns):
    """A3: SimpleCNN + GRU on 96 x 128 RGB-D (simple_cnn.py:35-93 takes any size): conv1 -> 23 x 31, conv2 -> 10 x 14, conv3 -> 8 x 12.
    16-frame minibatches: the geometry-general instantiations of the strip kernels run at engine level against the reference."""
    space = obs_space(ns, 96, 128)
    torch.manual_seed(0)
    pol = ns.policy.PointNavBaselinePolicy(space, ns.spaces.Discrete(4), hidden_size=64)
    cfg = make_config(clip_param=0.1, ppo_epoch=2, num_mini_batch=2, max_grad_norm=0.5, num_steps=8,
                      use_normalized_advantage=True, hidden_size=64, lr=2.5e-4, eps=1e-5)
    run_case(ns, "baseline_rgbd96x128", pol, space, cfg, T=8, N=4, seed=61, H=96, W=128)


SE_SEED = 124  # of seeds 0..699 the one whose smallest |pre-ReLU| over the stored minibatch is largest (3.9e-6; `make_golden.py se-seeds`)


# A second fixture of the same network at a seed that was NOT selected for its margin: the selection must not be the only
# thing that makes the deep-encoder comparison pass.  Its stored `mb0_relu_margin` tells the test how close the reference's own
# pre-ReLU activations come to zero (tests/test_gpu_policy.py compares mask-flip-aware: norm-wise bounds upstream of the first such layer).
SE_SEED_UNSELECTED = 7


def se_resnext_case(ns, seed=None, margin_only=False, name="se_resnext50_rgbd128"):
    """se_resneXt50 backbone (resnet.py:92-113,155-193,317-328): grouped 3x3 convolutions (cardinality 16, first
    block of every stage), expansion 2, squeeze-and-excitation gates; 1-layer GRU, 128x128 RGB-D."""
    space = obs_space(ns, 128, 128)
    torch.manual_seed(0)
    pol = ns.resnet_policy.PointNavResNetPolicy(space, ns.spaces.Discrete(4), hidden_size=64, num_recurrent_layers=1, rnn_type="GRU",
                                                backbone="se_resneXt50", normalize_visual_inputs=True)
    cfg = make_config(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, max_grad_norm=0.2, num_steps=3, use_normalized_advantage=False,
                      hidden_size=64, lr=2.5e-4, eps=1e-5)
    return run_case(ns, name, pol, space, cfg, T=3, N=2, seed=SE_SEED if seed is None else seed, H=128, W=128, sampled=True,
                    margin_only=margin_only)


GAUSS_CASE = dict(use_log_std=True, use_softplus=False, log_std_init=0.0, use_std_param=False, clamp_std=True, min_std=1e-6, max_std=1,
                  min_log_std=-5, max_log_std=2, action_activation="tanh")


def gaussian_case(ns):
    """Gaussian action head (GaussianNet + CustomNormal, utils/common.py:99-175) on a Box(2) action space with the
    ResNet policy's Linear previous-action embedding (resnet_policy.py:424-428,754-757) and the adaptive entropy penalty
    (LagrangeInequalityCoefficient, ppo.py:85-103,236-239,373-375): ResNet18 + 1-layer GRU, 128x128 RGB-D."""
    space = obs_space(ns, 128, 128)
    aspace = ns.spaces.Box(-1.0, 1.0, (2,), np.float32)
    pcfg = types.SimpleNamespace(action_distribution_type="gaussian", action_dist=types.SimpleNamespace(**GAUSS_CASE))
    torch.manual_seed(0)
    pol = ns.resnet_policy.PointNavResNetPolicy(space, aspace, hidden_size=64, num_recurrent_layers=1, rnn_type="GRU", backbone="resnet18",
                                                normalize_visual_inputs=True, policy_config=pcfg)
    # PPO only builds the adaptive coefficient for policies that expose `num_actions` (ppo.py:87-90; the hierarchical policies do,
    # NetPolicy does not): expose it, so that the fixture pins the Lagrangian path too
    pol.num_actions = pol.dim_actions
    cfg = make_config(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, max_grad_norm=0.2, num_steps=4, use_normalized_advantage=False,
                      hidden_size=64, lr=2.5e-4, eps=1e-5, entropy_coef=0.01, use_adaptive_entropy_pen=True, entropy_target_factor=0.5)
    run_case(ns, "gaussian_resnet18_rgbd128", pol, space, cfg, T=4, N=2, seed=55, H=128, W=128, sampled=True, num_actions=2,
             action_space=aspace)


if __name__ == "__main__":
    main()
