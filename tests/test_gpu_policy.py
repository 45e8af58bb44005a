"""GPU parity of the policy engine / Python plugin layer against (a) the golden fixtures produced by the
real reference and (b) the CPU oracle, on the same deterministic inputs.  Tolerances: 1e-4 relative on
values / losses / gradients / updated parameters (BASELINE.json), bit-exact sampled actions."""
import json
import os
import types

import numpy as np
import pytest
import torch

from oracle import functional as O
from oracle.fixtures import baseline_param_shapes, det_params, golden_sample, resnet_param_shapes
from test_oracle_golden import CASES, G, is_buffer, make_cfg, oracle_rollout

pytestmark = pytest.mark.gpu
GOAL = "pointgoal_with_gps_compass"


def space_for(c):
    from habitat_amd.common import spaces as S
    d = {}
    if c["rgb"]:
        d["rgb"] = S.Box(0, 255, (c["H"], c["W"], 3), np.uint8)
    if c["depth"]:
        d["depth"] = S.Box(0.0, 1.0, (c["H"], c["W"], 1), np.float32)
    if c.get("task") == "objectnav":
        from oracle import synth
        d["semantic"] = S.Box(0, synth.NUM_SEMANTIC_IDS - 1, (c["H"], c["W"], 1), np.int32)
        d["objectgoal"] = S.Box(0, synth.NUM_OBJECT_CATEGORIES - 1, (1,), np.int64)
        d["compass"] = S.Box(-np.pi, np.pi, (1,), np.float32)
        d["gps"] = S.Box(-1e9, 1e9, (2,), np.float32)
    else:
        d[GOAL] = S.Box(-1e9, 1e9, (2,), np.float32)
    if c.get("gauss"):
        return S.Dict(d), S.Box(-1.0, 1.0, (c["num_actions"],), np.float32)
    return S.Dict(d), S.Discrete(c.get("num_actions", 4))


def extra_of(obs):
    return {k: obs[k] for k in ("semantic", "objectgoal", "compass", "gps") if k in obs}


def build(case, z):
    """Policy (golden parameters) + RolloutStorage filled with the oracle's replay of the golden rollout."""
    from habitat_amd.common.rollout_storage import RolloutStorage
    from habitat_amd.rl.ppo import PointNavBaselinePolicy, PointNavResNetPolicy
    c = CASES[case]
    params, spec, buf, next_value = oracle_rollout(case, z)
    osp, asp = space_for(c)
    if c.get("kind", "baseline") == "resnet":
        rnn_type, rnn_layers = c.get("rnn", ("LSTM", 2))
        pcfg = None
        if c.get("gauss"):  # the ActionDistributionConfig the fixture was generated with (tests/golden/make_golden.py::GAUSS_CASE)
            pcfg = types.SimpleNamespace(action_distribution_type="gaussian",
                                         action_dist=dict(use_log_std=True, use_softplus=False, log_std_init=0.0, use_std_param=False,
                                                          clamp_std=True, min_std=1e-6, max_std=1, min_log_std=-5, max_log_std=2,
                                                          action_activation="tanh"))
        pol = PointNavResNetPolicy(osp, asp, hidden_size=c["hidden"], num_recurrent_layers=rnn_layers, rnn_type=rnn_type,
                                   backbone=c.get("backbone", "resnet18"), normalize_visual_inputs=True, max_frames=c["T"] * c["N"],
                                   max_envs=c["N"], policy_config=pcfg)
        if c.get("lagrange"):
            pol.num_actions = pol.dim_actions  # what makes PPO build the adaptive coefficient (ppo.py:87-90), as in the fixture
    else:
        pol = PointNavBaselinePolicy(osp, asp, hidden_size=c["hidden"], max_frames=c["T"] * c["N"], max_envs=c["N"])
    pol.load_state_dict(params)
    pol.to("cuda")
    st = RolloutStorage(c["T"], c["N"], osp, asp, pol, device="cuda", gae_variant="exact")
    return c, params, spec, buf, next_value, pol, st


def cfg_of(c):
    kw = dict(c["cfg"])
    if c.get("lagrange"):
        kw.update(use_adaptive_entropy_pen=True, entropy_target_factor=-c["lagrange"]["threshold"] / c["num_actions"],
                  entropy_coef=c["lagrange"]["init_alpha"])
    return make_cfg(**kw)


MARGINS = {}  # test id -> largest (error / tolerance) any rel_ok check of that test saw; written to gpurun_out/parity_margins.json


def rel_ok(got, ref, tol=1e-4, floor=1e-3):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref).max() / max(floor, np.abs(ref).max())
    tid = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    MARGINS[tid] = max(MARGINS.get(tid, 0.0), float(err / tol))
    return err <= tol


@pytest.fixture(scope="module", autouse=True)
def _dump_margins():
    yield
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_margins.json"), "w") as f:
        json.dump({"what": "largest observed error / tolerance per test (1.0 = at the limit)", "tests": MARGINS}, f, indent=1)


@pytest.mark.parametrize("case", list(CASES))
def test_rollout_act_matches_reference_golden(case):
    """policy.act through the storage, step by step, with the golden Exp(1) noise: sampled actions must be
    bit-identical to the reference's torch.multinomial draws, values / log-probs / hidden within 1e-4."""
    z = np.load(os.path.join(G, case + ".npz"))
    c, params, spec, buf, next_value, pol, st = build(case, z)
    T, N = c["T"], c["N"]
    dev = "cuda"
    obs_all = {k: v.to(dev) for k, v in buf["observations"].items()}
    st.insert_first_observations({k: v[0] for k, v in obs_all.items()})
    noise = torch.from_numpy(z["exp_noise"]).to(dev)
    pol.eval()
    for t in range(T):
        step = st.get_current_step(slice(0, N), 0)
        ad = pol.act(step["observations"], step["recurrent_hidden_states"], step["prev_actions"], step["masks"],
                     exp_noise=noise[t].contiguous())
        st.insert(next_recurrent_hidden_states=ad.rnn_hidden_states, actions=ad.actions,
                  action_log_probs=ad.action_log_probs, value_preds=ad.values)
        st.insert(next_observations={k: v[t + 1] for k, v in obs_all.items()}, rewards=buf["rewards"][t].to(dev),
                  next_masks=buf["masks"][t + 1].to(dev))
        st.advance_rollout()
    B = st.buffers
    if c.get("gauss"):  # continuous: mu + std * eps with the reference's own N(0, 1) draws
        assert rel_ok(B["actions"].cpu().numpy(), z["roll_actions"], tol=1e-5)
        assert rel_ok(B["prev_actions"].cpu().numpy(), z["roll_prev_actions"], tol=1e-5)
    else:
        assert np.array_equal(B["actions"].cpu().numpy(), z["roll_actions"]), "sampled actions differ from the reference"
        assert np.array_equal(B["prev_actions"].cpu().numpy(), z["roll_prev_actions"])
    assert rel_ok(B["action_log_probs"].cpu().numpy()[:T], z["roll_action_log_probs"][:T])
    assert rel_ok(B["value_preds"].cpu().numpy()[:T], z["roll_value_preds"][:T])
    assert rel_ok(B["recurrent_hidden_states"].cpu().numpy(), z["roll_recurrent_hidden_states"])
    last = st.get_last_step()
    nv = pol.get_value(last["observations"], last["recurrent_hidden_states"], last["prev_actions"], last["masks"])
    assert rel_ok(nv.cpu().numpy(), z["next_value"])
    st.compute_returns(nv, True, 0.99, 0.95)
    assert rel_ok(B["returns"].cpu().numpy()[:T], z["roll_returns"][:T])
    # returns from the reference's own value_preds must be bit-exact with the exact GAE kernel
    B["value_preds"].copy_(torch.from_numpy(z["roll_value_preds"]))
    st.compute_returns(torch.from_numpy(z["next_value"]).to(dev), True, 0.99, 0.95)
    assert np.array_equal(B["returns"].cpu().numpy()[:T], z["roll_returns"][:T])


def fill_storage(st, buf, z, T):
    dev = "cuda"
    B = st.buffers
    for k, v in buf["observations"].items():
        B["observations"][k].copy_(v)
    for k in ("actions", "prev_actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "recurrent_hidden_states"):
        B[k].copy_(torch.from_numpy(z["roll_" + k]))
    st.current_rollout_step_idxs = [T]


@pytest.mark.parametrize("case", list(CASES))
def test_minibatch_forward_loss_backward_vs_reference_golden(case):
    from habitat_amd.rl.ppo import PPO
    z = np.load(os.path.join(G, case + ".npz"))
    c, params, spec, buf, next_value, pol, st = build(case, z)
    cfg = cfg_of(c)
    T, N = c["T"], c["N"]
    fill_storage(st, buf, z, T)
    pol.train()
    ppo = PPO.from_config(pol, cfg)
    adv = ppo.get_advantages(st)
    assert rel_ok(adv.cpu().numpy(), z["advantages"])
    torch.manual_seed(c["seed"] + 1)
    batch = next(st.data_generator(adv, cfg.num_mini_batch))
    eng = pol.engine
    Bn = batch.T * batch.n
    Bf = st.buffers
    obs = Bf["observations"]
    v, lp, ent = (torch.zeros(Bn, device="cuda") for _ in range(3))
    eng.evaluate(obs.get("rgb"), obs.get("depth"), obs.get(GOAL), batch.rows, Bf["recurrent_hidden_states"], Bf["masks"],
                 Bf["actions"], batch.pack, Bn, batch.n, value=v, log_prob=lp, entropy=ent, prev_actions=Bf["prev_actions"],
                 extra=extra_of(obs))
    assert rel_ok(v.cpu().numpy(), z["mb0_value"].reshape(-1))
    assert rel_ok(lp.cpu().numpy(), z["mb0_logp"].reshape(-1))
    assert rel_ok(ent.cpu().numpy(), z["mb0_entropy"].reshape(-1))
    hfin = torch.zeros(batch.n, pol.num_recurrent_layers, c["hidden"], device="cuda")
    eng.final_hidden(hfin)
    assert rel_ok(hfin.cpu().numpy(), z["mb0_hidden"])
    # dict-style (reference-style) access to the lazily gathered batch equals the reference's gather
    assert batch["observations"]["depth"].shape[0] == Bn
    # fused loss + backward
    from habitat_amd import _lib
    import ctypes as C
    dv, dlp, dent = (torch.zeros(Bn, device="cuda") for _ in range(3))
    out = torch.zeros(24, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    if c.get("lagrange"):  # adaptive entropy penalty: the coefficient is the device scalar exp(log_alpha)
        lag = ppo.entropy_coef
        _lib.check(_lib.lib().hab_ppo_loss_ver(P(v), P(lp), P(ent), P(Bf["action_log_probs"]), P(adv), P(Bf["value_preds"]), P(Bf["returns"]),
                                               P(batch.rows), Bn, cfg.clip_param, cfg.value_loss_coef, 0.0, int(cfg.use_clipped_value_loss),
                                               None, None, None, 0, P(lag.log_alpha), lag.threshold, P(dv), P(dlp), P(dent), P(out),
                                               _lib.stream_ptr()))
        assert np.allclose(out[20].item(), z["mb0_grad_log_alpha"], rtol=1e-4)
    else:
        _lib.check(_lib.lib().hab_ppo_loss(P(v), P(lp), P(ent), P(Bf["action_log_probs"]), P(adv), P(Bf["value_preds"]), P(Bf["returns"]),
                                           P(batch.rows), Bn, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef,
                                           int(cfg.use_clipped_value_loss), P(dv), P(dlp), P(dent), P(out), _lib.stream_ptr()))
    assert np.allclose(out[:4].cpu().numpy(), z["mb0_losses"], rtol=1e-4, atol=1e-6)
    eng.backward(obs.get("rgb"), obs.get("depth"), obs.get(GOAL), batch.rows, Bf["actions"], batch.pack, dv, dlp, dent,
                 prev_actions=Bf["prev_actions"], extra=extra_of(obs))
    samp = golden_sample if c.get("sampled") else (lambda a: a)
    bad = []
    for k, g in eng.grad_views.items():
        if is_buffer(k):
            assert float(g.abs().max()) == 0.0  # buffers never receive a gradient
            continue
        ref = z["grad/" + k]
        got = samp(g.cpu().numpy()).reshape(ref.shape)
        if c.get("exact"):  # shallow encoder stored as strided samples + norms: full 1e-4 bar on both
            nr = float(z["gradnorm/" + k])
            nerr = abs(float(g.double().norm()) - nr) / max(1e-12, nr)
            if not rel_ok(got, ref, tol=1e-4, floor=1e-4) or nerr > 1e-4:
                bad.append((k, float(np.abs(got - ref).max()), float(np.abs(ref).max()), nerr))
        elif c.get("sampled"):
            # ReLU boundary: in this fixture the reference's own pre-ReLU activations come within 2.5e-6 of zero
            # (layer3.1.convs.1; all 21 GroupNorm outputs have |y| < 1e-5 somewhere), i.e. inside fp32 round-off of ANY other
            # summation order, so single mask bits legitimately flip and perturb upstream weight gradients by ~1/(pixels).
            # Parameters downstream of the first such layer agree to 1e-5; the rest are bounded norm-wise.  Elementwise
            # gradient parity of this network is pinned by test_resnet_engine_vs_oracle on inputs with a safe margin.
            err = np.linalg.norm((got - ref).astype(np.float64)) / max(1e-12, np.linalg.norm(ref.astype(np.float64)))
            nr = float(z["gradnorm/" + k])
            nerr = abs(float(g.double().norm()) - nr) / max(1e-12, nr)
            downstream = ("compression" in k or "visual_fc" in k or "state_encoder" in k or "_embed" in k
                          or k.startswith("action_") or k.startswith("critic") or (case == "resnet18_rgbd256" and "layer4" in k))
            if case.startswith("gaussian"):
                downstream = "backbone" not in k
            lim = 1e-4 if downstream else 2e-2
            if err > lim or nerr > lim:
                bad.append((k, err, nerr))
        elif not rel_ok(got, ref, tol=1e-4, floor=1e-4):
            bad.append((k, float(np.abs(got - ref).max()), float(np.abs(ref).max())))
    assert not bad, f"gradient mismatch: {bad}"


@pytest.mark.parametrize("case", ["resnet18_rgbd256", "objectnav_resnet50_256"])
def test_resnet_golden_mask_flip_accounting(case):
    """The loosened bound on the deep encoder's upstream weight gradients in the golden test is attributed, not assumed: count the
    ReLU mask bits on which the engine and the (reference-pinned) oracle disagree, overwrite exactly those saved activations with the
    oracle's sign, rerun the backward -- every gradient must then meet the same 2e-4 bound as the parameters downstream of the
    encoder, i.e. the residual of the golden test is the discontinuity of ReLU at pre-activations within fp32 round-off of zero
    and nothing else."""
    from habitat_amd.rl.ppo import PPO
    from habitat_amd import _lib
    import ctypes as C
    z = np.load(os.path.join(G, case + ".npz"))
    c, params, spec, buf, next_value, pol, st = build(case, z)
    cfg = make_cfg(**c["cfg"])
    T, N = c["T"], c["N"]
    fill_storage(st, buf, z, T)
    pol.train()
    ppo = PPO.from_config(pol, cfg)
    adv = ppo.get_advantages(st)
    torch.manual_seed(c["seed"] + 1)
    batch = next(st.data_generator(adv, cfg.num_mini_batch))
    eng = pol.engine
    Bn, Bf = batch.T * batch.n, st.buffers
    obs = Bf["observations"]
    v, lp, ent, dv, dlp, dent = (torch.zeros(Bn, device="cuda") for _ in range(6))
    eng.evaluate(obs.get("rgb"), obs.get("depth"), obs.get(GOAL), batch.rows, Bf["recurrent_hidden_states"], Bf["masks"],
                 Bf["actions"], batch.pack, Bn, batch.n, value=v, log_prob=lp, entropy=ent, prev_actions=Bf["prev_actions"],
                 extra=extra_of(obs))
    out = torch.zeros(16, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().hab_ppo_loss(P(v), P(lp), P(ent), P(Bf["action_log_probs"]), P(adv), P(Bf["value_preds"]), P(Bf["returns"]),
                                       P(batch.rows), Bn, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef,
                                       int(cfg.use_clipped_value_loss), P(dv), P(dlp), P(dent), P(out), _lib.stream_ptr()))

    # the oracle's activations AND gradients for the same minibatch, computed HERE (training-mode RunningMeanAndVar from the same
    # initial statistics).  The comparison below is against these, not against the stored golden gradients: torch-CPU sums in a
    # machine / thread-count dependent order, so the oracle on this host may itself sit on the other side of a ReLU kink than the
    # reference run that produced the fixture (tests/test_oracle_golden.py::test_minibatch_chunked_equals_whole shows one such bit)
    buf_z = dict(buf)  # the rollout as the storage holds it (the reference's own values / returns from the fixture)
    for k in ("actions", "prev_actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "recurrent_hidden_states"):
        buf_z[k] = torch.from_numpy(z["roll_" + k])
    ob = O.gather_minibatch(buf_z, adv.cpu(), batch.inds, T)
    taps = {}
    p_or = {k: (v_.clone().requires_grad_(True) if not is_buffer(k) else v_.clone()) for k, v_ in params.items()}
    ov, olp, oent, _ = O.evaluate_actions(p_or, spec, ob["observations"], ob["recurrent_hidden_states"], ob["prev_actions"], ob["masks"],
                                          ob["actions"], training=True, taps=taps)
    O.ppo_loss(ov, olp, oent, ob, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)[0].backward()

    def backward_and_errors():
        eng.backward(obs.get("rgb"), obs.get("depth"), obs.get(GOAL), batch.rows, Bf["actions"], batch.pack, dv, dlp, dent,
                     prev_actions=Bf["prev_actions"], extra=extra_of(obs))
        worst, vs_golden = {}, {}
        for k, g in eng.grad_views.items():
            if is_buffer(k):
                continue
            ref = p_or[k].grad.double()
            worst[k] = float((g.cpu().double() - ref).norm() / max(1e-30, float(ref.norm())))
            zg = z["grad/" + k]
            got = golden_sample(g.cpu().numpy()).reshape(zg.shape)
            vs_golden[k] = np.linalg.norm((got - zg).astype(np.float64)) / max(1e-12, np.linalg.norm(zg.astype(np.float64)))
        return worst, max(vs_golden.values())

    before, before_golden = backward_and_errors()
    from oracle.parity import resnet_relu_taps
    eng_taps = resnet_relu_taps(eng, c.get("backbone", "resnet18"))
    assert len(eng_taps) == len(taps["relu"])
    flips = {}
    for (name, a), t in zip(taps["relu"], eng_taps):
        ref_act = a.detach().permute(0, 2, 3, 1).contiguous().view(-1).cuda()
        assert ref_act.numel() == t.numel(), name
        diff = (t > 0) != (ref_act > 0)
        nflip = int(diff.sum())
        if nflip:
            flips[name] = (nflip, float(torch.maximum(t, ref_act)[diff].max()))  # how far from zero the disagreeing activations are
            t.copy_(torch.where(ref_act > 0, torch.clamp_min(t, 1e-20), torch.zeros_like(t)))
    rin = eng.tap(3).view(Bn, -1)  # HAB_TAP_RNN_IN: [:, :hidden] = ReLU(visual_fc)
    vfc = taps["visual_fc"].detach().cuda()
    dfc = (rin[:, :c["hidden"]] > 0) != (vfc > 0)
    if int(dfc.sum()):
        flips["visual_fc"] = (int(dfc.sum()), float(torch.maximum(rin[:, :c["hidden"]], vfc)[dfc].max()))
        rin[:, :c["hidden"]] = torch.where(vfc > 0, torch.clamp_min(rin[:, :c["hidden"]], 1e-20), torch.zeros_like(vfc))
    pool_err = float((eng.tap(7) - taps["pool"].detach().permute(0, 2, 3, 1).contiguous().view(-1).cuda()).abs().max())
    after, after_golden = backward_and_errors()
    total = sum(v_[0] for v_ in flips.values())
    n_act = sum(t.numel() for t in eng_taps)
    print(f"[{case}] ReLU mask bits that differ: {total} of {n_act}: {flips}; worst gradient error vs the oracle before {max(before.values()):.2e} "
          f"({max(before, key=before.get)}), after injecting the oracle's mask {max(after.values()):.2e} ({max(after, key=after.get)}); "
          f"vs golden {before_golden:.2e} -> {after_golden:.2e}; max-pool output max |err| {pool_err:.1e}")
    rep = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(rep, exist_ok=True)
    with open(os.path.join(rep, f"mask_flips_{case}.txt"), "w") as f:
        f.write(f"differing ReLU mask bits: {total} of {n_act}\n{flips}\nworst gradient error (norm-wise, engine vs the oracle on this host) "
                f"before: {max(before.values()):.3e} after injecting the oracle's mask: {max(after.values()):.3e}\n"
                f"(vs the stored reference golden, sampled: before {before_golden:.3e} after {after_golden:.3e})\n"
                f"max-pool output max |err| {pool_err:.2e}\n\nper parameter (before, after):\n")
        for k in before:
            f.write(f"  {before[k]:.3e} {after[k]:.3e}  {k}\n")
    assert all(mag < 1e-4 for _, mag in flips.values()), flips  # only activations within round-off of zero may disagree
    assert total <= 1e-5 * n_act, (total, n_act)
    bad = {k: e for k, e in after.items() if e > 2e-4}
    assert not bad, (bad, flips)


# hab_set_matrix_path masks (include/habitat_amd.h): everything on the fp32 MFMA kernels (0); the plain split-bf16 contraction alone (1:
# weight gradients, the observation convolution, every patch / strip kernel on their fp32 forms); the default minus the three kernels
# hard-wired to the 256 x 256 benchmark geometry (observation patch, conv2 forward / data-gradient strips): what an observation size
# other than 256 x 256 runs in production
MATRIX_PATHS = {"fp32_mfma": 0, "split_bf16_igemm_only": 1, "no_256x256_strip_kernels": 1023 & ~(64 | 256 | 512), "no_dense_gemm": 1023}


@pytest.mark.parametrize("path", list(MATRIX_PATHS))
@pytest.mark.parametrize("case", ["baseline_rgbd44", "c1_depth84_h512_4x32", "resnet18_rgbd256"])
def test_full_ppo_update_vs_reference_golden_on_every_matrix_path(case, path, monkeypatch):
    """The golden update through the ENGINE with the kernel-selection mask of hab_set_matrix_path off its default: the fp32 MFMA path
    and the im2col fallbacks stay pinned to the reference at engine level, not just kernel by kernel."""
    from habitat_amd import _lib
    L = _lib.lib()
    prev = L.hab_set_matrix_path(MATRIX_PATHS[path])
    try:
        test_full_ppo_update_vs_reference_golden(case, monkeypatch)
    finally:
        L.hab_set_matrix_path(prev)


@pytest.mark.parametrize("case", list(CASES))
def test_full_ppo_update_vs_reference_golden(case, monkeypatch):
    """PPO.update on the golden rollout with the golden minibatch permutations: learner metrics and every
    parameter after all Adam steps vs the reference."""
    from habitat_amd.rl.ppo import PPO
    z = np.load(os.path.join(G, case + ".npz"))
    c, params, spec, buf, next_value, pol, st = build(case, z)
    cfg = cfg_of(c)
    T, N = c["T"], c["N"]
    fill_storage(st, buf, z, T)
    pol.train()
    ppo = PPO.from_config(pol, cfg)
    if c.get("kind") == "resnet":
        # make_golden.py evaluated minibatch 0 in training mode before PPO.update: RunningMeanAndVar saw that batch once more
        torch.manual_seed(c["seed"] + 1)
        b0 = next(st.data_generator(ppo.get_advantages(st), cfg.num_mini_batch))
        Bf, obs = st.buffers, st.buffers["observations"]
        pol.engine.evaluate(obs.get("rgb"), obs.get("depth"), obs.get(GOAL), b0.rows, Bf["recurrent_hidden_states"], Bf["masks"],
                            Bf["actions"], b0.pack, b0.T * b0.n, b0.n, prev_actions=Bf["prev_actions"], extra=extra_of(obs))
    perms = [torch.from_numpy(p) for p in z["perms"]]
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perms.pop(0))
    metrics = ppo.update(st)
    for k, val in metrics.items():
        ref = float(z["metric/" + k])
        # losses: 1e-4 (BASELINE.json).  In the deep-encoder fixture the remaining learner statistics are taken after Adam
        # steps driven by gradients that contain legitimate ReLU-boundary flips (see the minibatch test): 1e-3 there.
        # (observed: <= 1.4e-3 of the statistic's own range -- value_pred_min of the ObjectNav fixture, predictions spanning +-0.03 --
        # everything else <= 5e-4; profiles/r03_parity_margins.json)
        tol = 1e-4 if (not c.get("sampled") or c.get("exact") or k in ("value_loss", "action_loss", "dist_entropy")) else 2e-3
        # true relative error (floor 1e-6 for statistics that are themselves ~0, e.g. ppo_fraction_clipped = 0).  min / mean / max of one
        # quantity are measured against that quantity's range: value_pred_min = -0.024 of predictions spanning [-0.02, 0.5] has an
        # absolute error of the predictions' scale, not of its own distance from zero
        fam = k.rsplit("_", 1)[0] if k.rsplit("_", 1)[-1] in ("min", "mean", "max") else None
        scale = max(abs(float(z["metric/" + q])) for q in metrics if fam and q.startswith(fam + "_")) if fam else abs(ref)
        MARGINS[os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + ":" + k] = abs(val - ref) / max(tol * scale, 1e-6)
        assert abs(val - ref) <= max(tol * scale, 1e-6), (k, val, ref, scale)
    samp = golden_sample if c.get("sampled") else (lambda a: a)
    for k, v in pol.state_dict().items():
        ref = z["post/" + k]
        got = samp(v.cpu().numpy()).reshape(ref.shape)
        # sampled (deep GroupNorm encoder) case: Adam turns a relative gradient perturbation into a fraction of lr per step;
        # 4 steps x lr 2.5e-4 bounds the drift by 1e-3, observed < 3e-4
        tol = 5e-4 * max(1.0, np.abs(ref).max()) if (c.get("sampled") and not c.get("exact")) else 1e-4 * max(1e-2, np.abs(ref).max())
        assert np.abs(got - ref).max() <= tol, k
    if c.get("lagrange"):
        assert "entropy_coef" in metrics
        assert abs(float(ppo.entropy_coef.log_alpha) - float(z["post_log_alpha"])) <= 1e-5


def test_autograd_bridge_matches_fused_path():
    """Reference-style usage: evaluate_actions on dense tensors + torch loss + loss.backward()."""
    case = "baseline_rgbd44"
    z = np.load(os.path.join(G, case + ".npz"))
    c, params, spec, buf, next_value, pol, st = build(case, z)
    cfg = make_cfg(**c["cfg"])
    T, N = c["T"], c["N"]
    fill_storage(st, buf, z, T)
    adv = torch.from_numpy(z["advantages"]).cuda()
    torch.manual_seed(c["seed"] + 1)
    batch = next(st.data_generator(adv, cfg.num_mini_batch))
    pol.train()
    v, lp, ent, h, _ = pol.evaluate_actions(batch["observations"], batch["recurrent_hidden_states"], batch["prev_actions"],
                                            batch["masks"], batch["actions"], batch["rnn_build_seq_info"])
    b = {k: batch[k] for k in ("action_log_probs", "advantages", "value_preds", "returns")}
    total, vl, al, de, _ = O.ppo_loss(v, lp, ent, b, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)
    for p_ in pol.parameters():
        p_.grad = None
    total.backward()
    assert np.allclose(np.array([vl.item(), al.item(), de.item(), total.item()]), z["mb0_losses"], rtol=1e-4, atol=1e-6)
    for k, p_ in pol.named_parameters():
        ref = z["grad/" + k]
        assert rel_ok(p_.grad.cpu().numpy(), ref, tol=1e-4, floor=1e-4), k


# (2 and 3 layers run the packed recurrence as a layer wavefront, csrc/rnn.hip: 64 -> 4 waves per workgroup, 128 / 512 -> 8)
@pytest.mark.parametrize("rnn_type,layers,hidden", [("LSTM", 2, 64), ("GRU", 2, 64), ("GRU", 1, 512), ("LSTM", 1, 256), ("LSTM", 3, 128),
                                                    ("GRU", 3, 64)])
def test_engine_lstm_gru_multilayer_vs_oracle(rnn_type, layers, hidden):
    """The engine also runs LSTM / multi-layer encoders on the baseline net; checked against the oracle's
    masked-scan restatement incl. all gradients (autograd on the oracle).  hidden 512 / 256 = the benchmark width: the
    8-wave recurrent kernels and the vector-load heads kernel only exist for hidden % 128 / % 256 == 0."""
    from habitat_amd.engine import DevicePackInfo, PolicyEngine
    H = W = 44
    T, n = 7, 3
    shapes = baseline_param_shapes(4, H, W, hidden, rnn_type=rnn_type, layers=layers)
    params = det_params(shapes, 11)
    eng = PolicyEngine(arch="simple_cnn", rnn_type=rnn_type, rnn_layers=layers, hidden=hidden, H=H, W=W, max_frames=T * n, max_envs=n)
    eng.load({k: v.cuda() for k, v in params.items()})
    rng = np.random.default_rng(0)
    B = T * n
    rgb = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8))
    depth = torch.from_numpy(rng.random((B, H, W, 1), dtype=np.float32))
    goal = torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32))
    masks = torch.from_numpy(rng.random((B, 1)) > 0.25)
    actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
    Lh = layers * (2 if rnn_type == "LSTM" else 1)
    h0 = torch.from_numpy(rng.standard_normal((n, Lh, hidden)).astype(np.float32))
    spec = O.NetSpec(kind="baseline", rnn_type=rnn_type, num_layers=layers, hidden=hidden)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    obs = {"rgb": rgb, "depth": depth, GOAL: goal}
    v, lp, ent, hfin = O.evaluate_actions(p, spec, obs, h0, torch.zeros(B, 1, dtype=torch.long), masks, actions)
    gv, glp, gent = (torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32)) for _ in range(3))
    ((v * gv).sum() + (lp * glp).sum() + (ent * gent).sum()).backward()
    pack = DevicePackInfo(np.logical_not(masks.view(T, n).numpy()), "cuda")
    dv, dl, de = (torch.zeros(B, device="cuda") for _ in range(3))
    eng.evaluate(rgb.cuda(), depth.cuda(), goal.cuda(), None, h0.cuda(), masks.cuda(), actions.cuda(), pack, B, n, value=dv, log_prob=dl, entropy=de)
    assert rel_ok(dv.cpu().numpy(), v.detach().numpy().reshape(-1))
    assert rel_ok(dl.cpu().numpy(), lp.detach().numpy().reshape(-1))
    assert rel_ok(de.cpu().numpy(), ent.detach().numpy().reshape(-1))
    hf = torch.zeros(n, Lh, hidden, device="cuda")
    eng.final_hidden(hf)
    assert rel_ok(hf.cpu().numpy(), hfin.detach().numpy())
    eng.backward(rgb.cuda(), depth.cuda(), goal.cuda(), None, actions.cuda(), pack, gv.view(-1).cuda(), glp.view(-1).cuda(), gent.view(-1).cuda())
    bad = [(k, float((g.cpu() - p[k].grad).abs().max()), float(p[k].grad.abs().max())) for k, g in eng.grad_views.items()
           if not rel_ok(g.cpu().numpy(), p[k].grad.numpy(), tol=1e-4, floor=1e-4)]
    assert not bad, bad


def test_blind_baseline_policy_vs_oracle():
    """PointNavBaselinePolicy without a visual sensor (SimpleCNN.is_blind: goal -> GRU -> heads; the reference's own DD-PPO test builds
    exactly this, test/test_ddppo_reduce.py:43-56, with Discrete(1)): act, evaluate, every gradient against the oracle."""
    from habitat_amd.common import spaces as S
    from habitat_amd.engine import DevicePackInfo
    from habitat_amd.rl.ppo import PointNavBaselinePolicy
    hidden, T, n = 64, 9, 3
    B = T * n
    for goal_key, n_act in ((GOAL, 1), ("pointgoal", 4)):
        torch.manual_seed(21)
        pol = PointNavBaselinePolicy(S.Dict({goal_key: S.Box(-1e9, 1e9, (2,), np.float32)}), S.Discrete(n_act), hidden_size=hidden,
                                     max_frames=B, max_envs=n)
        params = {k: v.detach().clone() for k, v in pol.state_dict().items()}
        assert not any("visual_encoder" in k for k in params)
        pol.to("cuda")
        pol.train()
        rng = np.random.default_rng(4)
        goal = torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32))
        masks = torch.from_numpy(rng.random((B, 1)) > 0.25)
        actions = torch.from_numpy(rng.integers(0, n_act, (B, 1)))
        h0 = torch.from_numpy(rng.standard_normal((n, 1, hidden)).astype(np.float32))
        spec = O.NetSpec(kind="baseline", hidden=hidden, num_actions=n_act)
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        v, lp, ent, hfin = O.evaluate_actions(p, spec, {goal_key: goal}, h0, torch.zeros(B, 1, dtype=torch.long), masks, actions)
        gv, glp, gent = (torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32)) for _ in range(3))
        ((v * gv).sum() + (lp * glp).sum() + (ent * gent).sum()).backward()
        eng = pol.engine
        pack = DevicePackInfo(np.logical_not(masks.view(T, n).numpy()), "cuda")
        dv, dl, de = (torch.zeros(B, device="cuda") for _ in range(3))
        eng.evaluate(None, None, goal.cuda(), None, h0.cuda(), masks.cuda(), actions.cuda(), pack, B, n, value=dv, log_prob=dl, entropy=de)
        assert rel_ok(dv.cpu().numpy(), v.detach().numpy().reshape(-1))
        assert rel_ok(dl.cpu().numpy(), lp.detach().numpy().reshape(-1), floor=1e-6)
        assert rel_ok(de.cpu().numpy(), ent.detach().numpy().reshape(-1), floor=1e-6)
        eng.backward(None, None, goal.cuda(), None, actions.cuda(), pack, gv.view(-1).cuda(), glp.view(-1).cuda(), gent.view(-1).cuda())
        bad = [(k, float((g.cpu() - p[k].grad).abs().max()), float(p[k].grad.abs().max())) for k, g in eng.grad_views.items()
               if not rel_ok(g.cpu().numpy(), p[k].grad.numpy(), tol=1e-4, floor=1e-4)]
        assert not bad, bad
        # act on n envs through the plugin surface: same sampled actions as the oracle given the same Exp(1) noise
        pol.eval()
        noise = torch.from_numpy(rng.exponential(1.0, (n, n_act)).astype(np.float32))
        ad = pol.act({goal_key: goal[:n].cuda()}, h0.cuda(), torch.zeros(n, 1, dtype=torch.long, device="cuda"), masks[:n].cuda(),
                     exp_noise=noise.cuda())
        with torch.no_grad():
            ref = O.act({k: v.detach() for k, v in p.items()}, spec, {goal_key: goal[:n]}, h0, torch.zeros(n, 1, dtype=torch.long), masks[:n],
                        exp_noise=noise)
        assert torch.equal(ad.actions.cpu(), ref["actions"])
        assert rel_ok(ad.values.cpu().numpy(), ref["values"].numpy())


def test_resnet_policy_pointgoal_and_proximity_embeddings_vs_oracle():
    """PointNavResNetNet's PointGoalSensor / ProximitySensor inputs (resnet_policy.py:489-515,694-700) beside the polar goal, gps and
    compass: values / log-probs and the gradients of all six embeddings against the oracle."""
    from habitat_amd.common import spaces as S
    from habitat_amd.engine import DevicePackInfo
    from habitat_amd.rl.ppo import PointNavResNetPolicy
    hidden, T, n, H, W = 64, 3, 2, 64, 64
    B = T * n
    box = lambda d: S.Box(-1e9, 1e9, (d,), np.float32)
    osp = S.Dict({"depth": S.Box(0.0, 1.0, (H, W, 1), np.float32), GOAL: box(2), "pointgoal": box(2), "proximity": box(1), "gps": box(2),
                  "compass": box(1)})
    torch.manual_seed(8)
    pol = PointNavResNetPolicy(osp, S.Discrete(4), hidden_size=hidden, backbone="resnet18", max_frames=B, max_envs=n)
    params = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    pol.to("cuda")
    pol.train()
    rng = np.random.default_rng(5)
    f32 = lambda *shape: torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    obs = {"depth": torch.from_numpy(rng.random((B, H, W, 1), dtype=np.float32)), GOAL: f32(B, 2).abs(), "pointgoal": f32(B, 2),
           "proximity": f32(B, 1).abs(), "gps": f32(B, 2), "compass": f32(B, 1)}
    masks = torch.from_numpy(rng.random((B, 1)) > 0.3)
    actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
    prev_actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
    h0 = f32(n, 1, hidden)
    spec = O.NetSpec(kind="resnet", rnn_type="GRU", num_layers=1, backbone="resnet18", baseplanes=32, visual_keys=("depth",), normalize=False,
                     hidden=hidden)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    v, lp, ent, _ = O.evaluate_actions(p, spec, obs, h0, prev_actions, masks, actions, training=True)
    gv, glp, gent = (f32(B, 1) for _ in range(3))
    ((v * gv).sum() + (lp * glp).sum() + (ent * gent).sum()).backward()
    eng = pol.engine
    pack = DevicePackInfo(np.logical_not(masks.view(T, n).numpy()), "cuda")
    dv, dl, de = (torch.zeros(B, device="cuda") for _ in range(3))
    extra = {k: obs[k].cuda() for k in ("pointgoal", "proximity", "gps", "compass")}
    eng.evaluate(None, obs["depth"].cuda(), obs[GOAL].cuda(), None, h0.cuda(), masks.cuda(), actions.cuda(), pack, B, n, value=dv, log_prob=dl,
                 entropy=de, prev_actions=prev_actions.cuda(), extra=extra)
    assert rel_ok(dv.cpu().numpy(), v.detach().numpy().reshape(-1))
    assert rel_ok(dl.cpu().numpy(), lp.detach().numpy().reshape(-1))
    eng.backward(None, obs["depth"].cuda(), obs[GOAL].cuda(), None, actions.cuda(), pack, gv.view(-1).cuda(), glp.view(-1).cuda(),
                 gent.view(-1).cuda(), prev_actions=prev_actions.cuda(), extra=extra)
    for k, g in eng.grad_views.items():
        if "_embed" in k:
            assert rel_ok(g.cpu().numpy(), p[k].grad.numpy(), tol=1e-4, floor=1e-5), k


@pytest.mark.parametrize("rnn_type,layers,force", [("LSTM", 2, True), ("GRU", 1, False)])
def test_blind_resnet_policy_vs_oracle(rnn_type, layers, force):
    """PointNavResNetPolicy with `force_blind_policy` (images in the observation space, ignored: resnet_policy.py:553-554) and with an
    observation space that has no image at all: the net is embeddings -> GRU / LSTM -> heads (no backbone, compression, visual_fc).
    evaluate, every gradient and act against the oracle (whose blind forward is pinned to the live reference on CPU:
    test_blind_resnet_policy_identical_to_live_reference); the layer wavefront of the 2-layer LSTM runs without the ReLU(fc) mask."""
    from habitat_amd.common import spaces as S
    from habitat_amd.engine import DevicePackInfo
    from habitat_amd.rl.ppo import PointNavResNetPolicy
    hidden, T, n, H, W = 64, 9, 3, 64, 64
    B = T * n
    box = lambda d: S.Box(-1e9, 1e9, (d,), np.float32)
    sp = {GOAL: box(2), "gps": box(2), "compass": box(1), "objectgoal": S.Box(0, 5, (1,), np.int64)}
    if force:
        sp = dict({"depth": S.Box(0.0, 1.0, (H, W, 1), np.float32)}, **sp)
    torch.manual_seed(31)
    pol = PointNavResNetPolicy(S.Dict(sp), S.Discrete(4), hidden_size=hidden, rnn_type=rnn_type, num_recurrent_layers=layers,
                               backbone="resnet18", force_blind_policy=force, max_frames=B, max_envs=n)
    params = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    assert pol.is_blind and not any("visual" in k for k in params)
    pol.to("cuda")
    pol.train()
    rng = np.random.default_rng(6)
    f32 = lambda *shape: torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    obs = {GOAL: f32(B, 2).abs(), "gps": f32(B, 2), "compass": f32(B, 1), "objectgoal": torch.from_numpy(rng.integers(0, 6, (B, 1)))}
    masks = torch.from_numpy(rng.random((B, 1)) > 0.3)
    actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
    prev_actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
    Lh = layers * (2 if rnn_type == "LSTM" else 1)
    h0 = f32(n, Lh, hidden)
    spec = O.NetSpec(kind="resnet", rnn_type=rnn_type, num_layers=layers, visual_keys=(), normalize=False, hidden=hidden)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    v, lp, ent, _ = O.evaluate_actions(p, spec, obs, h0, prev_actions, masks, actions, training=True)
    gv, glp, gent = (f32(B, 1) for _ in range(3))
    ((v * gv).sum() + (lp * glp).sum() + (ent * gent).sum()).backward()
    eng = pol.engine
    pack = DevicePackInfo(np.logical_not(masks.view(T, n).numpy()), "cuda")
    dv, dl, de = (torch.zeros(B, device="cuda") for _ in range(3))
    extra = {k: obs[k].cuda() for k in ("gps", "compass", "objectgoal")}
    eng.evaluate(None, None, obs[GOAL].cuda(), None, h0.cuda(), masks.cuda(), actions.cuda(), pack, B, n, value=dv, log_prob=dl, entropy=de,
                 prev_actions=prev_actions.cuda(), extra=extra)
    assert rel_ok(dv.cpu().numpy(), v.detach().numpy().reshape(-1))
    assert rel_ok(dl.cpu().numpy(), lp.detach().numpy().reshape(-1), floor=1e-6)
    assert rel_ok(de.cpu().numpy(), ent.detach().numpy().reshape(-1), floor=1e-6)
    eng.backward(None, None, obs[GOAL].cuda(), None, actions.cuda(), pack, gv.view(-1).cuda(), glp.view(-1).cuda(), gent.view(-1).cuda(),
                 prev_actions=prev_actions.cuda(), extra=extra)
    assert set(eng.grad_views) == set(p)
    bad = [(k, float((g.cpu() - p[k].grad).abs().max()), float(p[k].grad.abs().max())) for k, g in eng.grad_views.items()
           if not rel_ok(g.cpu().numpy(), p[k].grad.numpy(), tol=1e-4, floor=1e-4)]
    assert not bad, bad
    # act through the plugin surface (the observation dict may still carry the images: they are ignored)
    pol.eval()
    noise = torch.from_numpy(rng.exponential(1.0, (n, 4)).astype(np.float32))
    o_n = {k: t[:n].cuda() for k, t in obs.items()}
    if force:
        o_n["depth"] = torch.rand(n, H, W, 1, device="cuda")
    ad = pol.act(o_n, h0.cuda(), prev_actions[:n].cuda(), masks[:n].cuda(), exp_noise=noise.cuda())
    with torch.no_grad():
        ref = O.act({k: t.detach() for k, t in p.items()}, spec, {k: t[:n] for k, t in obs.items()}, h0, prev_actions[:n], masks[:n], exp_noise=noise)
    assert torch.equal(ad.actions.cpu(), ref["actions"])
    assert rel_ok(ad.values.cpu().numpy(), ref["values"].numpy())
    assert rel_ok(ad.rnn_hidden_states.cpu().numpy(), ref["rnn_hidden_states"].numpy())


RESNET_VARIANTS = [  # backbone, rnn, layers, H, W, visual key order, normalize
    ("resnet18", "GRU", 1, 128, 128, ("depth", "rgb"), True),
    ("resnet18", "LSTM", 2, 64, 96, ("rgb", "depth"), False),
    ("resnet50", "LSTM", 1, 128, 128, ("rgb", "depth"), True),
    ("resnet18", "GRU", 1, 128, 128, ("depth",), False),
    # SURVEY.md 8f N3 (resnet.py:296-345): grouped 3x3 convolutions (ResNeXt), squeeze-and-excitation gates, both, 23-block stage 3
    ("resneXt50", "GRU", 1, 128, 128, ("rgb", "depth"), True),
    ("se_resnet50", "GRU", 1, 128, 128, ("rgb", "depth"), False),
    ("se_resneXt50", "LSTM", 1, 128, 128, ("rgb", "depth"), True),
    ("se_resneXt101", "GRU", 1, 64, 64, ("rgb", "depth"), False),
    # odd / non-square observation sizes of the reference's test/test_baseline_resnet.py:22-73 (the 2x2 average pool floors,
    # the stem's padded 4-channel input, conv_patch_bf3's TW selection with Wo < 32, chunked GroupNorm at odd extents)
    ("resnet18", "GRU", 1, 62, 30, ("rgb", "depth"), True),
    ("resnet18", "LSTM", 2, 63, 84, ("rgb", "depth"), False),
    ("resnet50", "GRU", 1, 65, 30, ("rgb", "depth"), True),
    ("resnet18", "GRU", 1, 66, 64, ("depth",), False),
    ("resnet50", "GRU", 1, 64, 128, ("rgb",), True),
]


def test_resnet_geometry_with_unaligned_compression_width_is_refused_loudly():
    """100 x 180: final feature map 2 x 3 -> round(2048 / 6) = 341 compression channels (resnet_policy.py:222-228), not a multiple of the
    4-channel NHWC vector the kernels move -- the engine must refuse at construction (no silent fallback), and say so."""
    from habitat_amd._lib import HabError
    from habitat_amd.common import spaces as S
    from habitat_amd.rl.ppo import PointNavResNetPolicy
    osp = S.Dict({"rgb": S.Box(0, 255, (100, 180, 3), np.uint8), "depth": S.Box(0.0, 1.0, (100, 180, 1), np.float32),
                  GOAL: S.Box(-1e9, 1e9, (2,), np.float32)})
    pol = PointNavResNetPolicy(osp, S.Discrete(4), hidden_size=64, backbone="resnet18", max_frames=4, max_envs=2)
    with pytest.raises(HabError, match="unsupported"):
        pol.to("cuda")


@pytest.mark.parametrize("backbone,rnn_type,layers,H,W,keys,normalize", RESNET_VARIANTS)
def test_resnet_engine_vs_oracle(backbone, rnn_type, layers, H, W, keys, normalize):
    """PointNavResNetPolicy variants (BasicBlock / Bottleneck, visual key orders, RunningMeanAndVar on/off) against the
    oracle: intermediate activations, outputs, updated running statistics and every parameter gradient."""
    from habitat_amd.common import spaces as S
    from habitat_amd.engine import DevicePackInfo
    from habitat_amd.rl.ppo import PointNavResNetPolicy
    hidden, T, n = 64, 3, 2
    B = T * n
    d = {}
    for k in keys:
        d[k] = S.Box(0, 255, (H, W, 3), np.uint8) if k == "rgb" else S.Box(0.0, 1.0, (H, W, 1), np.float32)
    d[GOAL] = S.Box(-1e9, 1e9, (2,), np.float32)
    osp, asp = S.Dict(d), S.Discrete(4)
    n_in = sum(3 if k == "rgb" else 1 for k in keys)
    params = det_params(resnet_param_shapes(n_in, H, W, hidden, rnn_type=rnn_type, layers=layers, backbone=backbone), 31)
    pre = "net.visual_encoder.running_mean_and_var."
    if normalize:
        params[pre + "_mean"], params[pre + "_var"], params[pre + "_count"] = (
            torch.full((1, n_in, 1, 1), 0.3), torch.full((1, n_in, 1, 1), 0.05), torch.tensor(6.0))
    pol = PointNavResNetPolicy(osp, asp, hidden_size=hidden, num_recurrent_layers=layers, rnn_type=rnn_type, backbone=backbone,
                               normalize_visual_inputs=normalize, max_frames=B, max_envs=n)
    assert list(pol.state_dict().keys()) == [k for k, _ in resnet_param_shapes(n_in, H, W, hidden, rnn_type=rnn_type, layers=layers,
                                                                              backbone=backbone, normalize=normalize, with_buffers=True)]
    pol.load_state_dict(params)
    pol.to("cuda")
    pol.train()
    eng = pol.engine
    Lh = layers * (2 if rnn_type == "LSTM" else 1)
    spec = O.NetSpec(kind="resnet", rnn_type=rnn_type, num_layers=layers, backbone=backbone, baseplanes=32, visual_keys=keys,
                     normalize=normalize, hidden=hidden)

    def make_inputs(seed):
        rng = np.random.default_rng(seed)
        obs = {}
        if "rgb" in keys:
            obs["rgb"] = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8))
        if "depth" in keys:
            obs["depth"] = torch.from_numpy(rng.random((B, H, W, 1), dtype=np.float32))
        obs[GOAL] = torch.from_numpy(np.stack([rng.random(B) * 5, rng.uniform(-3.1, 3.1, B)], 1).astype(np.float32))
        masks = torch.from_numpy(rng.random((B, 1)) > 0.3)
        actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
        prev_actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
        h0 = torch.from_numpy(rng.standard_normal((n, Lh, hidden)).astype(np.float32))
        return rng, obs, masks, actions, prev_actions, h0

    # ReLU is discontinuous in its gradient: a pre-activation within fp32 round-off (~1e-6 after 20 layers) of zero may land on
    # either side in two correct implementations.  Elementwise gradient parity is therefore checked on the first input draw
    # whose smallest |pre-ReLU| value in the oracle clears that band with a margin.
    import torch.nn.functional as F
    orig_relu = F.relu
    best = None
    for seed in range(1, 61):
        rng, obs, masks, actions, prev_actions, h0 = make_inputs(seed)
        margin = [np.inf]

        def relu_probe(x, inplace=False):
            margin[0] = min(margin[0], float(x.detach().abs().min()))
            return orig_relu(x)

        F.relu = relu_probe
        try:
            with torch.no_grad():
                O.evaluate_actions(params, spec, obs, h0, prev_actions, masks, actions, training=True)
        finally:
            F.relu = orig_relu
        if best is None or margin[0] > best[0]:
            best = (margin[0], seed)
        if margin[0] > 2e-5:
            break
    rng, obs, masks, actions, prev_actions, h0 = make_inputs(best[1])
    p = {k: (v.clone().requires_grad_(True) if not is_buffer(k) else v.clone()) for k, v in params.items()}
    taps, rmv = {}, {}
    v, lp, ent, hfin = O.evaluate_actions(p, spec, obs, h0, prev_actions, masks, actions, training=True, taps=taps, rmv_out=rmv)
    gv, glp, gent = (torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32)) for _ in range(3))
    ((v * gv).sum() + (lp * glp).sum() + (ent * gent).sum()).backward()
    pack = DevicePackInfo(np.logical_not(masks.view(T, n).numpy()), "cuda")
    dv, dl, de = (torch.zeros(B, device="cuda") for _ in range(3))
    cu = lambda t: t.cuda() if t is not None else None
    eng.evaluate(cu(obs.get("rgb")), cu(obs.get("depth")), obs[GOAL].cuda(), None, h0.cuda(), masks.cuda(), actions.cuda(), pack, B, n,
                 value=dv, log_prob=dl, entropy=de, prev_actions=prev_actions.cuda())
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().numpy()
    x0 = eng.tap(5).cpu().numpy().reshape(B, H // 2, W // 2, 4)[..., :n_in]
    assert rel_ok(x0, nhwc(taps["enc_in"])), "encoder input (ingest + RunningMeanAndVar)"
    for tap_id, name in ((6, "stem"), (7, "pool"), (9, "layer1"), (10, "layer2"), (11, "layer3"), (12, "layer4"), (8, "compression")):
        ref = nhwc(taps[name])
        got = eng.tap(tap_id).cpu().numpy().reshape(ref.shape)
        assert rel_ok(got, ref, tol=1e-4), name
    assert rel_ok(eng.tap(3).cpu().numpy().reshape(B, -1)[:, :hidden + 64], taps["rnn_in"].detach().numpy(), tol=1e-4), "rnn_in"
    assert rel_ok(dv.cpu().numpy(), v.detach().numpy().reshape(-1), tol=1e-4)
    assert rel_ok(dl.cpu().numpy(), lp.detach().numpy().reshape(-1), tol=1e-4)
    assert rel_ok(de.cpu().numpy(), ent.detach().numpy().reshape(-1), tol=1e-4)
    hf = torch.zeros(n, Lh, hidden, device="cuda")
    eng.final_hidden(hf)
    assert rel_ok(hf.cpu().numpy(), hfin.detach().numpy(), tol=1e-4)
    if normalize:
        sd = pol.state_dict()
        for k in ("mean", "var", "count"):
            assert rel_ok(sd[pre + "_" + k].cpu().numpy(), rmv[k].numpy(), tol=1e-5), k
    eng.backward(cu(obs.get("rgb")), cu(obs.get("depth")), obs[GOAL].cuda(), None, actions.cuda(), pack, gv.view(-1).cuda(),
                 glp.view(-1).cuda(), gent.view(-1).cuda(), prev_actions=prev_actions.cuda())
    bad = [(k, float((g.cpu() - p[k].grad).abs().max()), float(p[k].grad.abs().max())) for k, g in eng.grad_views.items()
           if not is_buffer(k) and not rel_ok(g.cpu().numpy(), p[k].grad.numpy(), tol=1e-4, floor=1e-4)]
    assert not bad, bad
    # eval mode: statistics frozen, act() on n envs equals the oracle
    pol.eval()
    before = {k: v.clone() for k, v in pol.state_dict().items() if is_buffer(k)}
    o1 = {k: v[:n].cuda().contiguous() for k, v in obs.items()}
    noise = torch.from_numpy(rng.exponential(1.0, (n, 4)).astype(np.float32))
    ad = pol.act(o1, h0.cuda(), prev_actions[:n].cuda(), masks[:n].cuda(), exp_noise=noise.cuda())
    pp = {k: v.detach() for k, v in p.items()}
    pp.update({pre + "_" + k: val for k, val in rmv.items()})
    with torch.no_grad():
        ref = O.act(pp, spec, {k: v[:n] for k, v in obs.items()}, h0, prev_actions[:n], masks[:n], exp_noise=noise)
    assert torch.equal(ad.actions.cpu(), ref["actions"])
    assert rel_ok(ad.values.cpu().numpy(), ref["values"].numpy(), tol=1e-4)
    assert rel_ok(ad.rnn_hidden_states.cpu().numpy(), ref["rnn_hidden_states"].numpy(), tol=1e-4)
    for k, v0 in before.items():
        assert torch.equal(pol.state_dict()[k], v0), "RunningMeanAndVar must not change in eval mode"


@pytest.mark.parametrize("yaml_path,overrides", [
    ("pointnav/ppo_pointnav_habitat_iccv19.yaml", []),
    ("pointnav/ddppo_pointnav.yaml", ["habitat_baselines.rl.ddppo.backbone=resnet18"]),
    ("objectnav/ddppo_objectnav.yaml", []),
])
def test_trainer_update_cycles_from_yaml_entrypoints(yaml_path, overrides):
    """The registered trainer built from the YAML entrypoints runs full cycles (device rollout through the synthetic env source ->
    GAE -> PPO update) for the three experiment presets at a reduced size: finite losses, step accounting, parameters move."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401  (registers the trainer, policies, updaters, storage)
    size = 128
    ov = ["habitat_baselines.num_environments=4", "habitat_baselines.rl.ppo.num_steps=8", "habitat_baselines.num_updates=3",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", "habitat_baselines.checkpoint_folder=/tmp/habitat_amd_test_ckpt",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000"]
    for sname in ("rgb", "depth", "semantic"):
        ov += [f"habitat.simulator.sensors.{sname}.height={size}", f"habitat.simulator.sensors.{sname}.width={size}"]
    cfg = get_config(yaml_path, ov + overrides)
    if "objectnav" not in yaml_path:
        cfg.habitat.simulator.sensors.pop("semantic", None)
    trainer = baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)
    trainer._init_train()
    pol = trainer._agent.actor_critic
    before = pol.engine.params_flat.clone()
    for _ in range(2):
        losses = trainer.run_update_cycle()
        assert all(np.isfinite(v) for v in losses.values()), losses
    assert trainer.num_steps_done == 2 * 4 * 8 and trainer.num_updates_done == 2
    assert float((pol.engine.params_flat - before).abs().max()) > 0
    assert set(losses) >= {"value_loss", "action_loss", "dist_entropy", "grad_norm"}
    trainer.envs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shared_obs", [True, False])
def test_trainer_host_path_with_process_vector_env(shared_obs):
    """N1: the trainer's generic host path (async_step_at / wait_step_at per env, double-buffered halves) over worker PROCESSES.
    With the shared-memory observation plane the rollout rows must hold exactly what the workers produced: the same envs are
    replayed in-process with the actions the policy sent."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    from habitat_amd.core.host_env import make_host_env
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401
    size, N, T = 64, 4, 6
    ov = [f"habitat_baselines.num_environments={N}", f"habitat_baselines.rl.ppo.num_steps={T}", "habitat_baselines.num_updates=3",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", "habitat_baselines.checkpoint_folder=/tmp/habitat_amd_test_ckpt",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000",
          "habitat_baselines.vector_env_factory._target_=habitat_amd.common.env_factory.ProcessVectorEnvFactory",
          f"habitat_baselines.vector_env_factory.shared_obs={shared_obs}"]
    for sname in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{sname}.height={size}", f"habitat.simulator.sensors.{sname}.width={size}"]
    cfg = get_config("pointnav/ppo_pointnav_example.yaml", ov)
    cfg.habitat.simulator.sensors.pop("semantic", None)
    trainer = baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)
    trainer._init_train()
    try:
        assert not trainer._device_envs and trainer.envs.num_envs == N
        assert bool(trainer.envs.shared_obs_keys) == shared_obs
        st = trainer._agent.rollouts
        trainer._agent.eval()
        steps = trainer.collect_rollout()
        assert steps == N * T
        # replay: same seeds, same actions -> same observations / rewards / masks in the rollout rows
        local = [make_host_env(int(cfg.habitat.seed) + i, size, size, True, True, len(cfg.habitat.task.actions),
                               int(cfg.habitat.environment.max_episode_steps)) for i in range(N)]
        obs = [e.reset() for e in local]
        B = st.buffers
        for k in obs[0]:
            assert np.array_equal(B["observations"][k][0].cpu().numpy(), np.stack([o[k] for o in obs])), k
        for t in range(T):
            acts = B["actions"][t].cpu().numpy().reshape(N)
            for i, e in enumerate(local):
                o, r, d, _ = e.step(int(acts[i]))
                if d:
                    o = e.reset()
                for k in o:
                    assert np.array_equal(B["observations"][k][t + 1, i].cpu().numpy(), o[k]), (t, i, k)
                assert abs(float(B["rewards"][t, i]) - r) < 1e-6 and bool(B["masks"][t + 1, i]) == (not d)
        losses = trainer._update_agent()
        assert all(np.isfinite(v) for v in losses.values()), losses
    finally:
        trainer.envs.close()


@pytest.mark.gpu
def test_trainer_host_path_applies_obs_transforms():
    """N4 in the loop: 96x128 sensors from worker processes -> ResizeShortestEdge(64) -> CenterCropper(64) on the device -> the
    policy and the rollout storage are built for 64x64, and the stored rows equal the oracle's transform of what the envs emitted."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    from habitat_amd.core.host_env import make_host_env
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401
    N, T = 2, 3
    pre = "habitat_baselines.rl.policy.main_agent.obs_transforms"
    ov = [f"habitat_baselines.num_environments={N}", f"habitat_baselines.rl.ppo.num_steps={T}", "habitat_baselines.num_updates=2",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", "habitat_baselines.checkpoint_folder=/tmp/habitat_amd_test_ckpt",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000",
          "habitat_baselines.vector_env_factory._target_=habitat_amd.common.env_factory.ProcessVectorEnvFactory",
          f"{pre}.resize.type=ResizeShortestEdge", f"{pre}.resize.size=64",
          f"{pre}.crop.type=CenterCropper", f"{pre}.crop.height=64", f"{pre}.crop.width=64"]
    for sname in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{sname}.height=96", f"habitat.simulator.sensors.{sname}.width=128"]
    cfg = get_config("pointnav/ppo_pointnav_example.yaml", ov)
    cfg.habitat.simulator.sensors.pop("semantic", None)
    trainer = baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)
    trainer._init_train()
    try:
        B = trainer._agent.rollouts.buffers
        assert tuple(B["observations"]["rgb"].shape[2:]) == (64, 64, 3) and tuple(B["observations"]["depth"].shape[2:]) == (64, 64, 1)
        trainer._agent.eval()
        assert trainer.collect_rollout() == N * T
        local = [make_host_env(int(cfg.habitat.seed) + i, 96, 128, True, True, len(cfg.habitat.task.actions),
                               int(cfg.habitat.environment.max_episode_steps)) for i in range(N)]

        def tf(o):
            out = {}
            for k in ("rgb", "depth"):
                r = O.resize_shortest_edge(torch.from_numpy(o[k]).unsqueeze(0), 64)
                out[k] = O.center_crop(r, 64)[0].numpy()
            return out

        obs = [e.reset() for e in local]
        for i in range(N):
            for k, v in tf(obs[i]).items():
                assert np.array_equal(B["observations"][k][0, i].cpu().numpy(), v), (0, i, k)
        for t in range(T):
            acts = B["actions"][t].cpu().numpy().reshape(N)
            for i, e in enumerate(local):
                o, _, d, _ = e.step(int(acts[i]))
                if d:
                    o = e.reset()
                for k, v in tf(o).items():
                    assert np.array_equal(B["observations"][k][t + 1, i].cpu().numpy(), v), (t, i, k)
        losses = trainer._update_agent()
        assert all(np.isfinite(v) for v in losses.values()), losses
    finally:
        trainer.envs.close()


@pytest.mark.gpu
def test_trainer_device_path_applies_obs_transforms():
    """N4 on the device-env path: the env source writes 96x128 sensors into a staging row, ResizeShortestEdge(64) -> CenterCropper(64)
    (device kernels) write the 64x64 rollout rows -- equal to the oracle's transform of what an identically seeded env source emits."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    from habitat_amd.common.env_factory import SyntheticVectorEnv
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401
    N, T = 3, 3
    pre = "habitat_baselines.rl.policy.main_agent.obs_transforms"
    ov = [f"habitat_baselines.num_environments={N}", f"habitat_baselines.rl.ppo.num_steps={T}", "habitat_baselines.num_updates=2",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", "habitat_baselines.checkpoint_folder=/tmp/habitat_amd_test_ckpt",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000",
          f"{pre}.resize.type=ResizeShortestEdge", f"{pre}.resize.size=64",
          f"{pre}.crop.type=CenterCropper", f"{pre}.crop.height=64", f"{pre}.crop.width=64"]
    for sname in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{sname}.height=96", f"habitat.simulator.sensors.{sname}.width=128"]
    cfg = get_config("pointnav/ppo_pointnav_example.yaml", ov)
    cfg.habitat.simulator.sensors.pop("semantic", None)
    trainer = baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)
    trainer._init_train()
    try:
        assert trainer._device_envs and trainer._raw_obs is not None
        B = trainer._agent.rollouts.buffers
        assert tuple(B["observations"]["rgb"].shape[2:]) == (64, 64, 3) and tuple(B["observations"]["depth"].shape[2:]) == (64, 64, 1)
        trainer._agent.eval()
        assert trainer.collect_rollout() == N * T
        twin = SyntheticVectorEnv(N, 96, 128, seed=int(cfg.habitat.seed), num_actions=len(cfg.habitat.task.actions))
        rgb = torch.zeros(N, 96, 128, 3, dtype=torch.uint8, device="cuda")
        depth = torch.zeros(N, 96, 128, 1, device="cuda")
        goal = torch.zeros(N, 2, device="cuda")
        rew, nd = torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda")

        def tf(x):
            return O.center_crop(O.resize_shortest_edge(x.cpu(), 64), 64).numpy()

        twin.reset_into(rgb, depth, goal)
        for t in range(T + 1):
            assert np.array_equal(B["observations"]["rgb"][t].cpu().numpy(), tf(rgb)), t
            assert np.array_equal(B["observations"]["depth"][t].cpu().numpy(), tf(depth)), t
            assert np.array_equal(B["observations"][GOAL][t].cpu().numpy(), goal.cpu().numpy()), t
            if t < T:
                twin.step_into(rgb, depth, goal, rew, nd)
        losses = trainer._update_agent()
        assert all(np.isfinite(v) for v in losses.values()), losses
    finally:
        trainer.envs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backbone,rnn_type,layers", [("resnet18", "LSTM", 2), ("resnet50", "GRU", 1)])
def test_frozen_encoder_visual_features_vs_oracle(backbone, rnn_type, layers):
    """N3, rl.ddppo.train_encoder=False: (1) `visual_encoder(batch)` alone equals the oracle's ResNetEncoder in eval mode and leaves
    RunningMeanAndVar untouched; (2) with `visual_features` in the observations evaluate_actions / backward equal the oracle's
    net on the same features (resnet_policy.py:636-648), every non-encoder gradient matches and NO encoder gradient is written."""
    from habitat_amd.common import spaces as S
    from habitat_amd.engine import DevicePackInfo
    from habitat_amd.rl.ppo import PointNavResNetPolicy
    H = W = 128
    hidden, T, n = 64, 3, 2
    B = T * n
    keys = ("rgb", "depth")
    osp = S.Dict({"rgb": S.Box(0, 255, (H, W, 3), np.uint8), "depth": S.Box(0.0, 1.0, (H, W, 1), np.float32),
                  GOAL: S.Box(-1e9, 1e9, (2,), np.float32)})
    params = det_params(resnet_param_shapes(4, H, W, hidden, rnn_type=rnn_type, layers=layers, backbone=backbone), 17)
    pre = "net.visual_encoder.running_mean_and_var."
    params[pre + "_mean"], params[pre + "_var"], params[pre + "_count"] = (
        torch.full((1, 4, 1, 1), 0.3), torch.full((1, 4, 1, 1), 0.05), torch.tensor(6.0))
    pol = PointNavResNetPolicy(osp, S.Discrete(4), hidden_size=hidden, num_recurrent_layers=layers, rnn_type=rnn_type, backbone=backbone,
                               normalize_visual_inputs=True, max_frames=B, max_envs=n)
    pol.load_state_dict(params)
    for q in pol.visual_encoder.parameters():
        q.requires_grad_(False)
    pol.to("cuda")
    assert all(not q.requires_grad for q in pol.visual_encoder.parameters())  # survives the move into the flat arena
    eng = pol.engine
    Lh = layers * (2 if rnn_type == "LSTM" else 1)
    spec = O.NetSpec(kind="resnet", rnn_type=rnn_type, num_layers=layers, backbone=backbone, baseplanes=32, visual_keys=keys,
                     normalize=True, hidden=hidden)
    rng = np.random.default_rng(5)
    obs = {"rgb": torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)),
           "depth": torch.from_numpy(rng.random((B, H, W, 1), dtype=np.float32)),
           GOAL: torch.from_numpy(np.stack([rng.random(B) * 5, rng.uniform(-3.1, 3.1, B)], 1).astype(np.float32))}
    masks = torch.from_numpy(rng.random((B, 1)) > 0.3)
    actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
    prev_actions = torch.from_numpy(rng.integers(0, 4, (B, 1)))
    h0 = torch.from_numpy(rng.standard_normal((n, Lh, hidden)).astype(np.float32))
    # (1) the encoder alone, eval mode
    pol.eval()
    with torch.no_grad():
        ref_feats = O.resnet_encoder(params, "net.visual_encoder.", obs, keys, backbone, 32, False, True)
    assert tuple(pol.visual_encoder.output_shape) == tuple(ref_feats.shape[1:])
    buf0 = {k: v.clone() for k, v in pol.state_dict().items() if is_buffer(k)}
    feats = pol.visual_encoder({k: v.cuda() for k, v in obs.items()})
    assert rel_ok(feats.cpu().numpy(), ref_feats.numpy(), tol=1e-4)
    for k, v0 in buf0.items():
        assert torch.equal(pol.state_dict()[k], v0)
    # (2) training step on stored features: the oracle consumes the SAME features, so everything downstream is comparable at 1e-4
    pol.train()
    p = {k: (v.clone().requires_grad_(not is_buffer(k))) for k, v in params.items()}
    obs_f = dict(obs, visual_features=ref_feats)
    v, lp, ent, hfin = O.evaluate_actions(p, spec, obs_f, h0, prev_actions, masks, actions, training=True)
    gv, glp, gent = (torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32)) for _ in range(3))
    ((v * gv).sum() + (lp * glp).sum() + (ent * gent).sum()).backward()
    pack = DevicePackInfo(np.logical_not(masks.view(T, n).numpy()), "cuda")
    dv, dl, de = (torch.zeros(B, device="cuda") for _ in range(3))
    # the rollout arena has MORE rows than the minibatch: features are gathered through rows[] like every other sensor
    rows = torch.from_numpy(rng.permutation(B + 3)[:B].astype(np.int32))
    arena = {k: torch.zeros((B + 3,) + tuple(t.shape[1:]), dtype=t.dtype) for k, t in obs_f.items()}
    for k, t in obs_f.items():
        arena[k][rows.long()] = t
    arena_pa = torch.zeros(B + 3, 1, dtype=torch.long)
    arena_pa[rows.long()] = prev_actions
    arena_masks = torch.zeros(B + 3, 1, dtype=torch.bool)
    arena_masks[rows.long()] = masks
    arena_act = torch.zeros(B + 3, 1, dtype=torch.long)
    arena_act[rows.long()] = actions
    arena_h = torch.zeros(B + 3, Lh, hidden)
    arena_h[rows[:n].long()] = h0
    extra = {"visual_features": arena["visual_features"].cuda()}
    eng.grads_flat.fill_(0.0)
    eng.evaluate(arena["rgb"].cuda(), arena["depth"].cuda(), arena[GOAL].cuda(), rows.cuda(), arena_h.cuda(), arena_masks.cuda(),
                 arena_act.cuda(), pack, B, n, value=dv, log_prob=dl, entropy=de, prev_actions=arena_pa.cuda(), extra=extra)
    for got, ref in ((dv, v), (dl, lp), (de, ent)):
        assert rel_ok(got.cpu().numpy(), ref.detach().numpy().reshape(-1), tol=1e-4)
    eng.backward(arena["rgb"].cuda(), arena["depth"].cuda(), arena[GOAL].cuda(), rows.cuda(), arena_act.cuda(), pack, gv.view(-1).cuda(),
                 glp.view(-1).cuda(), gent.view(-1).cuda(), prev_actions=arena_pa.cuda(), extra=extra)
    for k, g in eng.grad_views.items():
        if is_buffer(k):
            continue
        if k.startswith("net.visual_encoder."):
            assert float(g.abs().max()) == 0.0, k                      # frozen: nothing written
            assert p[k].grad is None or float(p[k].grad.abs().max()) == 0.0
        else:
            assert rel_ok(g.cpu().numpy(), p[k].grad.numpy(), tol=1e-4, floor=1e-5), k
    for k, v0 in buf0.items():
        assert torch.equal(pol.state_dict()[k], v0), "a training forward on stored features must not touch RunningMeanAndVar"


@pytest.mark.gpu
def test_trainer_frozen_encoder_update_cycles():
    """train_encoder=False through the YAML entrypoint: the rollout stores `visual_features` (= the encoder applied to the stored
    sensors), updates move every parameter EXCEPT the visual encoder's, which stay bit-identical."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401
    size = 128
    ov = ["habitat_baselines.num_environments=4", "habitat_baselines.rl.ppo.num_steps=6", "habitat_baselines.num_updates=3",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", "habitat_baselines.checkpoint_folder=/tmp/habitat_amd_test_ckpt",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000",
          "habitat_baselines.rl.ddppo.backbone=resnet18", "habitat_baselines.rl.ddppo.train_encoder=False"]
    for sname in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{sname}.height={size}", f"habitat.simulator.sensors.{sname}.width={size}"]
    cfg = get_config("pointnav/ddppo_pointnav.yaml", ov)
    cfg.habitat.simulator.sensors.pop("semantic", None)
    trainer = baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)
    trainer._init_train()
    pol = trainer._agent.actor_critic
    sd0 = {k: v.clone() for k, v in pol.state_dict().items()}
    B = trainer._agent.rollouts.buffers
    assert tuple(B["observations"]["visual_features"].shape) == (7, 4) + tuple(pol.visual_encoder.output_shape)
    for _ in range(2):
        losses = trainer.run_update_cycle()
        assert all(np.isfinite(v) for v in losses.values()), losses
    sd1 = pol.state_dict()
    enc = [k for k in sd0 if k.startswith("net.visual_encoder.")]
    rest = [k for k in sd0 if not k.startswith("net.visual_encoder.")]
    changed = [(k, float((sd0[k] - sd1[k]).abs().max())) for k in enc if not torch.equal(sd0[k], sd1[k])]
    assert enc and not changed, f"frozen encoder parameters / statistics changed: {changed[:6]}"
    assert all(not torch.equal(sd0[k], sd1[k]) for k in rest if sd0[k].numel() > 1)
    # the stored features are the encoder's output for the stored sensors (row 0 = the last step of the previous rollout)
    pol.eval()
    o = B["observations"]
    again = pol.encode_visual({k: v[0] for k, v in o.items()})
    assert torch.equal(again, o["visual_features"][0])
    trainer.envs.close()


@pytest.mark.gpu
def test_train_checkpoint_then_eval_loop(tmp_path):
    """N4: PPOTrainer.train() writes a checkpoint, PPOTrainer.eval() (base_trainer.py:66-168 -> _eval_checkpoint ->
    HabitatEvaluator) reloads it and evaluates `test_episode_count` episodes over worker-process envs, pausing envs whose next
    episode is already covered; the aggregated statistics are the means over the recorded episodes."""
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401
    ck, tb = str(tmp_path / "ckpt"), str(tmp_path / "tb")
    ov = ["habitat_baselines.num_environments=3", "habitat_baselines.rl.ppo.num_steps=4", "habitat_baselines.num_updates=2",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=1", "habitat_baselines.checkpoint_interval=-1",
          "habitat_baselines.rl.ppo.hidden_size=64", f"habitat_baselines.checkpoint_folder={ck}", f"habitat_baselines.tensorboard_dir={tb}",
          f"habitat_baselines.eval_ckpt_path_dir={ck}", "habitat_baselines.test_episode_count=7",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000", "habitat.environment.max_episode_steps=5",
          "habitat_baselines.vector_env_factory._target_=habitat_amd.common.env_factory.ProcessVectorEnvFactory"]
    for sname in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{sname}.height=64", f"habitat.simulator.sensors.{sname}.width=64"]
    cfg = get_config("pointnav/ppo_pointnav_example.yaml", ov)
    cfg.habitat.simulator.sensors.pop("semantic", None)
    trainer = baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)
    trainer.train()
    ckpts = [f for f in os.listdir(ck) if f.startswith("ckpt.")]
    assert ckpts, os.listdir(ck)
    trained = torch.load(os.path.join(ck, sorted(ckpts)[-1]), map_location="cpu", weights_only=False)["state_dict"]
    ev = baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)
    ev.eval()
    # the evaluated agent carries the checkpoint's weights
    for k, v in ev._agent.actor_critic.state_dict().items():
        assert torch.equal(v.cpu(), trained[k].cpu()), k
    from habitat_amd.rl.ppo.evaluator import HabitatEvaluator
    stats = ev.last_eval_stats
    assert {"reward", "episode_return", "num_steps"} <= set(stats)
    assert 1.0 <= stats["num_steps"] <= 5.0  # max_episode_steps = 5
    # with this env the summed rewards of an episode ARE its `episode_return` measure
    assert abs(stats["reward"] - stats["episode_return"]) < 1e-4


@pytest.mark.gpu
def test_update_from_preempted_short_rollout(monkeypatch):
    """DD-PPO's preemptive straggler rule ends a rollout after t' < T steps (ppo_trainer.py:641-653); GAE and the minibatch
    generator then use current_rollout_step_idx (rollout_storage.py:182,237).  A t' = 4 of T = 6 rollout: bootstrap value from
    row t', returns, and the whole PPO.update (metrics, parameters) against the oracle run on the first t' steps only."""
    from habitat_amd.rl.ppo import PPO
    case, ts = "baseline_rgbd44", 4
    z = np.load(os.path.join(G, case + ".npz"))
    c, params, spec, buf, next_value, pol, st = build(case, z)
    cfg = make_cfg(**c["cfg"])
    T, N = c["T"], c["N"]
    assert ts < T
    fill_storage(st, buf, z, T)
    st.current_rollout_step_idxs = [ts]
    B = st.buffers
    # rows >= t' keep whatever an earlier rollout left there; like the reference (ppo.py:139-153) the advantage statistics run
    # over ALL T+1 rows, so those stale rows count -- the oracle gets the very same full buffers
    B["returns"].fill_(0.0)
    pol.eval()
    last = st.get_last_step()
    nv = pol.get_value({k: v.contiguous() for k, v in last["observations"].items()}, last["recurrent_hidden_states"], last["prev_actions"],
                       last["masks"])
    st.compute_returns(nv, True, cfg.gamma, cfg.tau)
    cpu = lambda v: v.cpu().clone()
    bufc = {k: ({kk: cpu(vv) for kk, vv in v.items()} if isinstance(v, dict) else cpu(v)) for k, v in B.items()}
    with torch.no_grad():
        feats, _ = O.net_forward(params, spec, {k: v[ts] for k, v in bufc["observations"].items()}, bufc["recurrent_hidden_states"][ts],
                                 bufc["prev_actions"][ts], bufc["masks"][ts])
        nv_ref = O.heads(params, feats)[2]
    assert rel_ok(nv.cpu().numpy(), nv_ref.numpy())
    ret_ref, vp_ref = O.compute_returns(bufc["rewards"], bufc["value_preds"], bufc["masks"], nv_ref, ts, True, cfg.gamma, cfg.tau)
    assert rel_ok(B["returns"][:ts].cpu().numpy(), ret_ref[:ts].numpy())
    bufc["returns"], bufc["value_preds"] = ret_ref, vp_ref
    torch.manual_seed(3)
    perm = torch.randperm(N)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = dict(step=0, m={k: torch.zeros_like(v) for k, v in p.items()}, v={k: torch.zeros_like(v) for k, v in p.items()})
    chunks = [list(perm.chunk(cfg.num_mini_batch)) for _ in range(cfg.ppo_epoch)]
    ref_metrics = O.ppo_update(p, spec, bufc, ts, cfg, opt, list(p.keys()), perms=chunks)
    pol.train()
    ppo = PPO.from_config(pol, cfg)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perm)
    metrics = ppo.update(st)
    for k in ("value_loss", "action_loss", "dist_entropy", "grad_norm"):
        assert abs(metrics[k] - ref_metrics[k]) <= 1e-4 * max(1.0, abs(ref_metrics[k])), (k, metrics[k], ref_metrics[k])
    for k, v in pol.state_dict().items():
        assert rel_ok(v.cpu().numpy(), p[k].detach().numpy(), tol=1e-4, floor=1e-2), k


def _small_trainer(extra=(), N=4, T=8, size=64, path="pointnav/ppo_pointnav_habitat_iccv19.yaml", ckpt="/tmp/habitat_amd_test_ckpt"):
    from habitat_amd.config.default import get_config
    from habitat_amd.common.baseline_registry import baseline_registry
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401
    ov = [f"habitat_baselines.num_environments={N}", f"habitat_baselines.rl.ppo.num_steps={T}", "habitat_baselines.num_updates=6",
          "habitat_baselines.total_num_steps=-1", "habitat_baselines.num_checkpoints=-1", "habitat_baselines.checkpoint_interval=1000000",
          "habitat_baselines.rl.ppo.hidden_size=64", f"habitat_baselines.checkpoint_folder={ckpt}",
          "habitat_baselines.rl.preemption.save_resume_state_interval=1000000000"]
    for s in ("rgb", "depth"):
        ov += [f"habitat.simulator.sensors.{s}.height={size}", f"habitat.simulator.sensors.{s}.width={size}"]
    cfg = get_config(path, ov + list(extra))
    cfg.habitat.simulator.sensors.pop("semantic", None)
    return baseline_registry.get_trainer(cfg.habitat_baselines.trainer_name)(cfg)


def test_resume_state_is_the_reference_wire_format(tmp_path):
    """`.habitat-resume-state.pth` contents (single_agent_access_mgr.py:253-263, ppo.py:377-384): the bare
    `actor_critic.state_dict()` (no prefix), `optim_state` = `torch.optim.Adam.state_dict()` of the optimised parameters in
    `parameters()` order, `lr_sched_state`.  A real torch.optim.Adam loads it and continues exactly like the fused optimiser; a new
    trainer resumed from it continues bit-identically."""
    tr = _small_trainer(ckpt=str(tmp_path))
    tr._init_train()
    for _ in range(2):
        tr.run_update_cycle()
    agent = tr._agent
    pol, upd = agent.actor_critic, agent.updater
    rs = agent.get_resume_state()
    assert list(rs["state_dict"].keys()) == list(pol.state_dict().keys())  # un-prefixed, reference order
    osd = rs["optim_state"]
    assert set(osd) == {"state", "param_groups"} and osd["param_groups"][0]["params"] == list(range(len(list(pol.parameters()))))
    n_steps = 2 * tr.config.habitat_baselines.rl.ppo.ppo_epoch * tr.config.habitat_baselines.rl.ppo.num_mini_batch
    assert all(float(s["step"]) == n_steps for s in osd["state"].values())
    # a genuine torch.optim.Adam over copies of the parameters accepts the state ...
    clones = [torch.nn.Parameter(p.detach().clone()) for p in pol.parameters()]
    adam = torch.optim.Adam(clones, lr=upd.optimizer.param_groups[0]["lr"], eps=upd.optimizer.param_groups[0]["eps"])
    adam.load_state_dict(osd)
    # ... and makes the same next step as the fused kernel (same gradients, no clipping)
    eng = pol.engine
    g = torch.randn_like(eng.grads_flat) * 1e-3
    eng.grads_flat.copy_(g)
    for c, (nm, p) in zip(clones, pol.named_parameters()):
        c.grad = p.grad.detach().clone()
    before = eng.params_flat.clone()
    upd.optimizer.step(max_grad_norm=0.0, grad_scale=1.0)
    adam.step()
    assert float((eng.params_flat - before).abs().max()) > 0
    for c, (nm, p) in zip(clones, pol.named_parameters()):
        assert torch.allclose(p.detach(), c.detach(), rtol=0, atol=2e-7), nm
    # a second trainer resumed from the state continues bit-identically
    rs = agent.get_resume_state()
    rs_cpu = {"state_dict": {k: v.cpu().clone() for k, v in rs["state_dict"].items()}, "optim_state": rs["optim_state"],
              "lr_sched_state": rs["lr_sched_state"]}
    tr2 = _small_trainer(ckpt=str(tmp_path))
    tr2._init_train(resume_state=None)
    tr2._agent.load_state_dict(rs_cpu)
    e2 = tr2._agent.actor_critic.engine
    for (k, v1), v2 in zip(pol.state_dict().items(), tr2._agent.actor_critic.state_dict().values()):
        assert torch.equal(v1, v2), k
    o1, o2 = upd.optimizer, tr2._agent.updater.optimizer
    for _i, nm, off, n, _shp in o1._slots():  # (the arena's alignment padding is not part of the state)
        assert torch.equal(o2.exp_avg[off:off + n], o1.exp_avg[off:off + n]), nm
        assert torch.equal(o2.exp_avg_sq[off:off + n], o1.exp_avg_sq[off:off + n]), nm
    assert o2.step_count == o1.step_count
    # round-1 resume files ({step, exp_avg, exp_avg_sq} flat arenas, prefixed state_dict) still load
    legacy = {"state_dict": {"actor_critic." + k: v for k, v in rs_cpu["state_dict"].items()},
              "optim_state": dict(step=upd.optimizer.step_count, exp_avg=upd.optimizer.exp_avg.cpu(), exp_avg_sq=upd.optimizer.exp_avg_sq.cpu())}
    tr2._agent.load_state_dict(legacy)
    assert torch.equal(o2.exp_avg, o1.exp_avg)
    tr.envs.close()
    tr2.envs.close()


def test_synthetic_env_host_path_double_buffered_halves_advance_once():
    """Round-1 advisor finding: with the double-buffered sampler the two halves are stepped out of phase (async_step_at for half 0
    while half 1 is awaited, ppo_trainer.py:743-768); every env must advance exactly once per rollout step.  The host-path rollout
    rows must equal the oracle generator's stream env for env."""
    from oracle import synth
    N, T, size = 4, 5, 64
    tr = _small_trainer(extra=["habitat_baselines.rl.ppo.use_double_buffered_sampler=True"], N=N, T=T, size=size)
    tr._init_train()
    tr._device_envs = False  # force the generic VectorEnv protocol on the synthetic source
    st = tr._agent.rollouts
    observations = tr.envs.post_step(tr.envs.reset())
    from habitat_amd.rl.ppo.ppo_trainer import batch_obs
    st.insert_first_observations(batch_obs(observations, tr.device))
    tr.current_episode_reward = tr.current_episode_reward.cpu()
    tr.running_episode_stats = {k: v.cpu() for k, v in tr.running_episode_stats.items()}
    tr._agent.eval()
    assert tr.collect_rollout() == N * T
    ref = synth.SyntheticEnvs(N, size, size, seed=tr.config.habitat.seed)
    o = ref.reset()
    B = st.buffers
    for t in range(T + 1):
        for k in ("rgb", "depth", GOAL):
            assert np.array_equal(B["observations"][k][t].cpu().numpy(), o[k]), (t, k)
        if t < T:
            o, r, d = ref.step()
            assert np.array_equal(B["rewards"][t].cpu().numpy().reshape(-1), r)
            assert np.array_equal(B["masks"][t + 1].cpu().numpy().reshape(-1), ~d)
    # a genuinely partial request (one env only): that env advances, the others keep their clock and observations
    envs = tr.envs
    t_before = envs._t.clone()
    rgb_before = envs._rgb.clone()
    envs.async_step_at(1, 0)
    ob1, r1, d1, _ = envs.wait_step_at(1)
    o, r, d = ref.step()
    assert np.array_equal(ob1["rgb"], o["rgb"][1]) and r1 == float(r[1]) and d1 == bool(d[1])
    moved = (envs._t != t_before).cpu().numpy()
    assert moved.tolist() == [False, True, False, False] or bool(d[1])  # (an episode end resets env 1's clock)
    keep = [0, 2, 3]
    assert torch.equal(envs._rgb[keep], rgb_before[keep])
    tr.envs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path,extra,size", [
    ("pointnav/ppo_pointnav_habitat_iccv19.yaml", (), 64),
    ("pointnav/ddppo_pointnav.yaml", ("habitat_baselines.rl.ddppo.backbone=resnet18",), 64),
])
def test_whole_update_teacher_forced_vs_oracle(path, extra, size):
    """oracle/parity.py::update_parity (the leg bench.py reports at the benchmark shape) on a small trainer built from the YAML
    entrypoints: the oracle produces a rollout and traces its update; every minibatch step of the HIP updater, restarted from the
    oracle's pre-step parameters + Adam moments (+ RunningMeanAndVar buffers for the ResNet policy), must reproduce the step's losses
    to 1e-4 (north_star), its gradient norm to 1e-4 (SimpleCNN) / 1e-3 (GroupNorm encoder on noise inputs: ReLU-boundary flips, see
    test_resnet_golden_mask_flip_accounting) and its parameter step to a fraction of lr."""
    import types
    from oracle import parity as PR
    N, T = 4, 16
    trainer = _small_trainer(extra=("habitat_baselines.rl.ppo.num_mini_batch=2", "habitat_baselines.rl.ppo.ppo_epoch=2") + tuple(extra),
                             N=N, T=T, size=size, path=path)
    trainer._init_train()
    trainer.run_update_cycle()  # (not the initial parameters: moments, statistics and clip decisions are exercised from a moved policy)
    pol = trainer._agent.actor_critic
    ppo = trainer.config.habitat_baselines.rl.ppo
    ocfg = types.SimpleNamespace(clip_param=ppo.clip_param, ppo_epoch=ppo.ppo_epoch, num_mini_batch=ppo.num_mini_batch,
                                 value_loss_coef=ppo.value_loss_coef, entropy_coef=ppo.entropy_coef, lr=ppo.lr, eps=ppo.eps,
                                 max_grad_norm=ppo.max_grad_norm, use_normalized_advantage=ppo.use_normalized_advantage,
                                 use_clipped_value_loss=ppo.use_clipped_value_loss, gamma=ppo.gamma, tau=ppo.tau)
    params = {k: v.detach().cpu().clone() for k, v in pol.state_dict().items()}
    spec = PR.spec_of(pol)
    trainable = [k for k, p_ in pol.named_parameters() if p_.requires_grad]
    buf, nv, perms, _ = PR.oracle_rollout(params, spec, N, T, size, size, pol.recurrent_hidden_size, pol.num_recurrent_layers, ocfg)
    ref_metrics, trace, final = PR.oracle_update_trace(params, spec, buf, T, ocfg, trainable, perms)
    assert len(trace) == 4 and [t["step"] for t in trace] == [0, 1, 2, 3]
    es = trainer._env_spec
    st = PR.storage_from_oracle(buf, nv, T, N, es.observation_space, es.action_space, pol, trainer.device, ocfg)
    assert PR.rel(st.buffers["returns"].cpu().numpy()[:T], buf["returns"].numpy()[:T]) <= 1e-5
    from habitat_amd.rl.ppo import PPO
    pol.load_state_dict(params)
    upd = PPO.from_config(pol, ocfg)
    rep = PR.update_parity(pol, upd, st, buf, trace, final, T, ocfg, trainable)
    tf, fr = rep["teacher_forced"], rep["free_running"]
    print(json.dumps(rep))
    deep = "ddppo" in path
    assert tf["max_rel_losses"] <= 1e-4, tf
    assert tf["max_rel_grad_norm"] <= (1e-3 if deep else 1e-4), tf
    assert max(tf["value_max_rel"]) <= 1e-4 and max(tf["log_prob_max_rel"]) <= 1e-4, tf
    assert max(tf["param_step_max_err_over_lr"]) <= (0.5 if deep else 0.05), tf  # (Adam's first steps: g / (|g| + eps) near g = 0)
    assert fr["param_max_abs_drift_before_step"][0] == 0.0
    assert fr["post_update_param_max_abs_diff"] <= 4 * ocfg.lr
    trainer.envs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["baseline", "resnet18"])
def test_auxiliary_loss_hook_vs_oracle(kind):
    """`aux_loss_modules` (rl/ppo/policy.py:253-291,386-394; rl/ppo/ppo.py:248): a registered auxiliary loss receives `aux_loss_state`
    = {rnn_output, perception_embed} from evaluate_actions on the autograd bridge, its loss joins the PPO loss and its gradients wrt
    those two tensors re-enter the engine's backward (hab_policy_set_extra_grads).  Against the oracle's autograd of the SAME total
    loss: every policy gradient and the module's own parameter gradient <= 1e-4; then a full PPO.update through the aux path moves
    policy and module parameters with finite metrics."""
    from habitat_amd.common import spaces as S
    from habitat_amd.common.baseline_registry import baseline_registry
    from habitat_amd.common.rollout_storage import RolloutStorage
    from habitat_amd.rl.ppo import PPO, PointNavBaselinePolicy, PointNavResNetPolicy

    @baseline_registry.register_auxiliary_loss(name="toy_aux")
    class ToyAux(torch.nn.Module):
        def __init__(self, action_space, net, loss_scale=0.1, **kw):
            super().__init__()
            assert net.output_size == net.perception_embedding_size and not net.is_blind
            self.scale = float(loss_scale)
            self.w = torch.nn.Parameter(torch.linspace(0.5, 1.5, net.output_size))

        def forward(self, aux_loss_state, batch):
            out, pe = aux_loss_state["rnn_output"], aux_loss_state["perception_embed"]
            assert batch["action"].shape[0] == out.shape[0]
            return dict(loss=self.scale * ((out * self.w).pow(2).mean() + torch.tanh((pe * self.w).sum(1)).mean()))

    H = W = 64 if kind == "resnet18" else 44
    T, N, hidden = 6, 4, 64
    osp = S.Dict({"rgb": S.Box(0, 255, (H, W, 3), np.uint8), "depth": S.Box(0.0, 1.0, (H, W, 1), np.float32),
                  GOAL: S.Box(-1e9, 1e9, (2,), np.float32)})
    asp = S.Discrete(4)
    torch.manual_seed(13)
    aux_cfg = {"toy_aux": {"loss_scale": 0.2}}
    if kind == "baseline":
        pol = PointNavBaselinePolicy(osp, asp, hidden_size=hidden, aux_loss_config=aux_cfg, max_frames=T * N, max_envs=N)
        spec = O.NetSpec(kind="baseline", hidden=hidden)
    else:
        pol = PointNavResNetPolicy(osp, asp, hidden_size=hidden, backbone="resnet18", aux_loss_config=aux_cfg, max_frames=T * N, max_envs=N)
        spec = O.NetSpec(kind="resnet", rnn_type="GRU", num_layers=1, backbone="resnet18", baseplanes=32, visual_keys=("rgb", "depth"),
                         normalize=False, hidden=hidden)
    assert list(pol.aux_loss_modules) == ["toy_aux"] and "aux_loss_modules.toy_aux.w" in pol.state_dict()
    params = {k: v.detach().clone() for k, v in pol.state_dict().items() if not k.startswith("aux_loss_modules.")}
    pol.to("cuda")
    pol.train()
    assert pol.aux_loss_modules["toy_aux"].w.is_cuda
    rng = np.random.default_rng(3)
    st = RolloutStorage(T, N, osp, asp, pol, device="cuda", gae_variant="scan")
    B = st.buffers
    B["observations"]["rgb"].copy_(torch.from_numpy(rng.integers(0, 256, (T + 1, N, H, W, 3), dtype=np.uint8)))
    B["observations"]["depth"].copy_(torch.from_numpy(rng.random((T + 1, N, H, W, 1), dtype=np.float32)))
    B["observations"][GOAL].copy_(torch.from_numpy(rng.standard_normal((T + 1, N, 2)).astype(np.float32)))
    B["masks"].copy_(torch.from_numpy(rng.random((T + 1, N, 1)) > 0.2))
    B["actions"].copy_(torch.from_numpy(rng.integers(0, 4, (T + 1, N, 1))))
    B["prev_actions"].copy_(torch.from_numpy(rng.integers(0, 4, (T + 1, N, 1))))
    B["recurrent_hidden_states"].copy_(torch.from_numpy(rng.standard_normal((T + 1, N, 1, hidden)).astype(np.float32)))
    for k in ("rewards", "value_preds", "returns", "action_log_probs"):
        B[k].copy_(torch.from_numpy((rng.standard_normal((T + 1, N, 1)) * (0.1 if k != "action_log_probs" else 0.05) -
                                     (1.3 if k == "action_log_probs" else 0.0)).astype(np.float32)))
    st.current_rollout_step_idxs = [T]
    cfg = types.SimpleNamespace(clip_param=0.2, ppo_epoch=1, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.01, lr=2.5e-4, eps=1e-5,
                                max_grad_norm=0.5, use_clipped_value_loss=True, use_normalized_advantage=False)
    ppo = PPO.from_config(pol, cfg)
    adv = ppo.get_advantages(st)
    torch.manual_seed(5)
    batch = next(st.data_generator(adv, 1))
    # ---- oracle: the same total loss by autograd ----
    inds = batch.inds
    take = lambda t: t[0:T, inds].flatten(0, 1).cpu()
    obs = {k: take(v) for k, v in B["observations"].items()}
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    w_ref = torch.linspace(0.5, 1.5, hidden).requires_grad_(True)
    taps = {}
    v_o, lp_o, ent_o, _ = O.evaluate_actions(p, spec, obs, B["recurrent_hidden_states"][0, inds].cpu(), take(B["prev_actions"]), take(B["masks"]),
                                             take(B["actions"]), training=True, taps=taps)
    ob = {"action_log_probs": take(B["action_log_probs"]), "advantages": take(adv), "value_preds": take(B["value_preds"]), "returns": take(B["returns"])}
    total_o, *_ = O.ppo_loss(v_o, lp_o, ent_o, ob, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)
    pe = taps["cnn_out"] if kind == "baseline" else taps["visual_fc"]
    aux_o = 0.2 * ((taps["rnn_out"] * w_ref).pow(2).mean() + torch.tanh((pe * w_ref).sum(1)).mean())
    (total_o + aux_o).backward()
    # ---- engine: evaluate_actions on the bridge + the module, torch loss, backward ----
    for q in pol.parameters():
        q.grad = None
    v, lp, ent, _, aux = pol.evaluate_actions(batch["observations"], batch["recurrent_hidden_states"], batch["prev_actions"], batch["masks"],
                                              batch["actions"], batch["rnn_build_seq_info"])
    assert abs(float(aux["toy_aux"]["loss"]) - float(aux_o)) <= 1e-4 * max(1.0, abs(float(aux_o)))
    b = {k: batch[k] for k in ("action_log_probs", "advantages", "value_preds", "returns")}
    total, *_ = O.ppo_loss(v, lp, ent, b, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)
    (total + aux["toy_aux"]["loss"]).backward()
    eng = pol.engine
    tol_n = 1e-4 if kind == "baseline" else 2e-2  # (norm-wise; deep encoder on noise inputs: ReLU-boundary flips upstream, see the golden tests)
    bad = []
    for k, g in eng.grad_views.items():
        if k in eng.buffer_names:
            continue
        r = p[k].grad.numpy().astype(np.float64)
        gg = g.cpu().numpy().astype(np.float64)
        err = np.linalg.norm(gg - r) / max(1e-30, np.linalg.norm(r))
        deep = "visual_encoder" in k or "visual_fc" in k
        if err > (tol_n if deep else 1e-4):
            bad.append((k, err))
    assert not bad, bad
    gw = pol.aux_loss_modules["toy_aux"].w.grad.cpu()
    assert rel_ok(gw.numpy(), w_ref.grad.numpy(), tol=1e-4, floor=1e-5)
    # and WITHOUT the hook's gradients the arena would differ: the aux term reaches the encoder (not only the module's own parameter)
    key = "net.state_encoder.rnn.weight_hh_l0"
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    v2, lp2, ent2, _ = O.evaluate_actions(p2, spec, obs, B["recurrent_hidden_states"][0, inds].cpu(), take(B["prev_actions"]), take(B["masks"]),
                                          take(B["actions"]), training=True)
    O.ppo_loss(v2, lp2, ent2, ob, cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.use_clipped_value_loss)[0].backward()
    assert float((p2[key].grad - p[key].grad).norm()) > 1e-3 * float(p[key].grad.norm())
    # ---- the updater's aux path end to end ----
    before = eng.params_flat.clone()
    w_before = pol.aux_loss_modules["toy_aux"].w.detach().clone()
    metrics = ppo.update(st)
    assert all(np.isfinite(x) for x in metrics.values()) and metrics["grad_norm"] > 0
    assert float((eng.params_flat - before).abs().max()) > 0 and float((pol.aux_loss_modules["toy_aux"].w - w_before).abs().max()) > 0
    assert torch.isfinite(ppo.last_aux_losses["toy_aux"]).all()
