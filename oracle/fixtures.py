"""TEST INFRASTRUCTURE ONLY -- deterministic inputs shared by tests/golden/make_golden.py (which runs
the real reference on them) and by the tests (which re-create them where the reference is absent)."""
import numpy as np
import torch

from . import synth

GOAL = "pointgoal_with_gps_compass"


def det_params(named_shapes, seed):
    """Deterministic parameter values for a list of (name, shape): N(0, 1/sqrt(fan_in)) for matrices /
    filters, N(0, 0.1) for biases, 1 + N(0, 0.1) for norm scales."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in named_shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) / np.sqrt(fan_in)
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        else:
            v = 0.1 * rng.standard_normal(shape)
        out[name] = torch.from_numpy(v.astype(np.float32))
    return out


def baseline_param_shapes(cin, H, W, hidden, num_actions=4, goal_dim=2, rnn_type="GRU", layers=1):
    """state_dict() names/shapes of PointNavBaselinePolicy (rl/ppo/policy.py:427-589) in reference order."""
    def co(x, k, s):
        return (x - k) // s + 1
    h, w = co(co(co(H, 8, 4), 4, 2), 3, 1), co(co(co(W, 8, 4), 4, 2), 3, 1)
    G = 3 if rnn_type == "GRU" else 4
    ve = "net.visual_encoder.cnn."
    shapes = [(ve + "0.weight", (32, cin, 8, 8)), (ve + "0.bias", (32,)), (ve + "2.weight", (64, 32, 4, 4)),
              (ve + "2.bias", (64,)), (ve + "4.weight", (32, 64, 3, 3)), (ve + "4.bias", (32,)),
              (ve + "6.weight", (hidden, 32 * h * w)), (ve + "6.bias", (hidden,))]
    rn = "net.state_encoder.rnn."
    for l in range(layers):
        i = hidden + goal_dim if l == 0 else hidden
        shapes += [(f"{rn}weight_ih_l{l}", (G * hidden, i)), (f"{rn}weight_hh_l{l}", (G * hidden, hidden)),
                   (f"{rn}bias_ih_l{l}", (G * hidden,)), (f"{rn}bias_hh_l{l}", (G * hidden,))]
    shapes += [("action_distribution.linear.weight", (num_actions, hidden)), ("action_distribution.linear.bias", (num_actions,)),
               ("critic.fc.weight", (1, hidden)), ("critic.fc.bias", (1,))]
    return shapes


def synth_rollout_inputs(envs: synth.SyntheticEnvs, T):
    """obs for steps 0..T, rewards/dones for steps 1..T."""
    obs = [envs.reset()]
    rew, done = [], []
    for _ in range(T):
        o, r, d = envs.step()
        obs.append(o)
        rew.append(r)
        done.append(d)
    return obs, np.stack(rew), np.stack(done)
