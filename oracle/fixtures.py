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


def resnet_param_shapes(n_in, H, W, hidden, num_actions=4, rnn_type="LSTM", layers=2, backbone="resnet18", baseplanes=32,
                        normalize=True, with_buffers=False, has_goal=True, n_obj=0, has_gps=False, has_compass=False, gauss=None):
    """state_dict() names/shapes of PointNavResNetPolicy (rl/ddppo/policy/resnet_policy.py:50-162,391-602) in reference
    order.  Buffers (RunningMeanAndVar statistics) are listed only with with_buffers=True."""
    import math
    if gauss is not None:  # continuous actions: Linear(A, 32) previous-action embedding, GaussianNet head (utils/common.py:112-149)
        shapes = [("net.prev_action_embedding.weight", (32, num_actions)), ("net.prev_action_embedding.bias", (32,))]
    else:
        shapes = [("net.prev_action_embedding.weight", (num_actions + 1, 32))]
    slots = 1
    if has_goal:
        shapes += [("net.tgt_embeding.weight", (32, 3)), ("net.tgt_embeding.bias", (32,))]
        slots += 1
    if n_obj:
        shapes += [("net.obj_categories_embedding.weight", (n_obj, 32))]
        slots += 1
    if has_gps:
        shapes += [("net.gps_embedding.weight", (32, 2)), ("net.gps_embedding.bias", (32,))]
        slots += 1
    if has_compass:
        shapes += [("net.compass_embedding.weight", (32, 2)), ("net.compass_embedding.bias", (32,))]
        slots += 1
    ve = "net.visual_encoder."
    if normalize and with_buffers:
        shapes += [(ve + "running_mean_and_var._mean", (1, n_in, 1, 1)), (ve + "running_mean_and_var._var", (1, n_in, 1, 1)),
                   (ve + "running_mean_and_var._count", ())]
    from .functional import RESNET_LAYERS
    kind, nblocks, resnext, se = RESNET_LAYERS[backbone]
    bottleneck = kind == "bottleneck"
    exp = (2 if resnext else 4) if bottleneck else 1
    card = baseplanes // 2 if resnext else 1

    def cg(w, g, cin, cout, k, groups=1):
        return [(w + ".weight", (cout, cin // groups, k, k)), (g + ".weight", (cout,)), (g + ".bias", (cout,))]

    bb = ve + "backbone."
    shapes += cg(bb + "conv1.0", bb + "conv1.1", n_in, baseplanes, 7)
    inplanes = baseplanes
    for li, nb in enumerate(nblocks):
        planes = (2 * baseplanes if resnext else baseplanes) * 2 ** li
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            bp = f"{bb}layer{li + 1}.{bi}."
            has_ds = bi == 0 and (stride != 1 or inplanes != planes * exp)
            if not bottleneck:
                shapes += cg(bp + "convs.0", bp + "convs.1", inplanes, planes, 3) + cg(bp + "convs.3", bp + "convs.4", planes, planes, 3)
            else:
                shapes += (cg(bp + "convs.0", bp + "convs.1", inplanes, planes, 1)
                           + cg(bp + "convs.3", bp + "convs.4", planes, planes, 3, card if bi == 0 else 1)
                           + cg(bp + "convs.6", bp + "convs.7", planes, planes * exp, 1))
            if has_ds:
                shapes += cg(bp + "downsample.0", bp + "downsample.1", inplanes, planes * exp, 1)
            if se and bottleneck:
                c_se = planes * exp
                shapes += [(bp + "se.excite.0.weight", (c_se // 16, c_se)), (bp + "se.excite.0.bias", (c_se // 16,)),
                           (bp + "se.excite.2.weight", (c_se, c_se // 16)), (bp + "se.excite.2.bias", (c_se,))]
            inplanes = planes * exp
    fh, fw = math.ceil((H // 2) / 32), math.ceil((W // 2) / 32)
    ncomp = int(round(2048 / (fh * fw)))
    shapes += cg(ve + "compression.0", ve + "compression.1", inplanes, ncomp, 3)
    shapes += [("net.visual_fc.1.weight", (hidden, ncomp * fh * fw)), ("net.visual_fc.1.bias", (hidden,))]
    G = 3 if rnn_type == "GRU" else 4
    rn = "net.state_encoder.rnn."
    for l in range(layers):
        i = hidden + 32 * slots if l == 0 else hidden
        shapes += [(f"{rn}weight_ih_l{l}", (G * hidden, i)), (f"{rn}weight_hh_l{l}", (G * hidden, hidden)),
                   (f"{rn}bias_ih_l{l}", (G * hidden,)), (f"{rn}bias_hh_l{l}", (G * hidden,))]
    if gauss is not None:
        k = num_actions if gauss["use_std_param"] else 2 * num_actions
        if gauss["use_std_param"]:
            shapes += [("action_distribution.std", (num_actions,))]
        shapes += [("action_distribution.mu_maybe_std.weight", (k, hidden)), ("action_distribution.mu_maybe_std.bias", (k,)),
                   ("critic.fc.weight", (1, hidden)), ("critic.fc.bias", (1,))]
        return shapes
    shapes += [("action_distribution.linear.weight", (num_actions, hidden)), ("action_distribution.linear.bias", (num_actions,)),
               ("critic.fc.weight", (1, hidden)), ("critic.fc.bias", (1,))]
    return shapes


def golden_sample(x, max_elems=2048):
    """Deterministic strided subsample used to keep golden fixtures of multi-million-element tensors small."""
    flat = np.asarray(x).reshape(-1)
    if flat.size <= max_elems:
        return flat
    return flat[:: flat.size // max_elems]
