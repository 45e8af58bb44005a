#!/usr/bin/env python3
"""Golden fixture for the auxiliary loss `cpca`: the REAL reference module (habitat_baselines/rl/ppo/cpc_aux_loss.py, loaded in place through
oracle/ref_loader.load_reference_aux) run on seeded inputs.

  python tests/golden/make_golden_cpca.py     # needs /root/reference; writes tests/golden/cpca.npz

Stored per case: the reference's rnn_build_seq_info arrays (its tie order between equally long fragments is numpy's argsort's and decides
which fragment gets which random draw), its seeded parameters, and its loss and gradients wrt rnn_output / perception_embed / every
parameter.  The inputs are re-created by `cpca_case_inputs` below (imported by the test).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

# name -> (action space description, module kwargs, T, N, H, seed)
CASES = {
    "discrete_defaults": (("discrete", 4), {}, 24, 6, 32, 0),
    "all_futures_kept": (("discrete", 4), dict(k=3, time_subsample=4, future_subsample=3, num_negatives=5, loss_scale=0.3), 24, 6, 32, 1),
    "negatives_with_replacement": (("discrete", 4), dict(num_negatives=200), 8, 4, 16, 2),
    "box": (("box", 2), {}, 24, 6, 32, 3),
    "task_of_argumentless_actions": (("task", 4), {}, 24, 6, 32, 4),
    "nested_dict": (("dict", 3), dict(k=6), 16, 5, 24, 5),
}
SEQ_KEYS = ("select_inds", "num_seqs_at_step", "sequence_lengths")


def action_space(M, desc, empty=None, task=None):
    """The case's action space built from module `M`'s Box / Discrete / Dict (ours or the reference's gym stub)."""
    kind, n = desc
    if kind == "discrete":
        return M.Discrete(n)
    if kind == "box":
        return M.Box(-1, 1, (n,), np.float32)
    if kind == "task":
        names = ["stop", "move_forward", "turn_left", "turn_right"][:n]
        return task({k: empty() for k in names})
    return M.Dict({"a": M.Discrete(4), "b": M.Box(-1, 1, (n - 1,), np.float32)})


def cpca_case_inputs(name):
    """-> dones [T, N] bool, rnn_output [P, H], perception_embed [P, H], action [P, A], net stand-in, seed."""
    desc, kw, T, N, H, seed = CASES[name]
    g = np.random.default_rng(100 + seed)
    dones = g.random((T, N)) < 0.12
    dones[0] = False
    tg = torch.Generator().manual_seed(200 + seed)
    P = T * N
    x, e = torch.randn(P, H, generator=tg), torch.randn(P, H, generator=tg)
    kind, n = desc
    if kind == "box":
        act = torch.rand(P, n, generator=tg) * 2 - 1
    elif kind == "dict":
        act = torch.cat([torch.randint(0, 4, (P, 1), generator=tg).float(), torch.rand(P, n - 1, generator=tg) * 2 - 1], 1)
    else:
        act = torch.randint(0, 4, (P, 1), generator=tg)
    net = types.SimpleNamespace(is_blind=False, output_size=H, perception_embedding_size=H)
    return dones, x, e, act, net, seed


def run_module(module, info, x, e, act, seed):
    """Seeds torch's generator, runs the loss, returns (loss, d rnn_output, d perception_embed, {parameter: gradient})."""
    xx, ee = x.clone().requires_grad_(), e.clone().requires_grad_()
    for p in module.parameters():
        p.grad = None
    torch.manual_seed(300 + seed)
    loss = module({"rnn_output": xx, "perception_embed": ee}, {"action": act.clone(), "rnn_build_seq_info": info})["loss"]
    loss.backward()
    return loss.detach(), xx.grad, ee.grad, {k: p.grad.clone() for k, p in module.named_parameters()}


def main():
    from oracle.ref_loader import load_reference_aux
    ns = load_reference_aux()
    out = {}
    for name, (desc, kw, T, N, H, seed) in CASES.items():
        dones, x, e, act, net, seed = cpca_case_inputs(name)
        torch.manual_seed(seed)
        ref = ns.cpc_aux_loss.CPCA(action_space(ns.spaces, desc, ns.EmptySpace, ns.ActionSpace), net, **kw)
        info = ns.rnn_state_encoder.build_rnn_build_seq_info(torch.device("cpu"), ns.rnn_state_encoder.build_pack_info_from_dones(dones))
        for k in SEQ_KEYS:
            out[f"{name}/info/{k}"] = info[k].numpy()
        for k, v in ref.state_dict().items():
            out[f"{name}/state/{k}"] = v.numpy()
        loss, gx, ge, gp = run_module(ref, info, x, e, act, seed)
        out[f"{name}/loss"], out[f"{name}/d_rnn_output"], out[f"{name}/d_perception_embed"] = loss.numpy(), gx.numpy(), ge.numpy()
        for k, v in gp.items():
            out[f"{name}/grad/{k}"] = v.numpy()
        print(name, float(loss), "starts:", int((info["sequence_lengths"] > 1).sum()), "sequences")
    np.savez_compressed(os.path.join(HERE, "cpca.npz"), **out)


if __name__ == "__main__":
    main()
