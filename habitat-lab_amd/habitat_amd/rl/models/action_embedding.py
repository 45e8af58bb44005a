"""Action embeddings for the auxiliary losses (reference: habitat_baselines/rl/models/action_embedding.py:18-146).  Host-side torch
modules: they are part of an auxiliary-loss module, which lives outside the engine's parameter arena (rl/ppo/policy.py,
`_build_aux_modules`), so they are ordinary nn.Modules.  Parameter / buffer NAMES follow the reference so that a checkpoint with
`aux_loss_modules.cpca.*` entries loads in either direction:

  _action_embed.embedding_modules.<i>.embedding.weight         Discrete leaf: table of n + 1 rows, row 0 = "no previous action"
  _action_embed.embedding_modules.<i>.{_action_low,_action_high,_freqs}   Box leaf: buffers of the sinusoidal embedding
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn


def _leaves(space):
    """Leaves of a (possibly nested) dictionary action space in declaration order (utils/common.py:677-682)."""
    sub = getattr(space, "spaces", None)
    if sub is not None:
        for v in sub.values():
            yield from _leaves(v)
    else:
        yield space


def _is_box(space) -> bool: return getattr(space, "low", None) is not None and getattr(space, "high", None) is not None
def _is_discrete(space) -> bool: return getattr(space, "spaces", None) is None and hasattr(space, "n")


class BoxActionEmbedding(nn.Module):
    """Continuous action -> [sin(x f_0) .. sin(x f_{b-1}), cos(x f_0) .. cos(x f_{b-1})] per action dimension, f_t = pi 2^t,
    b = dim_per_action / 2, x = the action mapped from [low, high] to [-1, 1] and clamped (action_embedding.py:18-72)."""

    def __init__(self, action_space, dim_per_action: int = 32):
        super().__init__()
        bands = dim_per_action // 2
        self._space_rank = len(action_space.shape)
        self.n_actions = int(np.prod(action_space.shape))
        self.register_buffer("_action_low", torch.as_tensor(np.array(action_space.low), dtype=torch.float32))
        self.register_buffer("_action_high", torch.as_tensor(np.array(action_space.high), dtype=torch.float32))
        self.register_buffer("_freqs", torch.logspace(0, bands - 1, bands, base=2.0, dtype=torch.float32) * math.pi)
        self.output_size = 2 * bands * self.n_actions

    def forward(self, action, masks=None):
        a = action.to(torch.float32)
        if masks is not None:
            a = a * masks.to(a.dtype)
        unit = ((a - self._action_low) * (2 / (self._action_high - self._action_low)) + 1).flatten(-self._space_rank).clamp(-1, 1)
        phase = (unit.unsqueeze(-1) * self._freqs).flatten(-2)
        return torch.cat((phase.sin(), phase.cos()), dim=-1)


class DiscreteActionEmbedding(nn.Module):
    """Table of n + 1 rows; action a reads row a + 1, a masked (episode-start) step reads row 0 (action_embedding.py:75-92)."""

    def __init__(self, action_space, dim_per_action: int):
        super().__init__()
        self.n_actions = 1
        self.output_size = dim_per_action
        self.embedding = nn.Embedding(int(action_space.n) + 1, dim_per_action)

    def forward(self, action, masks=None):
        row = action.long() + 1
        if masks is not None:
            row = row * masks.to(row.dtype)
        return self.embedding(row.squeeze(-1))


class ActionEmbedding(nn.Module):
    """One embedding per leaf of the action space, concatenated; leaf i reads columns [ptr_i, ptr_i + n_actions_i) of the action
    tensor.  A task action space whose leaves are all argument-less (habitat's ActionSpace of EmptySpace: pointnav / objectnav) is
    one Discrete table over the top-level `n` (action_embedding.py:95-146)."""

    def __init__(self, action_space, dim_per_action: int = 32):
        super().__init__()
        self.embedding_modules = nn.ModuleList()
        self.embedding_slices = []
        leaves = list(_leaves(action_space))
        if hasattr(action_space, "n") and all(not _is_box(s) and not _is_discrete(s) for s in leaves):
            leaves, spaces = [], [action_space]  # a table over the task's actions
        else:
            spaces = leaves
        ptr = 0
        for s in spaces:
            if _is_box(s):
                m = BoxActionEmbedding(s, dim_per_action)
            elif hasattr(s, "n"):
                m = DiscreteActionEmbedding(s, dim_per_action)
            else:
                raise RuntimeError(f"Unknown space: {s}")
            self.embedding_modules.append(m)
            self.embedding_slices.append(slice(ptr, ptr + m.n_actions))
            ptr += m.n_actions
        self._output_size = sum(m.output_size for m in self.embedding_modules)

    @property
    def output_size(self) -> int: return self._output_size

    def forward(self, action, masks=None):
        return torch.cat([m(action[..., sl], masks) for sl, m in zip(self.embedding_slices, self.embedding_modules)], dim=-1)
