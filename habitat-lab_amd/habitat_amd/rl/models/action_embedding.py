"""Action codes for the auxiliary losses -- what habitat_baselines/rl/models/action_embedding.py:18-146 computes, for `cpca`
(rl/ppo/cpc_aux_loss.py).  These are host-side torch modules: an auxiliary-loss module lives outside the engine's parameter arena
(rl/ppo/policy.py, `_build_aux_modules`).

Checkpoint layout (must equal the reference's, `aux_loss_modules.cpca._action_embed.*`):
  embedding_modules.<i>.embedding.weight                          leaf i is Discrete(n): [n + 1, 32] table, row 0 = "no action yet"
  embedding_modules.<i>._action_low / _action_high / _freqs       leaf i is a Box: bounds and the pi 2^t frequency ladder (buffers)
Leaf i of the (possibly nested) action space reads columns `embedding_slices[i]` of the action tensor; the codes are concatenated.
"""
from __future__ import annotations

import math
from typing import Iterator, List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _walk(space) -> Iterator:
    """Depth-first leaves of a dictionary action space, declaration order (utils/common.py:677-682)."""
    children = getattr(space, "spaces", None)
    if children is None:
        yield space
        return
    for child in children.values():
        yield from _walk(child)


def _kind(space) -> str:
    if getattr(space, "spaces", None) is not None:
        return "dict"
    if getattr(space, "low", None) is not None and getattr(space, "high", None) is not None:
        return "box"
    return "discrete" if hasattr(space, "n") else "empty"


class BoxActionEmbedding(nn.Module):
    """Sinusoidal code of a continuous action: per action dimension [sin(u f_0..f_{b-1}), cos(u f_0..f_{b-1})], f_t = pi 2^t,
    b = dim_per_action // 2 (action_embedding.py:18-72).  u is `(a - low) * 2 / (high - low) + 1` clamped to [-1, 1] -- the reference's
    expression as written: in-range actions land in [1, 3] and clamp to 1 (the `+ 1` reads like a slip for `- 1`); reproduced, because
    checkpoints trained with it and the loss values of the reference are the contract."""

    def __init__(self, action_space, dim_per_action: int = 32):
        super().__init__()
        bands = dim_per_action // 2
        lo, hi = (torch.tensor(np.array(b, dtype=np.float32)) for b in (action_space.low, action_space.high))
        self.register_buffer("_action_low", lo)
        self.register_buffer("_action_high", hi)
        self.register_buffer("_freqs", math.pi * torch.pow(torch.full((bands,), 2.0), torch.arange(bands, dtype=torch.float32)))
        self._event_dims = lo.dim()
        self.n_actions = lo.numel()
        self.output_size = 2 * bands * self.n_actions

    def forward(self, action: torch.Tensor, masks=None) -> torch.Tensor:
        a = action.float()
        if masks is not None:
            a = torch.where(masks.bool(), a, torch.zeros_like(a))
        u = torch.clamp((a - self._action_low) * (2 / (self._action_high - self._action_low)) + 1, -1, 1)
        phase = torch.flatten(u, -self._event_dims)[..., None] * self._freqs  # [..., n_actions, bands]
        phase = torch.flatten(phase, -2)
        return torch.cat([torch.sin(phase), torch.cos(phase)], dim=-1)


class DiscreteActionEmbedding(nn.Module):
    """Learned table over a Discrete(n) action, shifted by one so that row 0 can stand for a masked (episode-start) step
    (action_embedding.py:75-92)."""

    def __init__(self, action_space, dim_per_action: int):
        super().__init__()
        self.embedding = nn.Embedding(int(action_space.n) + 1, dim_per_action)
        self.n_actions, self.output_size = 1, dim_per_action

    def forward(self, action: torch.Tensor, masks=None) -> torch.Tensor:
        row = action.long() + 1
        if masks is not None:
            row = torch.where(masks.bool(), row, torch.zeros_like(row))
        return F.embedding(row[..., 0], self.embedding.weight)


class ActionEmbedding(nn.Module):
    """Code of a whole action (action_embedding.py:95-146).  A task action space all of whose leaves take no arguments (habitat's
    ActionSpace of EmptySpace -- PointNav, ObjectNav) is ONE table over its `n` actions; otherwise one code per Box / Discrete leaf."""

    def __init__(self, action_space, dim_per_action: int = 32):
        super().__init__()
        leaves = list(_walk(action_space))
        if hasattr(action_space, "n") and all(_kind(s) == "empty" for s in leaves):
            leaves = [action_space]
        codes: List[nn.Module] = []
        for s in leaves:
            kind = "discrete" if s is action_space and hasattr(s, "n") else _kind(s)
            if kind not in ("box", "discrete"):
                raise RuntimeError(f"Unknown space: {s}")
            codes.append(BoxActionEmbedding(s, dim_per_action) if kind == "box" else DiscreteActionEmbedding(s, dim_per_action))
        self.embedding_modules = nn.ModuleList(codes)
        ends = np.cumsum([m.n_actions for m in codes]).tolist()
        self.embedding_slices = [slice(e - m.n_actions, e) for e, m in zip(ends, codes)]
        self._output_size = int(sum(m.output_size for m in codes))

    @property
    def output_size(self) -> int: return self._output_size

    def forward(self, action: torch.Tensor, masks=None) -> torch.Tensor:
        return torch.cat([m(action[..., cols], masks) for cols, m in zip(self.embedding_slices, self.embedding_modules)], dim=-1)
