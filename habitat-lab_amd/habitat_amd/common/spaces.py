"""Minimal observation / action space descriptors (the subset of gym.spaces the PPO path reads:
shape, dtype, low/high, n, Dict.spaces).  If `gym`/`gymnasium` is installed its spaces work as well --
everything here is duck-typed."""
from __future__ import annotations

import collections

import numpy as np


class Space:
    shape = None
    dtype = None


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)

    def __repr__(self):
        return f"Box{self.shape}:{self.dtype}"


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def __repr__(self):
        return f"Discrete({self.n})"


class Dict(Space):
    def __init__(self, spaces=None):
        self.spaces = collections.OrderedDict(spaces or {})

    def items(self): return self.spaces.items()
    def keys(self): return self.spaces.keys()
    def values(self): return self.spaces.values()
    def __getitem__(self, k): return self.spaces[k]
    def __contains__(self, k): return k in self.spaces
    def __iter__(self): return iter(self.spaces)
    def __len__(self): return len(self.spaces)
    def __repr__(self): return f"Dict({dict(self.spaces)})"


def is_continuous_action_space(action_space) -> bool:
    """utils/common.py:679-697: a Box action space is continuous, Discrete is not."""
    if hasattr(action_space, "n"):
        return False
    if getattr(action_space, "low", None) is not None and getattr(action_space, "shape", None) is not None:
        return True
    raise NotImplementedError(f"Unknown action space {action_space}. Is neither continuous nor discrete")


def get_num_actions(action_space) -> int:
    """utils/common.py:729-746: Discrete(n) -> n, a 1-D Box -> its length."""
    if hasattr(action_space, "n"):
        return int(action_space.n)
    if is_continuous_action_space(action_space):
        assert len(action_space.shape) == 1, f"shape was {action_space.shape} but was expecting a 1D action"
        return int(action_space.shape[0])
    raise NotImplementedError(f"unsupported action space {action_space}")


def get_action_space_info(action_space):
    """utils/common.py:701-726 -> (shape, is_discrete); Discrete pointnav -> ((1,), True), Box(A,) -> ((A,), False)."""
    if hasattr(action_space, "n"):
        return (1,), True
    if is_continuous_action_space(action_space):
        return (get_num_actions(action_space),), False
    raise NotImplementedError(f"unsupported action space {action_space}")
