"""TensorDict: a nested dict of tensors that can be indexed like one tensor.

Same public behaviour as habitat_baselines/common/tensor_dict.py:57-405 (str key -> entry, any
other index -> a TensorDict of the indexed leaves; `set` with strict key matching; map / map_in_place;
from_tree / to_tree; flatten / from_flattened), written independently."""
from __future__ import annotations

import numbers
from typing import Any, Callable, Dict, List, Tuple, Union

import numpy as np
import torch

DictTree = Dict[str, Any]


def _as_tensor(v):
    if isinstance(v, torch.Tensor):
        return v
    if isinstance(v, np.ndarray):
        return torch.from_numpy(v)
    return torch.as_tensor(v)


class TensorDict(dict):
    @classmethod
    def from_tree(cls, tree: DictTree) -> "TensorDict":
        out = cls()
        for k, v in tree.items():
            out[k] = cls.from_tree(v) if isinstance(v, dict) else _as_tensor(v)
        return out

    def to_tree(self) -> DictTree:
        return {k: (v.to_tree() if isinstance(v, TensorDict) else v) for k, v in self.items()}

    def flatten(self) -> Tuple[List[Tuple[str, ...]], List[torch.Tensor]]:
        spec, leaves = [], []
        for k, v in self.items():
            if isinstance(v, TensorDict):
                s, l = v.flatten()
                spec += [(k,) + x for x in s]
                leaves += l
            else:
                spec.append((k,))
                leaves.append(v)
        return spec, leaves

    @classmethod
    def from_flattened(cls, spec, leaves) -> "TensorDict":
        out = cls()
        for path, leaf in zip(spec, leaves):
            node = out
            for key in path[:-1]:
                if key not in node:
                    dict.__setitem__(node, key, cls())
                node = dict.__getitem__(node, key)
            if path[-1] in node:
                raise RuntimeError(f"Key '{path[-1]}' already in the tree. Invalid spec.")
            dict.__setitem__(node, path[-1], _as_tensor(leaf))
        return out

    def __getitem__(self, index):
        if isinstance(index, str):
            return dict.__getitem__(self, index)
        return type(self)((k, v[index]) for k, v in self.items())

    def set(self, index, value, strict: bool = True) -> None:
        if isinstance(index, str):
            if isinstance(value, dict) and not isinstance(value, TensorDict):
                value = self.from_tree(value)
            elif not isinstance(value, TensorDict):
                value = _as_tensor(value)
            dict.__setitem__(self, index, value)
            return
        assert isinstance(value, dict)
        if strict and self.keys() != value.keys():
            raise KeyError(f"Keys don't match: Dest={self.keys()} Source={value.keys()}")
        for k in self.keys():
            if k not in value:
                if strict:
                    raise KeyError(f"Key {k} not in new value dictionary")
                continue
            v, dst = value[k], dict.__getitem__(self, k)
            if isinstance(v, dict):
                assert isinstance(dst, TensorDict)
                dst.set(index, v, strict=strict)
            else:
                assert not isinstance(dst, TensorDict)
                dst[index] = _as_tensor(v)

    def __setitem__(self, index, value):
        self.set(index, value)

    def map_func(self, func: Callable, prefix: str = "") -> "TensorDict":
        out = type(self)()
        for k, v in self.items():
            dict.__setitem__(out, k, v.map_func(func, f"{prefix}{k}.") if isinstance(v, TensorDict) else func(v))
        return out

    def map(self, func: Callable[[torch.Tensor], torch.Tensor]) -> "TensorDict":
        return self.map_func(func)

    def map_in_place(self, func: Callable[[torch.Tensor], torch.Tensor]) -> "TensorDict":
        for k, v in self.items():
            if isinstance(v, TensorDict):
                v.map_in_place(func)
            else:
                dict.__setitem__(self, k, func(v))
        return self

    def zip(self, *others):
        out = type(self)()
        for k, v in self.items():
            if isinstance(v, TensorDict):
                dict.__setitem__(out, k, v.zip(*[o[k] for o in others]))
            else:
                dict.__setitem__(out, k, (v,) + tuple(o[k] for o in others))
        return out

    def __deepcopy__(self, memo=None):
        return self.map(lambda t: t.clone())
