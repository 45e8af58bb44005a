"""Mean over the last `window_size` scalar samples (interface of habitat_baselines/common/windowed_running_mean.py:
add / mean / sum / count / __len__)."""
from __future__ import annotations

from collections import deque
from typing import Union

import torch


class WindowedRunningMean:
    def __init__(self, window_size: int):
        self.window_size = int(window_size)
        self._vals: deque = deque(maxlen=self.window_size)
        self.sum = 0.0

    def add(self, val: Union[float, int, torch.Tensor]) -> None:
        val = float(val)
        if len(self._vals) == self.window_size:
            self.sum -= self._vals[0]
        self._vals.append(val)
        self.sum += val

    def __iadd__(self, val):
        self.add(val)
        return self

    @property
    def count(self) -> int:
        return len(self._vals)

    def __len__(self) -> int:
        return len(self._vals)

    @property
    def mean(self) -> float:
        return self.sum / max(len(self._vals), 1)
