"""`g_timer.avg_time(key, level)` context manager / decorator with the keys of the reference
(habitat_baselines/utils/timing.py:17-103; level gated by HABITAT_TIMING_LEVEL) -- wall-clock windowed means that the
trainer exports as perf/<key>."""
from __future__ import annotations

import os
import time
from collections import deque
from functools import wraps


class _Mean:
    def __init__(self, window=100):
        self.q = deque(maxlen=window)

    def add(self, v): self.q.append(v)

    @property
    def mean(self): return sum(self.q) / max(1, len(self.q))


class _Ctx:
    def __init__(self, timer, key, active):
        self.timer, self.key, self.active = timer, key, active

    def __enter__(self):
        if self.active:
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.timer._stats.setdefault(self.key, _Mean()).add(time.perf_counter() - self.t0)
        return False

    def __call__(self, fn):
        @wraps(fn)
        def inner(*a, **k):
            with _Ctx(self.timer, self.key, self.active):
                return fn(*a, **k)
        return inner


class Timer:
    def __init__(self):
        self._stats = {}
        self._level = int(os.environ.get("HABITAT_TIMING_LEVEL", 0))

    def avg_time(self, key, level=0):
        return _Ctx(self, key, level <= self._level)

    def items(self): return self._stats.items()
    def __getitem__(self, k): return self._stats[k]


g_timer = Timer()

Timing = Timer  # the per-worker timers of rl/ver (habitat_baselines/utils/timing.py's class name)
