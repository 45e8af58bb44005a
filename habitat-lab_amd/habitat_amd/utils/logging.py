"""Logger + scalar writer.  `get_writer` mirrors habitat_baselines/common/tensorboard_utils.py:21-172: a TensorBoard writer when
the `tensorboard` package is importable and a directory is configured, otherwise a writer that keeps the last value of every
scalar in memory (`.scalars`) -- logging sinks are not on the accelerated path."""
from __future__ import annotations

import logging
import os
import sys

logger = logging.getLogger("habitat_amd")
if not logger.handlers:
    _h = logging.StreamHandler(sys.stderr)
    _h.setFormatter(logging.Formatter("%(asctime)s %(message)s"))
    logger.addHandler(_h)
    logger.setLevel(os.environ.get("HABITAT_BASELINES_LOG", "INFO"))


class ScalarWriter:
    def __init__(self, log_dir=None, flush_secs=30):
        self.scalars = {}
        self._tb = None
        if log_dir:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self._tb = SummaryWriter(log_dir, flush_secs=flush_secs)
            except Exception:
                self._tb = None

    def add_scalar(self, tag, value, step=None):
        self.scalars[tag] = (float(value), step)
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)

    def get_run_id(self):
        return None

    def close(self):
        if self._tb is not None:
            self._tb.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def get_writer(config, **kwargs):
    d = config.habitat_baselines.tensorboard_dir if config.habitat_baselines.get("writer_type", "tb") == "tb" else None
    return ScalarWriter(d if d and os.environ.get("HABITAT_AMD_TENSORBOARD", "0") == "1" else None, kwargs.get("flush_secs", 30))
