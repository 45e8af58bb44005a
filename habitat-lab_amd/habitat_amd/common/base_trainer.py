"""Trainer base classes: host control flow of habitat_baselines/common/base_trainer.py:34-338
(percent_done / is_done / should_checkpoint / resume-state cadence)."""
from __future__ import annotations

from typing import Dict

import torch

from habitat_amd.rl.ddppo.ddp_utils import SAVE_STATE, add_signal_handlers, is_slurm_batch_job


class BaseTrainer:
    supported_tasks = []

    def train(self) -> None: raise NotImplementedError
    def eval(self) -> None: raise NotImplementedError
    def save_checkpoint(self, file_name) -> None: raise NotImplementedError
    def load_checkpoint(self, checkpoint_path, *args, **kwargs) -> Dict: raise NotImplementedError

    def _add_preemption_signal_handlers(self):
        if is_slurm_batch_job():
            add_signal_handlers()


class BaseRLTrainer(BaseTrainer):
    device: torch.device

    def __init__(self, config) -> None:
        super().__init__()
        assert config is not None, "needs config file to initialize trainer"
        self.config = config
        self._flush_secs = 30
        self.num_updates_done = 0
        self.num_steps_done = 0
        self._last_checkpoint_percent = -1.0
        hb = config.habitat_baselines
        if (hb.num_updates != -1) == (hb.total_num_steps != -1):
            raise RuntimeError("Exactly one of num_updates and total_num_steps must be specified (the other -1).\n"
                               f" num_updates: {hb.num_updates} total_num_steps: {hb.total_num_steps}")
        if (hb.num_checkpoints != -1) == (hb.checkpoint_interval != -1):
            raise RuntimeError("Exactly one of num_checkpoints and checkpoint_interval must be specified (the other -1).\n"
                               f" num_checkpoints: {hb.num_checkpoints} checkpoint_interval: {hb.checkpoint_interval}")

    def percent_done(self) -> float:
        hb = self.config.habitat_baselines
        if hb.num_updates != -1:
            return self.num_updates_done / hb.num_updates
        return self.num_steps_done / hb.total_num_steps

    def is_done(self) -> bool:
        return self.percent_done() >= 1.0

    def should_checkpoint(self) -> bool:
        hb = self.config.habitat_baselines
        if hb.num_checkpoints != -1:
            every = 1 / hb.num_checkpoints
            if self._last_checkpoint_percent + every < self.percent_done():
                self._last_checkpoint_percent = self.percent_done()
                return True
            return False
        return (self.num_updates_done % hb.checkpoint_interval) == 0

    def _should_save_resume_state(self) -> bool:
        pre = self.config.habitat_baselines.rl.preemption
        return SAVE_STATE.is_set() or ((not pre.save_state_batch_only or is_slurm_batch_job())
                                       and (int(self.num_updates_done + 1) % pre.save_resume_state_interval) == 0)

    @property
    def flush_secs(self): return self._flush_secs

    @flush_secs.setter
    def flush_secs(self, value: int): self._flush_secs = value
