"""Trainer base classes: host control flow of habitat_baselines/common/base_trainer.py:34-338
(percent_done / is_done / should_checkpoint / resume-state cadence)."""
from __future__ import annotations

import copy
import glob
import os
import time
from typing import Dict, List, Optional

import torch

from habitat_amd.rl.ddppo.ddp_utils import (SAVE_STATE, add_signal_handlers, is_slurm_batch_job, load_resume_state,
                                             save_resume_state)
from habitat_amd.utils.logging import get_writer, logger


def get_checkpoint_id(ckpt_path: str) -> Optional[int]:
    """utils/common.py:333-347: the ID of `ckpt.ID.pth`."""
    nums: List[int] = [int(s) for s in os.path.basename(ckpt_path).split(".") if s.isdigit()]
    return nums[-1] if nums else None


def poll_checkpoint_folder(checkpoint_folder: str, previous_ckpt_ind: int) -> Optional[str]:
    """utils/common.py:350-377: the (previous_ckpt_ind + 1)-th checkpoint by modification time, or None."""
    assert os.path.isdir(checkpoint_folder), f"invalid checkpoint folder path {checkpoint_folder}"
    paths = [p for p in glob.glob(checkpoint_folder + "/*") if os.path.isfile(p) and "latest" not in p
             and not os.path.basename(p).startswith(".")]
    paths.sort(key=os.path.getmtime)
    ind = previous_ckpt_ind + 1
    return paths[ind] if ind < len(paths) else None


class BaseTrainer:
    supported_tasks = []

    def train(self) -> None: raise NotImplementedError
    def save_checkpoint(self, file_name) -> None: raise NotImplementedError
    def load_checkpoint(self, checkpoint_path, *args, **kwargs) -> Dict: raise NotImplementedError

    def _add_preemption_signal_handlers(self):
        if is_slurm_batch_job():
            add_signal_handlers()

    def _get_resume_state_config_or_new_config(self, resume_state_config):
        """base_trainer.py:46-60."""
        if self.config.habitat_baselines.load_resume_state_config and resume_state_config is not None:
            if self.config != resume_state_config:
                logger.warning("resuming with the ORIGINAL configuration of the run (load_resume_state_config=True); "
                               "the new configuration is ignored")
            return type(self.config).wrap(resume_state_config) if isinstance(resume_state_config, dict) else resume_state_config
        return copy.deepcopy(self.config)

    def eval(self) -> None:
        """base_trainer.py:66-168: evaluate `eval_ckpt_path_dir` (one file) or every checkpoint of the folder in order of
        creation, polling for new ones; progress survives preemption through the 'eval' resume state."""
        self._add_preemption_signal_handlers()
        hb = self.config.habitat_baselines
        resume_state = load_resume_state(self.config, filename_key="eval")
        if resume_state is not None:
            self.config = self._get_resume_state_config_or_new_config(resume_state["config"])
            prev_ckpt_ind = resume_state["prev_ckpt_ind"]
        else:
            prev_ckpt_ind = -1
        self.device = torch.device("cuda", hb.torch_gpu_id) if torch.cuda.is_available() else torch.device("cpu")
        with get_writer(self.config, flush_secs=self.flush_secs) as writer:
            if os.path.isfile(hb.eval_ckpt_path_dir) or not hb.eval.should_load_ckpt:
                idx = get_checkpoint_id(hb.eval_ckpt_path_dir) if hb.eval.should_load_ckpt else None
                self._eval_checkpoint(hb.eval_ckpt_path_dir, writer, checkpoint_index=idx if idx is not None else 0)
                return
            while True:
                current_ckpt = None
                while current_ckpt is None:
                    current_ckpt = poll_checkpoint_folder(hb.eval_ckpt_path_dir, prev_ckpt_ind)
                    if current_ckpt is None:
                        time.sleep(2)
                logger.info(f"=======current_ckpt: {current_ckpt}=======")
                prev_ckpt_ind += 1
                self._eval_checkpoint(checkpoint_path=current_ckpt, writer=writer, checkpoint_index=prev_ckpt_ind)
                save_resume_state({"config": self.config.to_dict() if hasattr(self.config, "to_dict") else self.config,
                                   "prev_ckpt_ind": prev_ckpt_ind}, self.config, filename_key="eval")
                if (prev_ckpt_ind + 1) == hb.num_checkpoints:
                    break

    def _eval_checkpoint(self, checkpoint_path: str, writer, checkpoint_index: int = 0) -> None:
        raise NotImplementedError


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
