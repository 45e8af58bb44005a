#!/usr/bin/env python3
"""CLI entry point with the shape of habitat_baselines/run.py:24-62:

    python -m habitat_amd.run --config-name=pointnav/ddppo_pointnav.yaml habitat_baselines.num_environments=64
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m habitat_amd.run --config-name=pointnav/ddppo_pointnav.yaml

(equivalent of the reference's rl/ddppo/single_node.sh: one process per GPU, RCCL over xGMI)."""
from __future__ import annotations

import argparse
import random

import numpy as np
import torch

from habitat_amd.common.baseline_registry import baseline_registry
from habitat_amd.config.default import get_config


def execute_exp(config, run_type: str) -> None:
    random.seed(config.habitat.seed)
    np.random.seed(config.habitat.seed)
    torch.manual_seed(config.habitat.seed)
    if config.habitat_baselines.force_torch_single_threaded and torch.cuda.is_available():
        torch.set_num_threads(1)
    import habitat_amd.rl.ppo.ppo_trainer  # noqa: F401  (registers "ppo" / "ddppo")
    import habitat_amd.rl.ver.ver_trainer  # noqa: F401  (registers "ver")
    trainer_init = baseline_registry.get_trainer(config.habitat_baselines.trainer_name)
    assert trainer_init is not None, f"{config.habitat_baselines.trainer_name} is not supported"
    trainer = trainer_init(config)
    if run_type == "train":
        trainer.train()
    elif run_type == "eval":
        trainer.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-name", required=True)
    ap.add_argument("--run-type", default="train", choices=["train", "eval"])
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    cfg = get_config(a.config_name, a.overrides)
    cfg.habitat_baselines.evaluate = a.run_type == "eval"
    execute_exp(cfg, a.run_type)


if __name__ == "__main__":
    main()
