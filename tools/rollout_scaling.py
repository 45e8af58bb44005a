#!/usr/bin/env python3
"""Device time of one rollout as a function of the number of environments (same network, same T): how much of a rollout step is latency
that a second, concurrent half-batch could hide.   usage: python tools/rollout_scaling.py [c2|c3] > gpurun_out/r06_rollout_scaling.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

for workload in (sys.argv[1:] or ["c2", "c3"]):
    for envs in (16, 32, 64, 128):
        trainer, cfg = bench.make_trainer(workload, 8, envs=envs)
        trainer._init_train()
        trainer.run_update_cycle()
        ts = []
        for _ in range(4):
            trainer._agent.pre_rollout()
            trainer._agent.eval()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            trainer.collect_rollout()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
            trainer._agent.rollouts.after_update()
        T = trainer._ppo_cfg.num_steps
        ms = sorted(ts)[len(ts) // 2]
        print(f"{workload} envs {envs:4d}: rollout {ms:7.2f} ms = {ms / T * 1e3:6.1f} us per step, {ms / T / envs * 1e3:6.2f} us per env-step", flush=True)
        trainer.envs.close()
        del trainer
        torch.cuda.empty_cache()
