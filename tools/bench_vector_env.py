#!/usr/bin/env python3
"""Transport throughput of the process-per-env VectorEnv: env-steps/s for N host envs producing 256x256 RGB-D,
shared-memory observation plane vs the pipe-pickling data path of the reference's VectorEnv
(habitat/core/vector_env.py:402-410 + habitat_baselines/utils/common.py:244-310 batch_obs).  Runs on the host alone; with a GPU
the batches are also uploaded.  usage: python tools/bench_vector_env.py [num_envs] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd.core.host_env import make_host_env  # noqa: E402
from habitat_amd.core.vector_env import VectorEnv  # noqa: E402


def stack_batch(observations, device):
    """The reference's batching: per-sensor stack of the per-env arrays, then upload."""
    out = {}
    for k in sorted(observations[0], key=lambda k: -np.asarray(observations[0][k]).nbytes):
        t = torch.from_numpy(np.stack([np.asarray(o[k]) for o in observations]))
        out[k] = t.pin_memory().to(device, non_blocking=True) if device.type == "cuda" else t
    return out


def run(n, steps, shared):
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    args = [(i, 256, 256, True, True, 4, 500, 0) for i in range(n)]
    with VectorEnv(make_host_env, args, shared_obs=shared) as envs:
        envs.reset()
        for it in range(steps + 2):
            if it == 2:
                t0 = time.perf_counter()
            for i in range(n):
                envs.async_step_at(i, 1)
            outs = [envs.wait_step_at(i) for i in range(n)]
            batch = envs.batched_obs(slice(0, n), device) if shared else stack_batch([o[0] for o in outs], device)
            if device.type == "cuda":
                torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del batch, outs
    return n * steps / dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    env_only = make_host_env(0, 256, 256, True, True, 4, 500)
    env_only.reset()
    t0 = time.perf_counter()
    for _ in range(50):
        env_only.step(1)
    gen = (time.perf_counter() - t0) / 50
    print(f"one env, in process: {gen * 1e3:.2f} ms / step (observation generation = the 'simulator' here)")
    for shared in (False, True):
        sps = run(n, steps, shared)
        print(f"{n} worker processes, {'shared-memory slab' if shared else 'pipe + pickle + np.stack'}: {sps:9.1f} env-steps/s "
              f"({sps * 458752 / 1e9:.2f} GB/s of observations)")


if __name__ == "__main__":
    main()
