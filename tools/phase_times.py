#!/usr/bin/env python3
"""Wall time of the two phases of an update cycle (rollout collection, GAE + PPO update) with device syncs in between
(development aid).  usage: python tools/phase_times.py [c2|c3|c5] [cycles]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tr, cfg = bench.make_trainer(wl, n + 2)
    tr._init_train()
    tr.run_update_cycle()
    torch.cuda.synchronize()
    tot = [0.0, 0.0, 0.0]
    for _ in range(n):
        tr._agent.pre_rollout()
        tr._agent.eval()
        t0 = time.perf_counter()
        steps = tr.collect_rollout()
        t1 = time.perf_counter()      # host done enqueuing
        torch.cuda.synchronize()
        t2 = time.perf_counter()      # device done
        losses = tr._update_agent()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        tr.num_updates_done += 1
        tr._coalesce_post_step(losses, steps)
        tot[0] += t1 - t0; tot[1] += t2 - t0; tot[2] += t3 - t2
    print(f"{wl}: rollout host-enqueue {tot[0] / n * 1e3:.1f} ms, rollout until device idle {tot[1] / n * 1e3:.1f} ms, "
          f"GAE + update {tot[2] / n * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
