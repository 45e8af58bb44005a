#!/usr/bin/env python3
"""Device time and host enqueue time of one rollout (C2 by default) with an explicit device synchronisation in front of it: says
whether the rollout is GPU-bound or host-bound on this box.  `bench.py`'s `phases.rollout_ms` brackets the rollout inside the running
pipeline instead.  Used for profiles/r05_ab_r04_vs_r05.txt (NOTEBOOK.md R5.2).   usage: python tools/rollout_probe.py [c2|c3|c5]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
trainer, cfg = bench.make_trainer(workload, 40)
trainer._init_train()
for _ in range(3):
    trainer.run_update_cycle()
torch.cuda.synchronize()
host, wall = [], []
for _ in range(8):
    trainer._agent.pre_rollout()
    trainer._agent.eval()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.collect_rollout()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    wall.append((t2 - t0) * 1e3)
    trainer._update_agent()
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps({"workload": workload, "host_enqueue_ms": round(med(host), 2), "device_ms": round(med(wall), 2),
                  "all_host": [round(h, 2) for h in host], "all_device": [round(w, 2) for w in wall]}))
