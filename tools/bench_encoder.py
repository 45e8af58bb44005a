#!/usr/bin/env python3
"""ResNet18 visual encoder (a5) at the north-star batch: frames/s and fraction of the fp32 MFMA roofline for
forward and forward+backward, timed with HIP events around the engine's encoder call sites.

  python tools/bench_encoder.py [frames_per_call=4096] [calls=2]      (8192 frames = 2 minibatches of 4096)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd.engine import DevicePackInfo, PolicyEngine  # noqa: E402

PEAK = 157.3e12
F_FWD, F_BWD = 2 * 168.82e6, 2 * (2 * 168.82e6 - 25.69e6)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    T = 128
    n = B // T
    eng = PolicyEngine(arch="resnet", backbone=18, baseplanes=32, normalize_visual_inputs=True, rnn_type="LSTM", rnn_layers=2, hidden=512,
                       H=256, W=256, max_frames=B, max_envs=64)
    g = torch.Generator(device="cuda").manual_seed(0)
    eng.params_flat.copy_(torch.randn(eng.params_flat.shape, device="cuda", generator=g) * 0.05)
    for nm, v in eng.views.items():
        if nm.endswith(".1.weight") or nm.endswith(".4.weight") or nm.endswith(".7.weight"):
            if v.dim() == 1:
                v.fill_(1.0)
    eng.repack()
    rgb = torch.randint(0, 256, (B, 256, 256, 3), dtype=torch.uint8, device="cuda", generator=g)
    depth = torch.rand(B, 256, 256, 1, device="cuda", generator=g)
    goal = torch.rand(B, 2, device="cuda", generator=g)
    masks = torch.ones(B, 1, dtype=torch.bool, device="cuda")
    actions = torch.zeros(B, 1, dtype=torch.long, device="cuda")
    h0 = torch.zeros(n, 4, 512, device="cuda")
    pack = DevicePackInfo(np.zeros((T, n), np.uint8), "cuda")
    dv = torch.randn(B, device="cuda", generator=g) * 1e-3

    def cycle():
        eng.evaluate(rgb, depth, goal, None, h0, masks, actions, pack, B, n, prev_actions=actions)
        eng.backward(rgb, depth, goal, None, actions, pack, dv, dv, dv, prev_actions=actions)

    cycle()
    torch.cuda.synchronize()
    res = {}
    for tag, name in ((11, "fwd"), (12, "bwd")):
        eng.probe_enable(tag)
        for _ in range(calls):
            cycle()
        torch.cuda.synchronize()
        ms, cnt = eng.probe_read()
        res[name] = ms / cnt
    eng.probe_enable(-1)
    fwd, bwd = res["fwd"], res["bwd"]
    print(f"resnet18 encoder, {B} frames/call x {calls} calls (256x256 RGB-D, fp32):")
    print(f"  forward          {fwd:9.2f} ms  {B / fwd * 1e3:10.0f} frames/s  {B * F_FWD / fwd / 1e9:7.1f} TFLOP/s  {B * F_FWD / fwd * 1e3 / PEAK:6.1%} of fp32 MFMA peak")
    print(f"  backward         {bwd:9.2f} ms  {B / bwd * 1e3:10.0f} frames/s  {B * F_BWD / bwd / 1e9:7.1f} TFLOP/s  {B * F_BWD / bwd * 1e3 / PEAK:6.1%}")
    tot = fwd + bwd
    print(f"  forward+backward {tot:9.2f} ms  {B / tot * 1e3:10.0f} frames/s  {B * (F_FWD + F_BWD) / tot / 1e9:7.1f} TFLOP/s  {B * (F_FWD + F_BWD) / tot * 1e3 / PEAK:6.1%}")


if __name__ == "__main__":
    main()
