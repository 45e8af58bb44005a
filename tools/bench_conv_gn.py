#!/usr/bin/env python3
"""Micro-benchmark of the fused convolution + GroupNorm kernel (csrc/conv_gn_slab.h) at the ResNet18 / ResNet50 rollout shapes
(development aid).  HIP-event timing per launch; HAB_CGS_ABLATE (1 no input staging, 2 no MFMAs, 4 no weight loads, 8 no statistics)
shows where the time goes.  usage: python tools/bench_conv_gn.py [B]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
CASES = [("l2 3x3 64>64 @16", 16, 16, 64, 64, 3, 1, 1, 16), ("l3.0 3x3s2 64>128", 16, 16, 64, 128, 3, 2, 1, 16),
         ("l3 3x3 128>128 @8", 8, 8, 128, 128, 3, 1, 1, 16), ("l3.0 ds 1x1s2 64>128", 16, 16, 64, 128, 1, 2, 0, 16),
         ("l4.0 3x3s2 128>256", 8, 8, 128, 256, 3, 2, 1, 16), ("l4 3x3 256>256 @4", 4, 4, 256, 256, 3, 1, 1, 16),
         ("comp 3x3 256>128 @4", 4, 4, 256, 128, 3, 1, 1, 1), ("r50 l4 1x1 256>1024", 4, 4, 256, 1024, 1, 1, 0, 16),
         # conv1x1_gn_stream.h: the bottleneck net's 1x1 layers whose frames do not fit LDS
         ("r50 l1 1x1 32>32 @32", 32, 32, 32, 32, 1, 1, 0, 16), ("r50 l1 1x1 32>128 @32", 32, 32, 32, 128, 1, 1, 0, 16),
         ("r50 l1 1x1 128>32 @32", 32, 32, 128, 32, 1, 1, 0, 16), ("r50 l2.0 1x1 128>64 @32", 32, 32, 128, 64, 1, 1, 0, 16),
         ("r50 l2.0 ds 1x1s2 128>256", 32, 32, 128, 256, 1, 2, 0, 16), ("r50 l2 1x1 256>64 @16", 16, 16, 256, 64, 1, 1, 0, 16),
         ("r50 l2 1x1 64>256 @16", 16, 16, 64, 256, 1, 1, 0, 16)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    print(f"B = {B}, HAB_CGS_ABLATE = {os.environ.get('HAB_CGS_ABLATE', '0')}")
    for name, H, W, Cc, Cout, K, s, p, G in CASES:
        Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
        x = torch.randn(B, H, W, Cc, device="cuda")
        wf = torch.randn(Cout, K, K, Cc, device="cuda") * 0.05
        pl = torch.zeros(3 * wf.numel(), dtype=torch.int16, device="cuda")
        _lib.check(L.hab_split_weight_planes(P(wf), Cout, K * K * Cc, P(pl), S()))
        g, b = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
        res = torch.randn(B, Ho, Wo, Cout, device="cuda")
        y = torch.empty(B, Ho, Wo, Cout, device="cuda")
        ys = [torch.empty_like(y) for _ in range(4)]

        def run(i):
            _lib.check(L.hab_conv_gn_fwd(P(x), P(pl), P(g), P(b), P(res), P(ys[i % 4]), None, None, None, B, H, W, Cc, Cout, K, K, s, p, G, 1,
                                         1e-5, S()))

        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            run(i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20))
        fl = 2.0 * B * Ho * Wo * Cout * K * K * Cc
        # the unfused pair on the same inputs (contraction [+ split-K second pass] + GroupNorm [statistics + apply])
        ws = torch.empty(1 << 24, device="cuda")
        raw = torch.empty_like(y)
        m_, r_ = torch.empty(B, G, device="cuda"), torch.empty(B, G, device="cuda")

        def run2(i):
            _lib.check(L.hab_conv2d_fwd(P(x), P(wf), None, P(raw), B, H, W, Cc, Cout, K, K, s, p, 0, P(ws), ws.numel(), S()))
            return L.hab_groupnorm_fwd(P(raw), P(ys[i % 4]), P(g), P(b), P(res), P(m_), P(r_), B, Ho * Wo, Cout, G, 1, 1e-5, P(ws), ws.numel(), S())

        un = float("nan")
        if run2(0) == 0:
            run2(1)
            torch.cuda.synchronize()
            ev[0].record()
            for i in range(20):
                run2(i)
                ev[i + 1].record()
            torch.cuda.synchronize()
            un = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20))[10]
        print(f"{name:26s} fused median {ts[10]:7.1f} us  min {ts[0]:7.1f} us   {fl / ts[10] / 1e6:7.1f} TF/s-eq    unfused pair {un:7.1f} us")


if __name__ == "__main__":
    main()
