#!/usr/bin/env python3
"""conv3 (3x3/1 64->32 at 30x30) and ResNet layer1 (3x3/1/1 32->32 at 32x32) forward / data gradient at rollout and learner batch sizes
(development aid).  usage: python tools/bench_conv3.py"""
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


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


ws = torch.empty(1 << 26, device="cuda")
for name, H, Cc, Cout, pad in (("conv3 64>32 @30", 30, 64, 32, 0), ("layer1 32>32 @32", 32, 32, 32, 1)):
    for B in (64, 512, 1024):
        Ho = H + 2 * pad - 2
        x = torch.randn(B, H, H, Cc, device="cuda")
        wf = torch.randn(Cout, 3, 3, Cc, device="cuda") * 0.05
        wd = torch.randn(Cc, 3, 3, Cout, device="cuda") * 0.05
        b = torch.zeros(Cout, device="cuda")
        y = torch.empty(B, Ho, Ho, Cout, device="cuda")
        dx = torch.empty(B, H, H, Cc, device="cuda")
        fl = 2.0 * B * Ho * Ho * Cout * 9 * Cc
        t = timeit(lambda: _lib.check(L.hab_conv2d_fwd(P(x), P(wf), P(b), P(y), B, H, H, Cc, Cout, 3, 3, 1, pad, 1, P(ws), ws.numel(), S())))
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wf.permute(0, 3, 1, 2).double(), padding=pad).relu().permute(0, 2, 3, 1)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        line = f"{name:18s} B {B:5d} fwd {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TF/s  rel err {err:.1e}"
        if Cc == Cout:
            t = timeit(lambda: _lib.check(L.hab_conv2d_dgrad(P(y), P(wd), None, None, P(dx), B, H, H, Cc, Cout, 3, 3, 1, pad, P(ws), ws.numel(), S())))
            line += f" | dgrad {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TF/s"
        print(line, flush=True)
