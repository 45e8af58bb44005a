#!/usr/bin/env python3
"""One conv layer (fwd, dgrad, wgrad) at a given shape, a few launches each -- target for rocprofv3 --pmc runs.
usage: python tools/bench_one.py B H W C Cout K stride pad [iters]"""
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
B, H, W, Cc, Cout, K, s, p = [int(v) for v in sys.argv[1:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 3
Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
ws = torch.empty(1 << 26, device="cuda")
x = torch.randn(B, H, W, Cc, device="cuda")
wf = torch.randn(Cout, K, K, Cc, device="cuda") * 0.05
wd = torch.randn(Cc, K, K, Cout, device="cuda") * 0.05
y = torch.empty(B, Ho, Wo, Cout, device="cuda")
dy = torch.randn(B, Ho, Wo, Cout, device="cuda")
dx = torch.empty(B, H, W, Cc, device="cuda")
dw = torch.empty(Cout, Cc, K, K, device="cuda")
for _ in range(iters):
    _lib.check(L.hab_conv2d_fwd(P(x), P(wf), None, P(y), B, H, W, Cc, Cout, K, K, s, p, 0, P(ws), ws.numel(), S()))
    _lib.check(L.hab_conv2d_dgrad(P(dy), P(wd), None, None, P(dx), B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S()))
    _lib.check(L.hab_conv2d_wgrad(P(x), P(dy), P(dw), None, B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S()))
torch.cuda.synchronize()
