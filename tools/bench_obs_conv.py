#!/usr/bin/env python3
"""One call site alone (profiling aid): the observation-ingest convolution forward at B frames.  usage: bench_obs_conv.py [B] [iters]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd import _lib
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = W = 256
rgb = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
depth = torch.rand(B, H, W, 1, device="cuda")
wf = torch.randn(32, 8, 8, 4, device="cuda") * 0.05
b = torch.zeros(32, device="cuda")
y = torch.empty(B, 63, 63, 32, device="cuda")
ws = torch.empty(1 << 24, device="cuda")
for _ in range(it):
    _lib.check(L.hab_obs_conv2d_fwd(P(rgb), P(depth), None, P(wf), P(b), P(y), B, H, W, 32, 8, 8, 4, 0, 1, P(ws), ws.numel(), S()))
torch.cuda.synchronize()
