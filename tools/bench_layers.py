#!/usr/bin/env python3
"""Per-layer micro-benchmark of the igemm kernels at the benchmark shapes (development aid).
Prints achieved TFLOP/s per contraction with HIP-event timing.  usage: python tools/bench_layers.py [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def conv_case(name, B, H, W, Cc, Cout, K, s, p, ws):
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    x = torch.randn(B, H, W, Cc, device="cuda")
    wf = torch.randn(Cout, K, K, Cc, device="cuda") * 0.05
    wd = torch.randn(Cc, K, K, Cout, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty(B, Ho, Wo, Cout, device="cuda")
    dy = torch.randn(B, Ho, Wo, Cout, device="cuda")
    dx = torch.empty(B, H, W, Cc, device="cuda")
    dw = torch.empty(Cout, Cc, K, K, device="cuda")
    fl = 2.0 * B * Ho * Wo * Cout * K * K * Cc
    t = timeit(lambda: _lib.check(L.hab_conv2d_fwd(P(x), P(wf), P(b), P(y), B, H, W, Cc, Cout, K, K, s, p, 1, P(ws), ws.numel(), S())))
    print(f"{name:28s} fwd   {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    t = timeit(lambda: _lib.check(L.hab_conv2d_dgrad(P(dy), P(wd), None, None, P(dx), B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S())))
    print(f"{name:28s} dgrad {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (algorithmic)")
    mask = torch.randn(B, H, W, Cc, device="cuda")
    t = timeit(lambda: _lib.check(L.hab_conv2d_dgrad(P(dy), P(wd), P(mask), None, P(dx), B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S())))
    print(f"{name:28s} dgrad {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (with the ReLU-mask epilogue)")
    t = timeit(lambda: _lib.check(L.hab_conv2d_wgrad(P(x), P(dy), P(dw), P(b), B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S())))
    print(f"{name:28s} wgrad {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ws = torch.empty(1 << 26, device="cuda")
    H = W = 256
    rgb = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
    depth = torch.rand(B, H, W, 1, device="cuda")
    wf = torch.randn(32, 8, 8, 4, device="cuda") * 0.05
    b = torch.zeros(32, device="cuda")
    y = torch.empty(B, 63, 63, 32, device="cuda")
    fl = 2.0 * B * 63 * 63 * 32 * 256
    t = timeit(lambda: _lib.check(L.hab_obs_conv2d_fwd(P(rgb), P(depth), None, P(wf), P(b), P(y), B, H, W, 32, 8, 8, 4, 0, 1, P(ws), ws.numel(), S())))
    print(f"{'simplecnn conv1 (obs ingest)':28s} fwd   {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s  obs {B * 458752 / t / 1e6:6.1f} GB/s")
    dw = torch.empty(32, 4, 8, 8, device="cuda")
    t = timeit(lambda: _lib.check(L.hab_obs_conv2d_wgrad(P(rgb), P(depth), None, P(y), P(dw), P(b), B, H, W, 32, 8, 8, 4, 0, P(ws), ws.numel(), S())))
    print(f"{'simplecnn conv1 (obs ingest)':28s} wgrad {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    conv_case("simplecnn conv2 4x4s2 32>64", B, 63, 63, 32, 64, 4, 2, 0, ws)
    conv_case("simplecnn conv3 3x3s1 64>32", B, 30, 30, 64, 32, 3, 1, 0, ws)
    conv_case("resnet l1 3x3 32>32 @32", B, 32, 32, 32, 32, 3, 1, 1, ws)
    conv_case("resnet l2 3x3 64>64 @16", B, 16, 16, 64, 64, 3, 1, 1, ws)
    conv_case("resnet l3 3x3 128>128 @8", B, 8, 8, 128, 128, 3, 1, 1, ws)
    conv_case("resnet l4 3x3 256>256 @4", B, 4, 4, 256, 256, 3, 1, 1, ws)
    # fc 25088 -> 512
    M, N, K = B, 512, 25088
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.01
    yy = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    t = timeit(lambda: _lib.check(L.hab_linear_fwd(P(x), K, P(w), K, None, P(yy), N, M, N, K, 1, 0, P(ws), ws.numel(), S())))
    print(f"{'fc 25088>512':28s} fwd   {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    dxx = torch.empty(M, K, device="cuda")
    t = timeit(lambda: _lib.check(L.hab_linear_dgrad(P(yy), N, P(w), K, None, 0, P(dxx), K, M, K, N, 0, P(ws), ws.numel(), S())))
    print(f"{'fc 25088>512':28s} dgrad {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    dww = torch.empty(N, K, device="cuda")
    t = timeit(lambda: _lib.check(L.hab_linear_wgrad(P(yy), N, P(x), K, P(dww), K, M, N, K, 0, 0, 0, P(ws), ws.numel(), S())))
    print(f"{'fc 25088>512':28s} wgrad {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    M = 64
    x = torch.randn(M, K, device="cuda")
    yy = torch.empty(M, N, device="cuda")
    t = timeit(lambda: _lib.check(L.hab_linear_fwd(P(x), K, P(w), K, None, P(yy), N, M, N, K, 1, 0, P(ws), ws.numel(), S())))
    print(f"{'fc 25088>512 (M=64, rollout)':28s} fwd   {t:8.3f} ms  weights {N * K * 4 / t / 1e6:6.1f} GB/s")


if __name__ == "__main__":
    main()
