#!/usr/bin/env python3
"""Development aid: error statistics (max, rms, signed mean -- all relative to the rms of the exact result) of the fp32-MFMA path and the
split-bf16 path against float64, per contraction kind."""
import ctypes as C, os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd import _lib
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.zeros(1 << 24, device="cuda")

def stats(name, ref, fn):
    out = {}
    for mode in (0, 15):
        L.hab_set_matrix_path(mode)
        y = fn().double().cpu()
        d = (y - ref) / ref.pow(2).mean().sqrt()
        out[mode] = (d.abs().max().item(), d.pow(2).mean().sqrt().item(), d.mean().item(), (d * ref.sign()).mean().item())
    for mode in (0, 15):
        print(f"{name:34s} mode {mode:2d}: max {out[mode][0]:.2e} rms {out[mode][1]:.2e} mean {out[mode][2]:+.2e} mean*sign(ref) {out[mode][3]:+.2e}")

def conv(B, H, W, Cc, Cout, K, s, p, pos=False):
    torch.manual_seed(0)
    x = torch.randn(B, Cc, H, W)
    if pos: x = x.relu()
    w = torch.randn(Cout, Cc, K, K) / np.sqrt(Cc * K * K)
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo)
    xd = x.double().requires_grad_(); wd_ = w.double().requires_grad_()
    y = F.conv2d(xd, wd_, None, stride=s, padding=p)
    y.backward(dy.double())
    xn, dyn = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    wf = w.permute(0, 2, 3, 1).contiguous().cuda()
    wdg = w.permute(1, 2, 3, 0).contiguous().cuda()
    tag = f"conv {Cc}>{Cout} k{K}s{s} B{B}@{H}" + (" relu-in" if pos else "")
    def fwd():
        o = torch.zeros(B, Ho, Wo, Cout, device="cuda")
        _lib.check(L.hab_conv2d_fwd(P(xn), P(wf), None, P(o), B, H, W, Cc, Cout, K, K, s, p, 0, P(ws), ws.numel(), S())); return o
    def dgrad():
        o = torch.zeros(B, H, W, Cc, device="cuda")
        _lib.check(L.hab_conv2d_dgrad(P(dyn), P(wdg), None, None, P(o), B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S())); return o
    def wgrad():
        o = torch.zeros(Cout, Cc, K, K, device="cuda")
        _lib.check(L.hab_conv2d_wgrad(P(xn), P(dyn), P(o), None, B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S())); return o
    stats(tag + " fwd", y.detach().permute(0, 2, 3, 1), fwd)
    stats(tag + " dgrad", xd.grad.permute(0, 2, 3, 1), dgrad)
    stats(tag + " wgrad", wd_.grad, wgrad)

conv(64, 32, 32, 32, 32, 3, 1, 1)
conv(64, 32, 32, 32, 32, 3, 1, 1, pos=True)
conv(64, 16, 16, 64, 64, 3, 1, 1, pos=True)
conv(16, 63, 63, 32, 64, 4, 2, 0, pos=True)
L.hab_set_matrix_path(15)
