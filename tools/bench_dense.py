"""GPU: the large Linear contractions through the C ABI (hab_linear_fwd / _dgrad / _wgrad) on the plain dense GEMM kernel
(csrc/dense_bf3.h, matrix-path bit 10) vs the igemm kernels: error against float64 and HIP-event time per call.

  python tools/bench_dense.py [--quick]   > gpurun_out/r06_dense_ab.txt
Shapes: SimpleCNN's visual fc at C2 (25088 -> 512): forward / data gradient per time chunk of 512 frames, weight gradient per 2048-frame
minibatch; plus the 2048-frame forms and the ResNet policies' recurrent input projection (4096 x 2048 x 576).
"""
import ctypes as C
import os
import sys

import torch

os.environ.setdefault("HAB_DENSE_MIN_MFLOP", "0")  # every applicable shape takes the dense kernel (the product default is 4000)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
PEAK_EQ = 2500.0 / 6.0


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def case(kind, M, N, K, reps, perm=False):
    """kind fwd: y[M][N] = relu(x[M][K] w[N][K]^T + b); dgrad: dx[M][K] = dy[M][N] w[N][K]; wgrad: dw[N][K] = dy[M][N]^T x[M][K]"""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    dy = torch.randn(M, N, device="cuda", generator=g)
    ws = torch.zeros(1 << 26, device="cuda")
    out = {}
    for mode_name, mode in (("igemm", 1023), ("dense", 4095 if "--force" in sys.argv else 2047)):
        L.hab_set_matrix_path(mode)
        if kind == "fwd":
            y = torch.zeros(M, N, device="cuda")
            fn = lambda: L.hab_linear_fwd(P(x), K, P(w), K, P(b), P(y), N, M, N, K, 1, 0, P(ws), ws.numel(), S())
            assert fn() == 0
            ref = torch.relu(x.double() @ w.double().t() + b.double())
            err = rel(y, ref)
        elif kind == "dgrad":
            dx = torch.zeros(M, K, device="cuda")
            fn = lambda: L.hab_linear_dgrad(P(dy), N, P(w), K, None, 0, P(dx), K, M, K, N, 0, P(ws), ws.numel(), S())
            assert fn() == 0
            err = rel(dx, dy.double() @ w.double())
        else:
            dw = torch.zeros(N, K, device="cuda")
            pc, ph = (32, K // 32) if perm else (0, 0)
            fn = lambda: L.hab_linear_wgrad(P(dy), N, P(x), K, P(dw), K, M, N, K, pc, ph, 0, P(ws), ws.numel(), S())
            assert fn() == 0
            ref = dy.double().t() @ x.double()
            if perm:  # column hw * 32 + c of x -> column c * HW + hw of dw
                ref = ref.view(N, K // 32, 32).transpose(1, 2).reshape(N, K)
            err = rel(dw, ref)
        us = timed(fn, reps)
        out[mode_name] = (us, err)
    L.hab_set_matrix_path(2047)
    fl = 2.0 * M * N * K
    a, d = out["igemm"], out["dense"]
    print(f"{kind:6s} M={M:5d} N={N:5d} K={K:5d}{' perm' if perm else '     '}  igemm {a[0]:8.1f} us ({fl / a[0] / 1e6:6.1f} TFLOP/s-eq, frac {fl / a[0] / 1e6 / PEAK_EQ:.3f}, "
          f"err {a[1]:.2e})   dense {d[0]:8.1f} us ({fl / d[0] / 1e6:6.1f} TFLOP/s-eq, frac {fl / d[0] / 1e6 / PEAK_EQ:.3f}, err {d[1]:.2e})   x{a[0] / d[0]:.2f}",
          flush=True)


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    reps = 5 if quick else 20
    if "--rnn" in sys.argv:  # the recurrent layer's projections and weight gradients at C2's minibatch (2048 rows; GRU 512, input 576)
        for kk in (512, 576):
            case("wgrad", 2048, 1536, kk, reps)
        case("dgrad", 2048, 1536, 576, reps)
        case("fwd", 512, 1536, 576, reps)
        case("fwd", 2048, 1536, 576, reps)
        case("wgrad", 4096, 2048, 512, reps)
        sys.exit(0)
    case("fwd", 512, 512, 25088, reps)
    case("dgrad", 512, 512, 25088, reps)
    case("wgrad", 2048, 512, 25088, reps, perm=True)
    if not quick:
        case("fwd", 2048, 512, 25088, reps)
        case("dgrad", 2048, 512, 25088, reps)
        case("wgrad", 2048, 512, 25088, reps, perm=False)
        case("fwd", 4096, 2048, 576, reps)
        case("dgrad", 4096, 2048, 576, reps)
        case("wgrad", 4096, 2048, 576, reps)
        case("fwd", 300, 384, 1024, reps)     # ragged M / N
        case("wgrad", 1024, 260, 388, reps)   # ragged output
