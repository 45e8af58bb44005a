import ctypes as C, os, sys, torch
sys.path.insert(0, "/root/repo/habitat-lab_amd")
from habitat_amd import _lib
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 26, device="cuda")
N, K = 512, 25088
w = torch.randn(N, K, device="cuda") * 0.01
bias = torch.randn(N, device="cuda")
for M in (16, 32, 48, 64, 128):
    x = torch.randn(M, K, device="cuda")
    y = torch.empty(M, N, device="cuda")
    fn = lambda: _lib.check(L.hab_linear_fwd(P(x), K, P(w), K, P(bias), P(y), N, M, N, K, 1, 0, P(ws), ws.numel(), S()))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    ref = (x.double() @ w.double().t() + bias.double()).relu()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"M {M:4d}: {a.elapsed_time(b) / 50 * 1e3:7.1f} us  rel err {err:.1e}", flush=True)
