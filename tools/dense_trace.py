"""(needs a library built with the development hooks: make -C habitat-lab_amd/csrc EXTRA=-DHAB_DENSE_DEV)
Development: phase timeline of workgroup 0 of the dense GEMM kernel (HAB_DENSE_ABLATE=16), shader-clock cycles per phase per wave."""
import ctypes as C, os, sys
os.environ["HAB_DENSE_ABLATE"] = os.environ.get("HAB_DENSE_ABLATE", "16")
os.environ.setdefault("HAB_DENSE_MIN_MFLOP", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd import _lib
L = C.CDLL(os.path.join(ROOT, "habitat-lab_amd", "habitat_amd", "libhabitat_amd.so"))
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
kind = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
M, N, K = 2048, 512, 25088
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; dy = torch.randn(M, N, device="cuda")
ws = torch.zeros(1 << 26, device="cuda")
for rep in range(2):
    if kind == "wgrad":
        dw = torch.zeros(N, K, device="cuda")
        rc = L.hab_linear_wgrad(P(dy), N, P(x), K, P(dw), K, M, N, K, 32, K // 32, 0, P(ws), C.c_size_t(ws.numel()), S())
    elif kind == "fwd":
        y = torch.zeros(512, N, device="cuda")
        rc = L.hab_linear_fwd(P(x), K, P(w), K, None, P(y), N, 512, N, K, 1, 0, P(ws), C.c_size_t(ws.numel()), S())
    else:
        dx = torch.zeros(512, K, device="cuda")
        rc = L.hab_linear_dgrad(P(dy), N, P(w), K, None, 0, P(dx), K, 512, K, N, 0, P(ws), C.c_size_t(ws.numel()), S())
    assert rc == 0, rc
    torch.cuda.synchronize()
buf = (C.c_longlong * 320)()
assert L.hab_debug_dense_trace(buf) == 0
t = torch.tensor(list(buf)).view(8, 8, 5)
base = int(t[:, 0, 0].min())
print(f"{kind}: per wave, k-tiles 8..15: [start | +stage/fetch(early) | +mfma | +stage/fetch(late) | +barrier]  (cycles since first stamp)")
for w_ in range(8):
    rows = []
    for k in range(8):
        s0 = int(t[w_, k, 0]) - base
        d = [int(t[w_, k, i + 1] - t[w_, k, i]) for i in range(4)]
        rows.append(f"{s0:6d}:{d[0]:5d}/{d[1]:5d}/{d[2]:5d}/{d[3]:5d}")
    print(f"wave {w_}: " + "  ".join(rows))
it = (t[:, 7, 4] - t[:, 0, 0]).float().mean() / 8
print(f"mean cycles per k-tile iteration: {float(it):.0f}")
