"""Split-bf16 matrix path (csrc/igemm_bf3.h, obs_conv_bf3.h) against a float64 reference: the path must be as accurate as the fp32
MFMA path (its arithmetic is fp32-equivalent: exact 3-term operand split, six partial products, dropped terms <= 2^-24)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from habitat_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def L():
    return _lib.lib()


_KEEP = []


def P(t):
    """Device pointer of a tensor for a ctypes call.  The tensor is kept alive for the next few dozen calls: a temporary that only lived
    inside P(...) would be freed -- and its block handed to the next allocation -- before the launch that reads it."""
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 96:
        del _KEEP[:48]
    return C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def both_paths(L, fn, mask):
    out = {}
    prev = L.hab_set_matrix_path(-1)
    extra = int(os.environ.get("HAB_TEST_EXTRA_PATH_BITS", "0"))  # development: further matrix-path bits on top of the mode under test
    try:
        for mode in (0, mask):
            L.hab_set_matrix_path(mode | extra if mode else 0)
            out[mode] = fn()
    finally:
        L.hab_set_matrix_path(prev)
    return out[0], out[mask]


def err_vs(ref64, y):
    ref = ref64.double()
    return ((y.double().cpu() - ref).abs().max() / ref.abs().max()).item()


def repack_fwd(L, w):  # OIHW -> [Cout][KH][KW][Cin]
    return w.permute(0, 2, 3, 1).contiguous().cuda()


@pytest.mark.parametrize("B,H,W,Cc,Cout,K,s,p", [(8, 63, 63, 32, 64, 4, 2, 0), (16, 16, 16, 64, 64, 3, 1, 1), (64, 8, 8, 128, 128, 3, 1, 1),
                                                  (3, 30, 30, 64, 32, 3, 1, 0)])
def test_conv_fwd_bf3_as_accurate_as_fp32_path(L, B, H, W, Cc, Cout, K, s, p):
    torch.manual_seed(0)
    x = torch.randn(B, Cc, H, W) * torch.rand(B, Cc, H, W).pow(4) * 50  # wide dynamic range
    w = torch.randn(Cout, Cc, K, K) / np.sqrt(Cc * K * K)
    b = torch.randn(Cout)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    xd, wf, bd = x.permute(0, 2, 3, 1).contiguous().cuda(), repack_fwd(L, w), b.cuda()
    ws = torch.zeros(1 << 22, device="cuda")

    def run():
        y = torch.zeros(ref.shape, device="cuda")
        _lib.check(L.hab_conv2d_fwd(P(xd), P(wf), P(bd), P(y), B, H, W, Cc, Cout, K, K, s, p, 0, P(ws), ws.numel(), S()))
        return y

    y0, y1 = both_paths(L, run, 1)
    e0, e1 = err_vs(ref, y0), err_vs(ref, y1)
    assert e1 <= 2 * e0 + 2e-7, (e0, e1)
    assert e1 < 3e-6


def test_obs_conv_bf3_as_accurate_as_fp32_path(L):
    torch.manual_seed(1)
    B, H, W = 5, 256, 256
    rgb = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8)
    depth = torch.rand(B, H, W, 1)
    x = torch.cat([rgb.double() / 255.0, depth.double()], -1).permute(0, 3, 1, 2)
    w = torch.randn(32, 4, 8, 8) / 16
    b = torch.randn(32)
    ref = F.relu(F.conv2d(x, w.double(), b.double(), stride=4)).permute(0, 2, 3, 1)
    wf, bd = repack_fwd(L, w), b.cuda()
    rg, dp = rgb.cuda(), depth.cuda()
    ws = torch.zeros(1 << 22, device="cuda")

    def run():
        y = torch.zeros(ref.shape, device="cuda")
        _lib.check(L.hab_obs_conv2d_fwd(P(rg), P(dp), None, P(wf), P(bd), P(y), B, H, W, 32, 8, 8, 4, 0, 1, P(ws), ws.numel(), S()))
        return y

    y0, y1 = both_paths(L, run, 2)
    e0, e1 = err_vs(ref, y0), err_vs(ref, y1)
    assert e1 <= 2 * e0 + 2e-7, (e0, e1)
    assert e1 < 3e-6


def test_linear_fwd_bf3_as_accurate_as_fp32_path(L):
    torch.manual_seed(2)
    M, N, K = 300, 512, 25088
    x = torch.randn(M, K) * torch.rand(M, K).pow(3) * 10
    w = torch.randn(N, K) * 0.01
    ref = x.double() @ w.double().t()
    xd, wd = x.cuda(), w.cuda()
    ws = torch.zeros(1 << 24, device="cuda")

    def run():
        y = torch.zeros(M, N, device="cuda")
        _lib.check(L.hab_linear_fwd(P(xd), K, P(wd), K, None, P(y), N, M, N, K, 0, 0, P(ws), ws.numel(), S()))
        return y

    y0, y1 = both_paths(L, run, 1)
    e0, e1 = err_vs(ref, y0), err_vs(ref, y1)
    assert e1 <= 2 * e0 + 2e-7, (e0, e1)


def test_bf3_has_no_systematic_bias_on_same_sign_products(L):
    """All-positive operands: a truncating split would lose a same-signed 2^-24 fraction of EVERY product (error grows ~K); the
    round-to-nearest split keeps the error of the fp32 path."""
    torch.manual_seed(3)
    M, N, K = 512, 256, 8192
    x = torch.rand(M, K) + 0.5
    w = torch.rand(N, K) + 0.5
    ref = x.double() @ w.double().t()
    xd, wd = x.cuda(), w.cuda()
    ws = torch.zeros(1 << 24, device="cuda")

    def run():
        y = torch.zeros(M, N, device="cuda")
        _lib.check(L.hab_linear_fwd(P(xd), K, P(wd), K, None, P(y), N, M, N, K, 0, 0, P(ws), ws.numel(), S()))
        return y

    y0, y1 = both_paths(L, run, 1)
    e0, e1 = err_vs(ref, y0), err_vs(ref, y1)
    mean1 = ((y1.double().cpu() - ref) / ref).mean().abs().item()
    assert e1 <= 1.5 * e0 + 3e-8, (e0, e1)
    assert mean1 < 3e-8, mean1


def test_obs_wgrad_bf3_as_accurate_as_fp32_path(L):
    torch.manual_seed(4)
    B, H, W = 6, 256, 256
    rgb = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8)
    depth = torch.rand(B, H, W, 1)
    x = torch.cat([rgb.double() / 255.0, depth.double()], -1).permute(0, 3, 1, 2)
    dy = torch.randn(B, 63, 63, 32) * torch.rand(B, 63, 63, 32).pow(3)
    w = torch.zeros(32, 4, 8, 8, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, stride=4).backward(dy.double().permute(0, 3, 1, 2))
    ref, ref_b = w.grad, dy.double().sum((0, 1, 2))
    rg, dp, dyd = rgb.cuda(), depth.cuda(), dy.cuda()
    ws = torch.zeros(1 << 24, device="cuda")

    def run():
        dw = torch.zeros(32, 4, 8, 8, device="cuda")
        db = torch.zeros(32, device="cuda")
        _lib.check(L.hab_obs_conv2d_wgrad(P(rg), P(dp), None, P(dyd), P(dw), P(db), B, H, W, 32, 8, 8, 4, 0, P(ws), ws.numel(), S()))
        return torch.cat([dw.flatten(), db])

    y0, y1 = both_paths(L, run, 2)
    full = torch.cat([ref.flatten(), ref_b])
    e0, e1 = err_vs(full, y0), err_vs(full, y1)
    assert e1 <= 2 * e0 + 2e-7, (e0, e1)
    assert e1 < 3e-6


@pytest.mark.parametrize("B,H,W,Cc,Cout,K,s,p", [(8, 63, 63, 32, 64, 4, 2, 0), (16, 16, 16, 64, 64, 3, 1, 1), (64, 8, 8, 128, 128, 3, 1, 1),
                                                  (3, 30, 30, 64, 32, 3, 1, 0)])
def test_conv_wgrad_bf3_as_accurate_as_fp32_path(L, B, H, W, Cc, Cout, K, s, p):
    torch.manual_seed(5)
    x = torch.randn(B, Cc, H, W) * torch.rand(B, Cc, H, W).pow(4) * 50
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo)
    w = torch.zeros(Cout, Cc, K, K, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, stride=s, padding=p).backward(dy.double())
    ref = torch.cat([w.grad.flatten(), dy.double().sum((0, 2, 3))])
    xd, dyd = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    ws = torch.zeros(1 << 24, device="cuda")

    def run():
        dw = torch.zeros(Cout, Cc, K, K, device="cuda")
        db = torch.zeros(Cout, device="cuda")
        _lib.check(L.hab_conv2d_wgrad(P(xd), P(dyd), P(dw), P(db), B, H, W, Cc, Cout, K, K, s, p, P(ws), ws.numel(), S()))
        return torch.cat([dw.flatten(), db])

    y0, y1 = both_paths(L, run, 4 | 8)
    e0, e1 = err_vs(ref, y0), err_vs(ref, y1)
    assert e1 <= 2 * e0 + 2e-7, (e0, e1)
    assert e1 < 3e-6


def test_linear_dgrad_wgrad_bf3_as_accurate_as_fp32_path(L):
    torch.manual_seed(6)
    M, N, K = 700, 512, 3136  # rows, out features, in features
    x = torch.randn(M, K) * torch.rand(M, K).pow(3) * 10
    w = torch.randn(N, K) * 0.02
    dy = torch.randn(M, N)
    ref_dx = dy.double() @ w.double()
    ref_dw = dy.double().t() @ x.double()
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    ws = torch.zeros(1 << 24, device="cuda")

    def run():
        dx = torch.zeros(M, K, device="cuda")
        dw = torch.zeros(N, K, device="cuda")
        _lib.check(L.hab_linear_dgrad(P(dyd), N, P(wd), K, None, 0, P(dx), K, M, K, N, 0, P(ws), ws.numel(), S()))
        _lib.check(L.hab_linear_wgrad(P(dyd), N, P(xd), K, P(dw), K, M, N, K, 0, 0, 0, P(ws), ws.numel(), S()))
        return dx, dw

    (dx0, dw0), (dx1, dw1) = both_paths(L, run, 4)
    for ref, y0, y1 in ((ref_dx, dx0, dx1), (ref_dw, dw0, dw1)):
        e0, e1 = err_vs(ref, y0), err_vs(ref, y1)
        assert e1 <= 2 * e0 + 2e-7, (e0, e1)


@pytest.mark.parametrize("B,H,W,Cc,Cout,p", [(5, 32, 32, 32, 32, 1), (3, 30, 30, 64, 32, 0), (7, 16, 16, 32, 32, 1), (2, 21, 19, 32, 32, 1),
                                            (130, 32, 32, 32, 32, 1)])
def test_patch_resident_conv_fwd_dgrad_as_accurate_as_fp32_path(L, B, H, W, Cc, Cout, p):
    """conv_patch_bf3.h (stride-1 3x3, N = 32): forward with bias + ReLU and data gradient with residual add + ReLU mask against
    float64, incl. tiles that run over the image's last rows (H % 4 != 0), idle columns (W < 32) and > 131072 rows (pre-split weights)."""
    torch.manual_seed(8)
    x = torch.randn(B, Cc, H, W) * torch.rand(B, Cc, H, W).pow(3) * 20
    w = torch.randn(Cout, Cc, 3, 3) / np.sqrt(Cc * 9)
    b = torch.randn(Cout)
    xd64 = x.double().requires_grad_()
    pre = F.conv2d(xd64, w.double(), b.double(), stride=1, padding=p)
    ref_y = F.relu(pre).permute(0, 2, 3, 1)
    Ho, Wo = pre.shape[2:]
    dy = torch.randn(B, Cout, Ho, Wo)
    pre.backward(dy.double())
    mask, add = torch.randn(B, H, W, Cc), torch.randn(B, H, W, Cc)
    ref_dx = (xd64.grad.permute(0, 2, 3, 1) + add.double()) * (mask > 0)
    xn, dyn = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    wf, wdg, bd = w.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(1, 2, 3, 0).contiguous().cuda(), b.cuda()
    md, ad = mask.cuda(), add.cuda()
    ws = torch.zeros(1 << 22, device="cuda")

    def run():
        y = torch.zeros(B, Ho, Wo, Cout, device="cuda")
        dx = torch.full((B, H, W, Cc), 7.0, device="cuda")
        _lib.check(L.hab_conv2d_fwd(P(xn), P(wf), P(bd), P(y), B, H, W, Cc, Cout, 3, 3, 1, p, 1, P(ws), ws.numel(), S()))
        _lib.check(L.hab_conv2d_dgrad(P(dyn), P(wdg), P(md), P(ad), P(dx), B, H, W, Cc, Cout, 3, 3, 1, p, P(ws), ws.numel(), S()))
        return y, dx

    (y0, dx0), (y1, dx1) = both_paths(L, run, 16)
    for ref, a0, a1 in ((ref_y, y0, y1), (ref_dx, dx0, dx1)):
        e0, e1 = err_vs(ref, a0), err_vs(ref, a1)
        assert e1 <= 2 * e0 + 2e-7, (e0, e1)
        assert e1 < 3e-6


def _with_path(L, mode, fn):
    prev = L.hab_set_matrix_path(-1)
    try:
        L.hab_set_matrix_path(mode)
        return fn()
    finally:
        L.hab_set_matrix_path(prev)


def test_producer_consumer_kernels_match_the_plain_split_kernels(L):
    """Matrix-path bit 5 (igemm_bf3_ws.h / obs_conv_bf3_ws.h, on by default where it measured faster): same split, same operand order,
    same sign schedule and split-K plan as igemm_bf3_kernel -- the long-K 128 x 128 forward-form tiles must come out BIT-identical with
    and without it; the observation-ingest convolution (different tile loop) must be as accurate against float64."""
    torch.manual_seed(5)
    # fc 25088 -> 512 forward (LinearFwdProb, K >= 2048): M = 300 rows, partial last tile
    M, N, K = 300, 512, 25088
    x = (torch.randn(M, K) * torch.rand(M, K).pow(3) * 10).cuda()
    w = (torch.randn(N, K) * 0.01).cuda()
    ws = torch.zeros(1 << 24, device="cuda")

    def lin():
        y = torch.zeros(M, N, device="cuda")
        _lib.check(L.hab_linear_fwd(P(x), K, P(w), K, None, P(y), N, M, N, K, 0, 0, P(ws), ws.numel(), S()))
        return y

    assert torch.equal(_with_path(L, 31, lin), _with_path(L, 63, lin))
    # 3x3 256 -> 256 at 4x4 (ResNet18 layer4), forward and data gradient: K = 2304
    B, Cc = 70, 256
    xc = torch.randn(B, 4, 4, Cc, device="cuda")
    wf = (torch.randn(Cc, 3, 3, Cc, device="cuda") / 48)
    bias = torch.randn(Cc, device="cuda")

    def conv():
        y = torch.zeros(B, 4, 4, Cc, device="cuda")
        dx = torch.zeros(B, 4, 4, Cc, device="cuda")
        _lib.check(L.hab_conv2d_fwd(P(xc), P(wf), P(bias), P(y), B, 4, 4, Cc, Cc, 3, 3, 1, 1, 1, P(ws), ws.numel(), S()))
        _lib.check(L.hab_conv2d_dgrad(P(xc), P(wf), None, None, P(dx), B, 4, 4, Cc, Cc, 3, 3, 1, 1, P(ws), ws.numel(), S()))
        return torch.cat([y.view(-1), dx.view(-1)])

    assert torch.equal(_with_path(L, 31, conv), _with_path(L, 63, conv))
    # observation-ingest convolution at the benchmark geometry, odd frame count
    Bo, H, W = 7, 256, 256
    rgb = torch.randint(0, 256, (Bo, H, W, 3), dtype=torch.uint8)
    depth = torch.rand(Bo, H, W, 1)
    xo = torch.cat([rgb.double() / 255.0, depth.double()], -1).permute(0, 3, 1, 2)
    wo = torch.randn(32, 4, 8, 8) / 16
    bo = torch.randn(32)
    ref = F.relu(F.conv2d(xo, wo.double(), bo.double(), stride=4)).permute(0, 2, 3, 1)
    wfo, bd, rg, dp = repack_fwd(L, wo), bo.cuda(), rgb.cuda(), depth.cuda()

    def obs():
        y = torch.zeros(ref.shape, device="cuda")
        _lib.check(L.hab_obs_conv2d_fwd(P(rg), P(dp), None, P(wfo), P(bd), P(y), Bo, H, W, 32, 8, 8, 4, 0, 1, P(ws), ws.numel(), S()))
        return y

    e_plain, e_ws = err_vs(ref, _with_path(L, 31, obs)), err_vs(ref, _with_path(L, 63, obs))
    assert e_ws <= 2 * e_plain + 2e-7 and e_ws < 3e-6, (e_plain, e_ws)


@pytest.mark.parametrize("B,H,W", [(5, 256, 256), (3, 96, 128), (2, 256, 64), (67, 44, 44)])
def test_obs_conv_patch_resident_as_accurate_as_fp32_path(L, B, H, W):
    """Matrix-path bit 6 (obs_conv_patch.h): the first convolution with the observation patch resident in LDS -- same six partial
    products as obs_conv_bf3.h in another grouping -- against float64, next to the fp32 MFMA path: full benchmark geometry (Ho = Wo =
    63: partial last tile, one idle lane), a narrow image (second 32-pixel half empty), one with fewer rows than a tile, and the
    golden fixtures' 44 x 44 with a frame count that is not a multiple of anything."""
    torch.manual_seed(11)
    rgb = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8)
    depth = torch.rand(B, H, W, 1) * torch.rand(B, H, W, 1).pow(3)
    x = torch.cat([rgb.double() / 255.0, depth.double()], -1).permute(0, 3, 1, 2)
    w = torch.randn(32, 4, 8, 8) / 16
    b = torch.randn(32)
    ref = F.relu(F.conv2d(x, w.double(), b.double(), stride=4)).permute(0, 2, 3, 1)
    wf, bd, rg, dp = repack_fwd(L, w), b.cuda(), rgb.cuda(), depth.cuda()
    ws = torch.zeros(1 << 22, device="cuda")

    def run():
        y = torch.full(ref.shape, -7.0, device="cuda")
        _lib.check(L.hab_obs_conv2d_fwd(P(rg), P(dp), None, P(wf), P(bd), P(y), B, H, W, 32, 8, 8, 4, 0, 1, P(ws), ws.numel(), S()))
        return y

    e_f32, e_patch = err_vs(ref, _with_path(L, 0, run)), err_vs(ref, _with_path(L, 2 | 64, run))
    assert e_patch <= 2 * e_f32 + 2e-7 and e_patch < 3e-6, (e_f32, e_patch)
    # gathered through rows[] (the minibatch indirection of the update) and bit-reproducible
    rows = torch.randperm(B).int().cuda()
    y_rows = torch.zeros(ref.shape, device="cuda")

    def run_rows():
        _lib.check(L.hab_obs_conv2d_fwd(P(rg), P(dp), P(rows), P(wf), P(bd), P(y_rows), B, H, W, 32, 8, 8, 4, 0, 1, P(ws), ws.numel(), S()))
        return y_rows.clone()

    a = _with_path(L, 2 | 64, run_rows)
    assert torch.equal(a, _with_path(L, 2 | 64, run_rows))
    assert err_vs(ref[rows.cpu().long()], a) < 3e-6


@pytest.mark.parametrize("B,H,W,Cc,Cout,p,K,st", [(3, 30, 30, 64, 32, 0, 3, 1), (37, 30, 30, 64, 32, 0, 3, 1), (5, 32, 32, 32, 32, 1, 3, 1),
                                                  (67, 32, 32, 32, 32, 1, 3, 1), (4, 16, 16, 64, 64, 1, 3, 1), (131, 16, 16, 64, 64, 1, 3, 1),
                                                  (1, 32, 32, 32, 32, 1, 3, 1), (3, 63, 63, 32, 64, 0, 4, 2), (41, 63, 63, 32, 64, 0, 4, 2)])
def test_wgrad3x3_strip_resident_as_accurate_as_fp32_path(L, B, H, W, Cc, Cout, p, K, st):
    """Matrix-path bit 7 (wgrad3x3_bf3.h): the 3x3 / stride-1 (and SimpleCNN conv2's 4x4 / stride-2: k-slot groups that cross output rows,
    strided taps) weight gradient with the strip resident in LDS and fragments built by LDS transpose reads, against float64 -- every weight (a transposed / shifted tap or a swapped channel half shows as an O(1) error), the
    bias gradient, frame counts that do not divide into the workgroups' strip ranges, one frame (most workgroups idle), padded
    borders (first / last strip of an image), and bit-for-bit reproducibility.  As accurate as the fp32 MFMA path."""
    torch.manual_seed(11)
    x = torch.randn(B, Cc, H, W) * torch.rand(B, Cc, H, W).pow(4) * 50
    Ho, Wo = (H + 2 * p - K) // st + 1, (W + 2 * p - K) // st + 1
    dy = torch.randn(B, Cout, Ho, Wo)
    w = torch.zeros(Cout, Cc, K, K, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, stride=st, padding=p).backward(dy.double())
    ref = torch.cat([w.grad.flatten(), dy.double().sum((0, 2, 3))])
    xd, dyd = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    ws = torch.zeros(1 << 24, device="cuda")

    def run():
        ws.normal_()  # stale slab contents must not matter
        dw = torch.zeros(Cout, Cc, K, K, device="cuda")
        db = torch.zeros(Cout, device="cuda")
        _lib.check(L.hab_conv2d_wgrad(P(xd), P(dyd), P(dw), P(db), B, H, W, Cc, Cout, K, K, st, p, P(ws), ws.numel(), S()))
        return torch.cat([dw.flatten(), db])

    y_fp32 = _with_path(L, 0, run)
    y_igemm = _with_path(L, 4 | 8, run)
    y_strip = _with_path(L, 4 | 8 | 128, run)
    assert not torch.equal(y_strip, y_igemm), "bit 7 did not select another kernel"
    e0, e1, e2 = err_vs(ref, y_fp32), err_vs(ref, y_igemm), err_vs(ref, y_strip)
    assert e2 <= 2 * e0 + 2e-7 and e2 < 3e-6, (e0, e1, e2)
    # per-element check in units of the weight's own magnitude scale (max-norm alone would hide one wrong small tap)
    d = (y_strip.double().cpu() - ref).abs()
    assert float(d.max()) <= 3e-6 * float(ref.abs().max())
    assert torch.equal(y_strip, _with_path(L, 4 | 8 | 128, run)), "not reproducible"


# conv2's input geometry = conv1's output: 63 x 63 at 256^2 observations (the compile-time instantiation), 55 at 224^2, 31 at 128^2, 20 at
# 84^2 (simple_cnn.py:35-93 takes any size), non-square, and the smallest input a 4x4 filter accepts
C2_GEOMS = [(63, 63), (55, 55), (31, 31), (20, 20), (20, 31), (33, 14), (4, 5)]


@pytest.mark.parametrize("H,W", C2_GEOMS, ids=lambda v: str(v))
@pytest.mark.parametrize("B,relu,with_bias", [(16, 1, True), (37, 0, True), (64, 1, False), (131, 1, True)])
def test_conv2_strip_resident_forward_as_accurate_as_fp32_path(L, B, relu, with_bias, H, W):
    """Matrix-path bit 8 (conv2_fwd_strip.h): SimpleCNN conv2's forward with the input strip in LDS and the filter slices in the waves'
    registers, partial sums of the eight waves folded in LDS: every output against float64 (a wrong tap, column class, swizzle slot,
    channel block or wave order shows as an O(1) error), bias / no bias, ReLU / none, frame counts that leave workgroups with ragged
    strip ranges, bit-for-bit reproducibility; as accurate as the fp32 MFMA path, and NOT the implicit-GEMM kernel's bits (so the new
    kernel really ran)."""
    if (H, W) != (63, 63) and B in (37, 64):
        pytest.skip("the runtime-geometry instantiation is covered at B = 16 and 131")
    torch.manual_seed(12)
    x = torch.randn(B, 32, H, W) * torch.rand(B, 32, H, W).pow(4) * 50
    w = torch.randn(64, 32, 4, 4) / np.sqrt(32 * 16)
    b = torch.randn(64) if with_bias else None
    ref = F.conv2d(x.double(), w.double(), b.double() if with_bias else None, stride=2).permute(0, 2, 3, 1)
    if relu:
        ref = F.relu(ref)
    xd, wf = x.permute(0, 2, 3, 1).contiguous().cuda(), repack_fwd(L, w)
    bd = b.cuda() if with_bias else None
    ws = torch.zeros(1 << 22, device="cuda")

    def run():
        y = torch.full(ref.shape, 7.0, device="cuda")
        _lib.check(L.hab_conv2d_fwd(P(xd), P(wf), P(bd), P(y), B, H, W, 32, 64, 4, 4, 2, 0, relu, P(ws), ws.numel(), S()))
        return y

    y_fp32 = _with_path(L, 0, run)
    y_igemm = _with_path(L, 1, run)
    y_strip = _with_path(L, 1 | 256, run)
    assert not torch.equal(y_strip, y_igemm), "bit 8 did not select another kernel"
    e0, e2 = err_vs(ref, y_fp32), err_vs(ref, y_strip)
    assert e2 <= 2 * e0 + 2e-7 and e2 < 3e-6, (e0, e2)
    d = (y_strip.double().cpu() - ref).abs()
    assert float(d.max()) <= 3e-6 * float(ref.abs().max())
    assert torch.equal(y_strip, _with_path(L, 1 | 256, run)), "not reproducible"


@pytest.mark.parametrize("H,W", C2_GEOMS, ids=lambda v: str(v))
@pytest.mark.parametrize("B,with_mask", [(16, True), (37, True), (64, False)])
def test_conv2_strip_resident_data_gradient_as_accurate_as_fp32_path(L, B, with_mask, H, W):
    """Matrix-path bit 9 (conv2_dgrad_strip.h): SimpleCNN conv2's data gradient with the dY strip in LDS and the filter slices in the
    waves' registers (four taps of a row class folded in LDS): every element of dX against float64 -- border cells that read dY outside
    the image, the half-empty last cell row / column (h = w = 62 has no odd neighbour), the ReLU mask, no mask, ragged frame counts,
    bit-for-bit reproducibility; as accurate as the fp32 MFMA path and not the implicit-GEMM kernel's bits."""
    if (H, W) != (63, 63) and B == 64:
        pytest.skip("the runtime-geometry instantiation is covered at B = 16 and 37")
    torch.manual_seed(13)
    Ho, Wo = (H - 4) // 2 + 1, (W - 4) // 2 + 1
    w = torch.randn(64, 32, 4, 4) / np.sqrt(32 * 16)
    dy = torch.randn(B, 64, Ho, Wo) * torch.rand(B, 64, Ho, Wo).pow(3) * 20
    x64 = torch.zeros(B, 32, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x64, w.double(), None, stride=2).backward(dy.double())
    ref = x64.grad.permute(0, 2, 3, 1)
    mask = torch.randn(B, H, W, 32)
    if with_mask:
        ref = ref * (mask > 0)
    dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
    wdg = w.permute(1, 2, 3, 0).contiguous().cuda()  # [ci][kh][kw][co]
    md = mask.cuda() if with_mask else None
    ws = torch.zeros(1 << 22, device="cuda")

    def run():
        dx = torch.full((B, H, W, 32), 7.0, device="cuda")
        _lib.check(L.hab_conv2d_dgrad(P(dyn), P(wdg), P(md), None, P(dx), B, H, W, 32, 64, 4, 4, 2, 0, P(ws), ws.numel(), S()))
        return dx

    d_fp32 = _with_path(L, 0, run)
    d_igemm = _with_path(L, 1, run)
    d_strip = _with_path(L, 1 | 512, run)
    assert not torch.equal(d_strip, d_igemm), "bit 9 did not select another kernel"
    e0, e2 = err_vs(ref, d_fp32), err_vs(ref, d_strip)
    assert e2 <= 2 * e0 + 2e-7 and e2 < 3e-6, (e0, e2)
    assert float((d_strip.double().cpu() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    assert torch.equal(d_strip, _with_path(L, 1 | 512, run)), "not reproducible"


def test_non_finite_operands_on_the_split_path_are_pinned(L):
    """NOTEBOOK.md 4 / INTEGRATION.md: the 3-term split of +-Inf is (+-Inf, NaN, NaN), so on the split-bf16 path an Inf or NaN operand
    yields NaN in EVERY output it contributes to and nowhere else; the fp32 MFMA path (hab_set_matrix_path(0)) propagates +-Inf through
    products with finite non-zero factors.  Pinned so that a change of the split (or of the dispatch) that alters this shows up."""
    torch.manual_seed(0)
    B, H, W, Cc, Cout, K = 2, 12, 12, 32, 32, 3
    x = torch.rand(B, H, W, Cc) + 0.5
    w = torch.rand(Cout, K, K, Cc) + 0.1  # strictly positive: an Inf input reaches every output of its receptive field with weight > 0
    hit = (1, 5, 6, 7)  # (image, row, column, channel) of the non-finite input
    ws = torch.zeros(1 << 20, device="cuda")
    wf = w.cuda()

    def run(xin, mode):
        prev = L.hab_set_matrix_path(mode)
        try:
            y = torch.zeros(B, H, W, Cout, device="cuda")
            _lib.check(L.hab_conv2d_fwd(P(xin.cuda()), P(wf), None, P(y), B, H, W, Cc, Cout, K, K, 1, 1, 0, P(ws), ws.numel(), S()))
            return y.cpu()
        finally:
            L.hab_set_matrix_path(prev)

    reach = torch.zeros(B, H, W, dtype=torch.bool)
    reach[hit[0], hit[1] - 1:hit[1] + 2, hit[2] - 1:hit[2] + 2] = True  # 3x3, padding 1: the 9 output pixels that read the input pixel
    clean = {m: run(x, m) for m in (0, 1)}
    for bad in (float("inf"), float("-inf"), float("nan")):
        xb = x.clone()
        xb[hit] = bad
        for mode in (0, 1):
            y = run(xb, mode)
            assert torch.isfinite(y[~reach]).all() and torch.equal(y[~reach], clean[mode][~reach]), (bad, mode)
            if mode == 1 or bad != bad:
                assert torch.isnan(y[reach]).all(), (bad, mode)  # split path: NaN wherever the operand is read
            else:
                assert (y[reach] == bad).all(), (bad, mode)      # fp32 MFMA path: the infinity itself
