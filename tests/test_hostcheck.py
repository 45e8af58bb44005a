"""CPU: the igemm gather/epilogue functors (problems.h) executed on the host vs torch (oracle arithmetic).

Pins im2col index math, packed-weight layouts, OIHW scatter, flatten permutation and masks without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "hostcheck", "libhab_hostcheck.so")


@pytest.fixture(scope="module")
def hc():
    if not os.path.exists(SO):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC",
                               os.path.join(HERE, "hostcheck", "hostcheck.hip"), "-o", SO])
    return C.CDLL(SO)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def repack(hc, w, cpad=None):
    Cout, Cin, KH, KW = w.shape
    cpad = cpad or Cin
    wf = torch.zeros(Cout, KH, KW, cpad)
    wd = torch.zeros(Cin, KH, KW, Cout) if cpad == Cin else None
    hc.hc_repack_conv(P(w), P(wf), P(wd), Cout, Cin, KH, KW, cpad)
    return wf, wd


CONVS = [  # B, H, W, C, Cout, K, stride, pad
    (2, 20, 20, 32, 64, 4, 2, 0),
    (2, 9, 9, 64, 32, 3, 1, 0),
    (1, 12, 10, 32, 32, 3, 1, 1),
    (2, 12, 12, 32, 64, 3, 2, 1),
    (2, 11, 13, 32, 64, 1, 2, 0),
    (1, 16, 16, 4, 32, 7, 2, 3),
]


@pytest.mark.parametrize("B,H,W,Cc,Cout,K,s,p", CONVS)
def test_conv_fwd_dgrad_wgrad(hc, B, H, W, Cc, Cout, K, s, p):
    torch.manual_seed(0)
    x = torch.randn(B, Cc, H, W, requires_grad=True)
    w = torch.randn(Cout, Cc, K, K, requires_grad=True) * 0.1
    w.retain_grad()
    b = torch.randn(Cout)
    y_ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p))
    wf, wd = repack(hc, w.detach())
    xh = nhwc(x.detach())
    Ho, Wo = y_ref.shape[2:]
    y = torch.zeros(B, Ho, Wo, Cout)
    assert hc.hc_conv2d_fwd(P(xh), P(wf), P(b), P(y), B, H, W, Cc, Cout, K, K, s, p, 1) == 0
    assert torch.allclose(y, nhwc(y_ref), atol=1e-4, rtol=1e-4)
    # backward wrt pre-activation of THIS conv, given upstream grad on the relu output
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dy_pre = nhwc(gy * (y_ref > 0))  # producer-side ReLU mask already applied
    dx = torch.zeros(B, H, W, Cc)
    if Cc % 4 == 0 and Cout % 4 == 0:
        assert hc.hc_conv2d_dgrad(P(dy_pre), P(wd), None, None, P(dx), B, H, W, Cc, Cout, K, K, s, p) == 0
        assert torch.allclose(dx, nhwc(x.grad), atol=1e-4, rtol=1e-4)
        # with mask + add epilogue
        m = torch.randn(B, H, W, Cc)
        add = torch.randn(B, H, W, Cc)
        dx2 = torch.zeros(B, H, W, Cc)
        assert hc.hc_conv2d_dgrad(P(dy_pre), P(wd), P(m), P(add), P(dx2), B, H, W, Cc, Cout, K, K, s, p) == 0
        assert torch.allclose(dx2, (nhwc(x.grad) + add) * (m > 0), atol=1e-4, rtol=1e-4)
        dw = torch.zeros_like(w)
        assert hc.hc_conv2d_wgrad(P(xh), P(dy_pre), P(dw), B, H, W, Cc, Cout, K, K, s, p) == 0
        assert torch.allclose(dw, w.grad, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("has_rgb,has_depth,H,W", [(1, 1, 36, 36), (1, 1, 12, 16), (1, 1, 18, 22), (0, 1, 28, 32), (1, 0, 20, 24)])
def test_obs_conv(hc, has_rgb, has_depth, H, W):
    torch.manual_seed(1)
    nrows, B = 7, 3
    rgb = torch.randint(0, 256, (nrows, H, W, 3), dtype=torch.uint8) if has_rgb else None
    depth = torch.rand(nrows, H, W, 1) if has_depth else None
    rows = torch.tensor([5, 0, 3], dtype=torch.int32)
    parts = []
    if has_rgb:
        parts.append(rgb[rows.long()].permute(0, 3, 1, 2).float() / 255.0)
    if has_depth:
        parts.append(depth[rows.long()].permute(0, 3, 1, 2))
    x = torch.cat(parts, 1)
    Cin = x.shape[1]
    w = (torch.randn(32, Cin, 8, 8) * 0.1).requires_grad_()
    b = torch.randn(32)
    y_ref = F.relu(F.conv2d(x, w, b, stride=4))
    wf, _ = repack(hc, w.detach())
    Ho, Wo = y_ref.shape[2:]
    y = torch.zeros(B, Ho, Wo, 32)
    assert hc.hc_obs_conv2d_fwd(P(rgb), P(depth), P(rows), P(wf), P(b), P(y), B, H, W, 32, 8, 8, 4, 0, 1) == 0
    assert torch.allclose(y, nhwc(y_ref), atol=1e-4, rtol=1e-4)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dy = nhwc(gy * (y_ref > 0))
    dw = torch.zeros_like(w)
    assert hc.hc_obs_conv2d_wgrad(P(rgb), P(depth), P(rows), P(dy), P(dw), B, H, W, 32, 8, 8, 4, 0) == 0
    assert torch.allclose(dw, w.grad, atol=2e-4, rtol=1e-4)


def test_div255_equals_ieee_division_for_all_bytes(hc):
    """The observation gathers scale uint8 by a reciprocal multiply + one Newton step; it must equal x / 255.0f."""
    assert hc.hc_div255_mismatches() == 0


def test_linear_family(hc):
    torch.manual_seed(2)
    M, N, K = 9, 13, 22  # unaligned -> scalar gather path
    x = torch.randn(M, K, requires_grad=True)
    w = torch.randn(N, K, requires_grad=True)
    b = torch.randn(N)
    y_ref = F.relu(F.linear(x, w, b))
    ldy = 16
    y = torch.zeros(M, ldy)
    assert hc.hc_linear_fwd(P(x), K, P(w), K, P(b), P(y), ldy, M, N, K, 1, 0) == 0
    assert torch.allclose(y[:, :N], y_ref, atol=1e-4)
    gy = torch.randn(M, N)
    y_ref.backward(gy)
    dyp = torch.zeros(M, ldy)
    dyp[:, :N] = gy * (y_ref > 0)
    dx = torch.zeros(M, 24)
    wpad = torch.zeros(N, 24)
    wpad[:, :K] = w.detach()
    assert hc.hc_linear_dgrad(P(dyp), ldy, P(wpad), 24, None, 0, 0, P(dx), 24, M, K, N, 0) == 0
    assert torch.allclose(dx[:, :K], x.grad, atol=1e-4)
    xpad = torch.zeros(M, 24)
    xpad[:, :K] = x.detach()
    dw = torch.zeros(N, K)
    assert hc.hc_linear_wgrad(P(dyp), ldy, P(xpad), 24, P(dw), K, M, N, K, 0, 0, 0) == 0
    assert torch.allclose(dw, w.grad, atol=1e-4)
    # masked dgrad: only first mask_cols columns are masked
    mask = torch.randn(M, 24)
    dxm = torch.zeros(M, 24)
    assert hc.hc_linear_dgrad(P(dyp), ldy, P(wpad), 24, P(mask), 24, 10, P(dxm), 24, M, K, N, 0) == 0
    ref = x.grad.clone()
    ref[:, :10] *= (mask[:, :10] > 0)
    assert torch.allclose(dxm[:, :K], ref, atol=1e-4)


def test_flatten_linear(hc):
    """Linear after nn.Flatten of an NCHW map, fed by NHWC activations (simple_cnn.py:91-92)."""
    torch.manual_seed(3)
    B, Cc, Hh, Ww, N = 3, 32, 4, 5, 16
    a = torch.randn(B, Cc, Hh, Ww, requires_grad=True)
    w = torch.randn(N, Cc * Hh * Ww, requires_grad=True) * 0.1
    w.retain_grad()
    y_ref = F.linear(a.flatten(1), w)
    wp = torch.zeros_like(w)
    hc.hc_repack_flatten(P(w.detach()), P(wp), N, Cc, Hh * Ww)
    ah = nhwc(a.detach()).reshape(B, -1)
    y = torch.zeros(B, N)
    K = Cc * Hh * Ww
    assert hc.hc_linear_fwd(P(ah), K, P(wp), K, None, P(y), N, B, N, K, 0, 0) == 0
    assert torch.allclose(y, y_ref, atol=1e-4)
    gy = torch.randn(B, N)
    y_ref.backward(gy)
    dw = torch.zeros(N, K)
    assert hc.hc_linear_wgrad(P(gy), N, P(ah), K, P(dw), K, B, N, K, Cc, Hh * Ww, 0) == 0
    assert torch.allclose(dw, w.grad, atol=1e-4)
    da = torch.zeros(B, K)
    assert hc.hc_linear_dgrad(P(gy), N, P(wp), K, None, 0, 0, P(da), K, B, K, N, 0) == 0
    assert torch.allclose(da.view(B, Hh, Ww, Cc), nhwc(a.grad), atol=1e-4)


DMA_CONVS = [  # B, H, W, C, Cout, K, stride, pad   (C, Cout % 32 == 0: the LDS-DMA staged kernels)
    (2, 20, 20, 32, 64, 4, 2, 0),   # SimpleCNN conv2 geometry: merged stride classes
    (3, 9, 9, 64, 32, 3, 1, 0),
    (2, 12, 10, 32, 32, 3, 1, 1),   # padding: buffer-range zero fill + tap masks
    (2, 12, 12, 32, 64, 3, 2, 1),   # 3x3 / 2: unequal stride classes (per-class only)
    (2, 11, 13, 32, 64, 1, 2, 0),   # 1x1 / 2: classes without taps
    (2, 10, 12, 32, 32, 4, 2, 1),   # kernel = 2 x stride with padding: merged, N = 128
    (2, 9, 7, 16, 32, 2, 2, 0),     # kernel = stride, odd extent: merged, N = 64
]


@pytest.mark.parametrize("B,H,W,Cc,Cout,K,s,p", DMA_CONVS)
def test_dma_staged_functors_on_host(hc, B, H, W, Cc, Cout, K, s, p):
    """The LDS-DMA staged kernels' address functors (window, row offset, tap mask, scalar tap offsets, buffer range check) and
    the vector epilogue, emulated on the host: forward, per-class data gradient and the merged-stride-class data gradient."""
    torch.manual_seed(1)
    x = torch.randn(B, Cc, H, W, requires_grad=True)
    w = (torch.randn(Cout, Cc, K, K) * 0.1).requires_grad_()
    b = torch.randn(Cout)
    y_ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p))
    wf, wd = repack(hc, w.detach())
    xh = nhwc(x.detach())
    Ho, Wo = y_ref.shape[2:]
    if Cc % 32 == 0:
        y = torch.full((B, Ho, Wo, Cout), 7.0)
        assert hc.hc_conv2d_fwd_dma(P(xh), P(wf), P(b), P(y), B, H, W, Cc, Cout, K, K, s, p, 1) == 0
        assert torch.allclose(y, nhwc(y_ref), atol=1e-4, rtol=1e-4)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dy_pre = nhwc(gy * (y_ref > 0))
    m, add = torch.randn(B, H, W, Cc), torch.randn(B, H, W, Cc)
    ref = (nhwc(x.grad) + add) * (m > 0)
    for merged in (0, 1, 2):  # 2: the merged problem through its register-staged functors (igemm_bf3.h path)
        dx = torch.full((B, H, W, Cc), 7.0)  # every element must be written
        rc = hc.hc_conv2d_dgrad_dma(P(dy_pre), P(wd), P(m), P(add), P(dx), B, H, W, Cc, Cout, K, K, s, p, merged)
        if merged and (K % s != 0 or s == 1):
            assert rc == -2  # merged form needs kernel % stride == 0
            continue
        assert rc == 0, rc
        assert torch.allclose(dx, ref, atol=1e-4, rtol=1e-4), merged
