"""Fused convolution + GroupNorm (+ residual, + ReLU) of the small-batch ResNet passes (csrc/conv_gn_slab.h, conv1x1_gn_stream.h) against a float64
evaluation of the reference's op chain (nn.Conv2d(bias=False) -> nn.GroupNorm -> + identity -> ReLU; rl/ddppo/policy/resnet.py:51-69)
and against the unfused kernels of this library."""
import ctypes as C

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
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 96:
        del _KEEP[:48]
    return C.c_void_p(t.data_ptr())


S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

# (B, H, W, C, Cout, K, stride, pad, groups): ResNet18 / ResNet50 layers behind layer1 at 256^2 observations, the 1x1 downsample
# and bottleneck convolutions, the compression layer (one group), odd geometries, frame counts that do not fill the last workgroup
CASES = [
    (5, 32, 32, 32, 64, 1, 2, 0, 16),     # layer2.0 downsample (K = 32: most waves idle)
    (9, 16, 16, 64, 64, 3, 1, 1, 16),     # layer2: 256 pixels per frame, groups of 4
    (64, 16, 16, 64, 128, 3, 2, 1, 16),   # layer3.0 conv0
    (33, 8, 8, 128, 128, 3, 1, 1, 16),    # layer3
    (64, 8, 8, 128, 256, 3, 2, 1, 16),    # layer4.0 conv0: 16 pixels per frame, 2 frames per workgroup
    (7, 4, 4, 256, 256, 3, 1, 1, 16),     # layer4, odd frame count
    (64, 4, 4, 256, 128, 3, 1, 1, 1),     # compression: one group over all 128 channels
    (6, 4, 4, 256, 1024, 1, 1, 0, 16),    # ResNet50 layer4 expansion: groups of 64
    (6, 8, 8, 128, 512, 1, 1, 0, 16),     # ResNet50 layer3 expansion: groups of 32
    (3, 15, 8, 64, 64, 3, 1, 1, 16),      # 120 pixels per frame (128-row tile, 8 idle rows)
    (10, 5, 3, 48, 96, 3, 1, 1, 12),      # 15 pixels, 2 frames per tile, C not a power of two, groups of 8
    (4, 2, 2, 256, 256, 3, 1, 1, 16),     # 4 pixels per frame: 8 frames per tile
    # 1x1 convolutions whose frames do not fit LDS -> csrc/conv1x1_gn_stream.h (activations streamed, whole frame in the accumulators)
    (5, 32, 32, 32, 32, 1, 1, 0, 16),     # ResNet50 layer1.0 conv1: 1024 pixels, groups of 2
    (3, 32, 32, 32, 128, 1, 1, 0, 16),    # layer1 expansion / downsample: groups of 8, 4 channel slabs per frame
    (3, 16, 32, 128, 32, 1, 1, 0, 16),    # K = 128, 512 pixels (the 1024-pixel form of layer1.1 conv1 stays on the unfused pair)
    (2, 32, 16, 128, 64, 1, 1, 0, 16),    # groups of 4
    (3, 32, 32, 128, 256, 1, 2, 0, 16),   # layer2.0 downsample: stride 2, 256 output pixels, groups of 16
    (2, 30, 30, 64, 64, 1, 1, 0, 2),      # 900 pixels (idle lanes in the last tile), groups of 32
    (2, 10, 25, 256, 96, 1, 1, 0, 24),    # 250 pixels, two channel chunks
    (2, 20, 25, 128, 96, 1, 1, 0, 24),    # 500 pixels: two tiles per wave
    (3, 16, 16, 256, 64, 1, 1, 0, 16),    # ResNet50 layer2.1 conv1: 256 pixels x 256 channels (one tile per wave, all channels in one batch)
    (2, 15, 15, 256, 512, 1, 2, 0, 16),   # layer3.0 downsample form on odd geometry: stride 2, 64 output pixels of a 225-pixel frame
]


def reference(x, w, gamma, beta, res, groups, s, p, relu):
    raw = F.conv2d(x.double(), w.double(), None, stride=s, padding=p)
    y = F.group_norm(raw, groups, gamma.double(), beta.double(), 1e-5)
    B = x.shape[0]
    rg = raw.reshape(B, groups, -1)
    mean, var = rg.mean(-1), rg.var(-1, unbiased=False)
    if res is not None:
        y = y + res.double()
    if relu:
        y = y.clamp_min(0)
    return raw.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1), mean, (var + 1e-5).rsqrt()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("with_res,relu,save", [(True, 1, True), (False, 0, False)])
def test_conv_gn_fused_vs_float64(L, case, with_res, relu, save):
    B, H, W, Cc, Cout, K, s, p, groups = case
    torch.manual_seed(B * 131 + Cout)
    x = torch.randn(B, Cc, H, W) * torch.rand(B, Cc, H, W).pow(2) * 4
    w = torch.randn(Cout, Cc, K, K) / np.sqrt(Cc * K * K)
    gamma, beta = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.3
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    res = torch.randn(B, Cout, Ho, Wo) if with_res else None
    raw64, y64, mean64, rstd64 = reference(x, w, gamma, beta, res, groups, s, p, relu)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wf = w.permute(0, 2, 3, 1).contiguous().cuda()
    n = wf.numel()
    planes = torch.zeros(3 * n, dtype=torch.int16, device="cuda")
    Kr = K * K * Cc
    _lib.check(L.hab_split_weight_planes(P(wf), Cout, Kr, P(planes), S()))
    # the planes are the exact split (p0 + p1 + p2 == w), stored in MFMA fragment order
    pl = planes.view(3, n).cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    back = pl.view(np.float32).astype(np.float64).sum(0).astype(np.float32)
    co, k = np.meshgrid(np.arange(Cout), np.arange(Kr), indexing="ij")
    pos = ((co // 32) * (Kr // 16) + k // 16) * 512 + ((k % 16) // 8 * 32 + co % 32) * 8 + k % 8
    assert np.array_equal(back[pos], wf.cpu().numpy().reshape(Cout, Kr))
    resd = res.permute(0, 2, 3, 1).contiguous().cuda() if with_res else None
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda")
    raw = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda") if save else None
    mean = torch.full((B, groups), float("nan"), device="cuda") if save else None
    rstd = torch.full((B, groups), float("nan"), device="cuda") if save else None
    g, b = gamma.cuda(), beta.cuda()
    _lib.check(L.hab_conv_gn_fwd(P(xd), P(planes), P(g), P(b), P(resd), P(y), P(raw), P(mean), P(rstd), B, H, W, Cc, Cout, K, K, s, p,
                                 groups, relu, 1e-5, S()))
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    scale = y64.abs().max().item()
    assert (y.double().cpu() - y64).abs().max().item() <= 2e-5 * scale
    if save:
        assert (raw.double().cpu() - raw64).abs().max().item() <= 3e-6 * raw64.abs().max().item()
        assert (mean.double().cpu() - mean64).abs().max().item() <= 1e-5 * max(1.0, mean64.abs().max().item())
        assert ((rstd.double().cpu() - rstd64).abs() / rstd64).max().item() <= 1e-5
    # the unfused kernels of this library on the same inputs
    ws = torch.zeros(1 << 22, device="cuda")
    raw2 = torch.zeros(B, Ho, Wo, Cout, device="cuda")
    y2 = torch.zeros(B, Ho, Wo, Cout, device="cuda")
    m2, r2 = torch.zeros(B, groups, device="cuda"), torch.zeros(B, groups, device="cuda")
    _lib.check(L.hab_conv2d_fwd(P(xd), P(wf), None, P(raw2), B, H, W, Cc, Cout, K, K, s, p, 0, P(ws), ws.numel(), S()))
    rc = L.hab_groupnorm_fwd(P(raw2), P(y2), P(g), P(b), P(resd), P(m2), P(r2), B, Ho * Wo, Cout, groups, relu, 1e-5, P(ws), ws.numel(), S())
    if rc != -2:  # (the stand-alone GroupNorm kernels need Cout / 4 to divide or be a multiple of 256; the fused kernel does not)
        _lib.check(rc)
        assert (y - y2).abs().max().item() <= 2e-5 * scale
    # deterministic: a second launch gives the same bits
    y3 = torch.zeros_like(y)
    _lib.check(L.hab_conv_gn_fwd(P(xd), P(planes), P(g), P(b), P(resd), P(y3), None, None, None, B, H, W, Cc, Cout, K, K, s, p, groups,
                                 relu, 1e-5, S()))
    assert torch.equal(y, y3)


def test_conv_gn_fused_leaves_large_1x1_frames_to_the_unfused_pair(L):
    # layer1.1 conv1 of ResNet50 (1x1, 128 -> 32 at 32 x 32): one workgroup would stream 512 KB -- slower than the contraction spread
    # over the chip + GroupNorm (tools/bench_conv_gn.py); refused, the engine runs the unfused pair
    x = torch.zeros(2, 32, 32, 128, device="cuda")
    pl = torch.zeros(3 * 32 * 128, dtype=torch.int16, device="cuda")
    g = torch.ones(32, device="cuda")
    y = torch.zeros(2, 32, 32, 32, device="cuda")
    assert L.hab_conv_gn_fwd(P(x), P(pl), P(g), P(g), None, P(y), None, None, None, 2, 32, 32, 128, 32, 1, 1, 1, 0, 16, 1, 1e-5, S()) == -2


def test_conv_gn_fused_refuses_uncovered_geometries(L):
    x = torch.zeros(2, 32, 32, 32, device="cuda")
    pl = torch.zeros(3 * 64 * 288, dtype=torch.int16, device="cuda")
    g = torch.ones(32, device="cuda")
    y = torch.zeros(2, 32, 32, 32, device="cuda")
    # 1024 pixels per frame (ResNet layer1): not covered -> the caller runs the unfused pair
    assert L.hab_conv_gn_fwd(P(x), P(pl), P(g), P(g), None, P(y), None, None, None, 2, 32, 32, 32, 32, 3, 3, 1, 1, 16, 1, 1e-5, S()) == -2
    # mean without rstd
    assert L.hab_conv_gn_fwd(P(x), P(pl), P(g), P(g), None, P(y), None, P(y), None, 2, 8, 8, 32, 32, 3, 3, 1, 1, 8, 1, 1e-5, S()) == -1


def test_conv_gn_fused_refuses_inputs_larger_than_lds(L):
    # layer2.0 conv0 of ResNet18: 256 output pixels per frame from a 32 x 32 x 32 input -- the covered shape, but the input image of one
    # frame (1024 pixels x 40 x 6 bytes) does not fit LDS: refused, the engine runs the unfused pair for this layer
    x = torch.zeros(2, 32, 32, 32, device="cuda")
    pl = torch.zeros(3 * 64 * 288, dtype=torch.int16, device="cuda")
    g = torch.ones(64, device="cuda")
    y = torch.zeros(2, 16, 16, 64, device="cuda")
    assert L.hab_conv_gn_fwd(P(x), P(pl), P(g), P(g), None, P(y), None, None, None, 2, 32, 32, 32, 64, 3, 3, 2, 1, 16, 1, 1e-5, S()) == -2


def _rmv_affine(xd, creal=4):
    """16 floats of the staging-time normalisation (include/habitat_amd.h: hab_stem_conv_fwd) for made-up statistics, and the tensor
    normalised by hab_running_mean_var_normalize with the same statistics."""
    L = _lib.lib()
    mean = torch.tensor([0.4, -0.2, 0.1, 0.7], device="cuda")[:creal].contiguous()
    var = torch.tensor([0.5, 2.0, 0.003, 1.3], device="cuda")[:creal].contiguous()  # (0.003: below the 1e-2 floor)
    # the two coefficients as the DEVICE computes them (rsqrtf): normalising 1 with mean 0 gives the scale, normalising 0 the offset
    one, zero = torch.zeros(1, 4, device="cuda"), torch.zeros(1, 4, device="cuda")
    one[:, :creal] = 1.0
    _lib.check(L.hab_running_mean_var_normalize(P(one), 1, 4, creal, P(torch.zeros_like(mean)), P(var), S()))
    _lib.check(L.hab_running_mean_var_normalize(P(zero), 1, 4, creal, P(mean), P(var), S()))
    norm = torch.zeros(16, device="cuda")
    norm[:4] = one[0]
    norm[8:12] = zero[0]
    xn = xd.clone()
    _lib.check(L.hab_running_mean_var_normalize(P(xn), xn.numel() // 4, 4, creal, P(mean), P(var), S()))
    torch.cuda.synchronize()
    return norm, xn


@pytest.mark.parametrize("B,H,W", [(3, 128, 128), (2, 64, 64), (5, 31, 42), (2, 21, 128), (1, 9, 7)])
def test_stem_conv_strip_vs_float64(L, B, H, W):
    """csrc/stem_conv_strip.h (7x7 / 2 / 3, 4 -> 32, input strip in LDS) against float64 and against the im2col contraction."""
    torch.manual_seed(H * 7 + W)
    x = torch.randn(B, 4, H, W) * torch.rand(B, 4, H, W).pow(2) * 3
    w = torch.randn(32, 4, 7, 7) / 14.0
    ref = F.conv2d(x.double(), w.double(), None, stride=2, padding=3).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1], ref.shape[2]
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wf = w.permute(0, 2, 3, 1).contiguous().cuda()
    planes = torch.zeros(3 * 14 * 512, dtype=torch.int16, device="cuda")
    _lib.check(L.hab_stem_split_weights(P(wf), P(planes), S()))
    y = torch.full((B, Ho, Wo, 32), float("nan"), device="cuda")
    _lib.check(L.hab_stem_conv_fwd(P(xd), P(planes), P(y), B, H, W, None, None, 0, S()))
    torch.cuda.synchronize()
    e1 = ((y.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    ws = torch.zeros(1 << 22, device="cuda")
    y2 = torch.zeros_like(y)
    _lib.check(L.hab_conv2d_fwd(P(xd), P(wf), None, P(y2), B, H, W, 4, 32, 7, 7, 2, 3, 0, P(ws), ws.numel(), S()))
    e0 = ((y2.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert e1 <= 2 * e0 + 2e-7 and e1 < 3e-6, (e0, e1)
    y3 = torch.zeros_like(y)
    _lib.check(L.hab_stem_conv_fwd(P(xd), P(planes), P(y3), B, H, W, None, None, 0, S()))
    assert torch.equal(y, y3)
    # GroupNorm partial statistics of the strips of 8 output rows, written with the output
    for groups in (8, 16, 32):
        nstrips = (Ho + 7) // 8
        part = torch.full((B, nstrips, groups, 2), float("nan"), device="cuda")
        _lib.check(L.hab_stem_conv_fwd(P(xd), P(planes), P(y3), B, H, W, None, P(part), groups, S()))
        torch.cuda.synchronize()
        assert torch.equal(y, y3)
        yd = y.double().cpu()
        for k in range(nstrips):
            blk = yd[:, 8 * k:8 * k + 8].reshape(B, -1, groups, 32 // groups)
            mean = blk.mean(dim=(1, 3))
            m2 = ((blk - mean[:, None, :, None]) ** 2).sum(dim=(1, 3))
            pk = part[:, k].double().cpu()
            assert (pk[..., 0] - mean).abs().max() <= 1e-6 * blk.abs().max(), (groups, k)
            assert ((pk[..., 1] - m2).abs() <= 1e-5 * m2 + 1e-9).all(), (groups, k)
    # RunningMeanAndVar applied while staging == the normalised tensor through the plain kernel, bit for bit
    norm, xn = _rmv_affine(xd)
    _lib.check(L.hab_stem_conv_fwd(P(xn), P(planes), P(y), B, H, W, None, None, 0, S()))
    _lib.check(L.hab_stem_conv_fwd(P(xd), P(planes), P(y3), B, H, W, P(norm), None, 0, S()))
    assert torch.equal(y, y3)


def test_stem_conv_strip_refuses_wide_inputs(L):
    x = torch.zeros(1, 8, 300, 4, device="cuda")
    planes = torch.zeros(3 * 14 * 512, dtype=torch.int16, device="cuda")
    y = torch.zeros(1, 4, 150, 32, device="cuda")
    assert L.hab_stem_conv_fwd(P(x), P(planes), P(y), 1, 8, 300, None, None, 0, S()) == -2


@pytest.mark.parametrize("B,H,W,creal", [(3, 128, 128, 4), (300, 32, 32, 4), (2, 63, 64, 1), (5, 30, 96, 3)])
def test_stem_wgrad_strip_vs_float64(L, B, H, W, creal):
    """csrc/stem_wgrad_strip.h (weight gradient of the 7x7 / 2 / 3 stem, operands in LDS, transpose reads) against float64 and against
    the implicit-GEMM weight gradient."""
    torch.manual_seed(H + W + B)
    x = torch.randn(B, 4, H, W) * torch.rand(B, 4, H, W).pow(2) * 3
    x[:, creal:] = 0  # padding channels of the encoder input are zero
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, 32, Ho, Wo) * torch.rand(B, 32, Ho, Wo).pow(2)
    xr = x[:, :creal].double().requires_grad_(False)
    w = torch.zeros(32, creal, 7, 7, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w, None, stride=2, padding=3).backward(dy.double())
    ref = w.grad
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    ws = torch.zeros(1 << 22, device="cuda")
    dw = torch.full((32, creal, 7, 7), float("nan"), device="cuda")
    _lib.check(L.hab_stem_conv_wgrad(P(xd), P(dyd), P(dw), B, H, W, creal, P(ws), ws.numel(), None, S()))
    torch.cuda.synchronize()
    e1 = ((dw.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    dw2 = torch.zeros(32, 4, 7, 7, device="cuda")
    _lib.check(L.hab_conv2d_wgrad(P(xd), P(dyd), P(dw2), None, B, H, W, 4, 32, 7, 7, 2, 3, P(ws), ws.numel(), S()))
    e0 = ((dw2[:, :creal].double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert e1 <= 2 * e0 + 2e-7 and e1 < 3e-6, (e0, e1)
    dw3 = torch.zeros_like(dw)
    _lib.check(L.hab_stem_conv_wgrad(P(xd), P(dyd), P(dw3), B, H, W, creal, P(ws), ws.numel(), None, S()))
    assert torch.equal(dw, dw3)
    norm, xn = _rmv_affine(xd, creal)
    _lib.check(L.hab_stem_conv_wgrad(P(xn), P(dyd), P(dw), B, H, W, creal, P(ws), ws.numel(), None, S()))
    _lib.check(L.hab_stem_conv_wgrad(P(xd), P(dyd), P(dw3), B, H, W, creal, P(ws), ws.numel(), P(norm), S()))
    assert torch.equal(dw, dw3)


def test_stem_wgrad_strip_refuses_uncovered_widths(L):
    x = torch.zeros(1, 42, 42, 4, device="cuda")
    dy = torch.zeros(1, 21, 21, 32, device="cuda")
    dw = torch.zeros(32, 4, 7, 7, device="cuda")
    ws = torch.zeros(1 << 21, device="cuda")
    assert L.hab_stem_conv_wgrad(P(x), P(dy), P(dw), 1, 42, 42, 4, P(ws), ws.numel(), None, S()) == -2  # Wo = 21: not a multiple of 16
