"""GPU parity of the kernel-level C-ABI entry points (include/habitat_amd.h) against the CPU oracle.
Integer / index results are compared bit-exactly; fp32 results with the tolerance stated per test."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import functional as O
from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from habitat_amd import _lib
    return _lib.lib()


_KEEP = []  # tensors whose pointers were handed to the library stay alive until the test ends


@pytest.fixture(autouse=True)
def _release_kept_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def P(t):
    if t is None:
        return None
    _KEEP.append(t)  # a temporary like `x.cuda()` would otherwise be freed (and its block reused) before the launch
    return C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ck(code):
    assert code == 0, f"habitat_amd error {code}"


def test_loaded_native_library(L):
    assert L.hab_abi_version() == 1


# ------------------------------------------------------------------------------------------------
def test_synth_bit_exact_vs_oracle(L):
    N, H, W, seed, off = 5, 20, 24, 100, 7
    dev = "cuda"
    rgb = torch.zeros(N, H, W, 3, dtype=torch.uint8, device=dev)
    depth = torch.zeros(N, H, W, 1, device=dev)
    goal = torch.zeros(N, 2, device=dev)
    rew = torch.zeros(N, device=dev)
    nd = torch.zeros(N, dtype=torch.uint8, device=dev)
    et = torch.zeros(N, dtype=torch.int64, device=dev)
    since = torch.zeros(N, dtype=torch.int64, device=dev)
    env = synth.SyntheticEnvs(N, H, W, seed=seed, env_offset=off)
    o = env.reset()
    ck(L.hab_synth_step(P(rgb), P(depth), P(goal), P(rew), P(nd), P(et), P(since), seed, off, N, H, W, 0, S()))
    assert np.array_equal(rgb.cpu().numpy(), o["rgb"])
    assert np.array_equal(depth.cpu().numpy(), o["depth"])
    assert np.array_equal(goal.cpu().numpy(), o["pointgoal_with_gps_compass"])
    for _ in range(30):
        o, r, d = env.step()
        ck(L.hab_synth_step(P(rgb), P(depth), P(goal), P(rew), P(nd), P(et), P(since), seed, off, N, H, W, 1, S()))
        assert np.array_equal(rgb.cpu().numpy(), o["rgb"])
        assert np.array_equal(depth.cpu().numpy(), o["depth"])
        assert np.array_equal(goal.cpu().numpy(), o["pointgoal_with_gps_compass"])
        assert np.array_equal(rew.cpu().numpy(), r)
        assert np.array_equal(nd.cpu().numpy().astype(bool), ~d)


def test_synth_objectnav_sensors_bit_exact_vs_oracle(L):
    N, H, W, seed, off = 4, 12, 16, 100, 3
    dev = "cuda"
    sem = torch.zeros(N, H, W, 1, dtype=torch.int32, device=dev)
    og = torch.zeros(N, 1, dtype=torch.int64, device=dev)
    cp, gps = torch.zeros(N, 1, device=dev), torch.zeros(N, 2, device=dev)
    rew, nd = torch.zeros(N, device=dev), torch.zeros(N, dtype=torch.uint8, device=dev)
    et, since = torch.zeros(N, dtype=torch.int64, device=dev), torch.zeros(N, dtype=torch.int64, device=dev)
    env = synth.SyntheticEnvs(N, H, W, seed=seed, env_offset=off, task="objectnav")
    o = env.reset()
    for step in range(6):
        ck(L.hab_synth_step(None, None, None, P(rew), P(nd), P(et), P(since), seed, off, N, H, W, int(step > 0), S()))
        ck(L.hab_synth_objectnav_sensors(P(sem), P(og), P(cp), P(gps), P(et), seed, off, N, H, W, S()))
        assert np.array_equal(sem.cpu().numpy(), o["semantic"]) and np.array_equal(og.cpu().numpy(), o["objectgoal"])
        assert np.array_equal(cp.cpu().numpy(), o["compass"]) and np.array_equal(gps.cpu().numpy(), o["gps"])
        o, r, d = env.step()


@pytest.mark.parametrize("T,N", [(128, 64), (5, 3), (1, 1), (33, 70), (200, 2)])
@pytest.mark.parametrize("use_gae", [1, 0])
def test_compute_returns(L, T, N, use_gae):
    torch.manual_seed(T * 100 + N)
    rewards = torch.randn(T + 1, N, 1)
    vp = torch.randn(T + 1, N, 1)
    masks = torch.rand(T + 1, N, 1) > 0.1
    nv = torch.randn(N, 1)
    ref, ref_vp = O.compute_returns(rewards, vp, masks, nv, T, bool(use_gae), 0.99, 0.95)
    for variant, exact in ((0, True), (1, False)):
        if variant == 1 and not use_gae:
            continue
        d = [t.cuda().contiguous() for t in (rewards, vp, masks, torch.zeros(T + 1, N, 1), nv)]
        ck(L.hab_compute_returns(P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), T, N, 0.99, 0.95, use_gae, variant, S()))
        got = d[3].cpu()
        rows = T if use_gae else T + 1
        if exact:  # same operation order as the reference loop: bitwise
            assert torch.equal(got[:rows], ref[:rows])
        else:      # wavefront scan: different association, fp32 round-off
            assert torch.allclose(got[:rows], ref[:rows], rtol=1e-5, atol=1e-5)
        if use_gae:
            assert torch.equal(d[1].cpu(), ref_vp)


def test_returns_scan_linearity_full_size(L):
    """Size-independent property at the benchmark shape: GAE is linear in (rewards, values)."""
    T, N = 128, 64
    torch.manual_seed(0)
    masks = (torch.rand(T + 1, N, 1) > 0.04).cuda()
    def run(r, v, nv):
        ret = torch.zeros(T + 1, N, 1, device="cuda")
        v = v.clone()
        ck(L.hab_compute_returns(P(r), P(v), P(masks), P(ret), P(nv), T, N, 0.99, 0.95, 1, 1, S()))
        return ret[:T]
    r1, v1, n1 = (torch.randn(T + 1, N, 1, device="cuda"), torch.randn(T + 1, N, 1, device="cuda"), torch.randn(N, 1, device="cuda"))
    r2, v2, n2 = (torch.randn(T + 1, N, 1, device="cuda"), torch.randn(T + 1, N, 1, device="cuda"), torch.randn(N, 1, device="cuda"))
    a, b, c = run(r1, v1, n1), run(r2, v2, n2), run(r1 + 2 * r2, v1 + 2 * v2, n1 + 2 * n2)
    assert torch.allclose(c, a + 2 * b, rtol=1e-4, atol=1e-4)


def test_advantages(L):
    torch.manual_seed(1)
    ret, vp = torch.randn(129, 64, 1), torch.randn(129, 64, 1)
    for mode, norm in ((0, False), (1, True)):
        ref = O.get_advantages(ret, vp, norm)
        adv = torch.zeros(129, 64, 1, device="cuda")
        stats = torch.zeros(4, device="cuda")
        rd, vd = ret.cuda(), vp.cuda()  # keep the device copies alive across the call
        ck(L.hab_advantages(P(rd), P(vd), P(adv), ret.numel(), mode, None, P(stats), S()))
        assert torch.allclose(adv.cpu(), ref, rtol=1e-5, atol=1e-6)
    # three-phase distributed form on one rank == biased normalisation (ddppo.py:59-84)
    ref = O.get_advantages(ret, vp, True, world_size=2)
    adv = torch.zeros(129, 64, 1, device="cuda")
    stats = torch.zeros(4, device="cuda")
    r, v = ret.cuda(), vp.cuda()
    ck(L.hab_advantages(P(r), P(v), P(adv), ret.numel(), 2, None, P(stats), S()))
    ck(L.hab_advantages(P(r), P(v), P(adv), ret.numel(), 4, P(stats), P(stats), S()))
    ck(L.hab_advantages(P(r), P(v), P(adv), ret.numel(), 3, P(stats), None, S()))
    assert torch.allclose(adv.cpu(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("clip_value", [1, 0])
def test_ppo_loss_fwd_bwd(L, clip_value):
    torch.manual_seed(2)
    B, rows_total = 1000, 1500
    v = torch.randn(B, 1, requires_grad=True)
    lp = (torch.randn(B, 1) * 0.3 - 1.2).requires_grad_()
    ent = (torch.rand(B, 1) + 0.5).requires_grad_()
    rows = torch.randperm(rows_total)[:B].int()
    store = {k: torch.randn(rows_total, 1) for k in ("action_log_probs", "advantages", "value_preds", "returns")}
    store["action_log_probs"] = store["action_log_probs"] * 0.3 - 1.2
    # make some ratios land exactly inside / on / outside the clip range and some zero advantages
    store["advantages"][rows[:10].long()] = 0.0
    batch = {k: t[rows.long()] for k, t in store.items()}
    total, vl, al, de, ratio = O.ppo_loss(v, lp, ent, batch, 0.2, 0.5, 0.01, bool(clip_value))
    total.backward()
    dev = lambda t: t.detach().reshape(-1).cuda().contiguous()
    dv, dlp, dent = (torch.zeros(B, device="cuda") for _ in range(3))
    out = torch.zeros(16, device="cuda")
    ck(L.hab_ppo_loss(P(dev(v)), P(dev(lp)), P(dev(ent)), P(dev(store["action_log_probs"])), P(dev(store["advantages"])),
                      P(dev(store["value_preds"])), P(dev(store["returns"])), P(rows.cuda()), B, 0.2, 0.5, 0.01, clip_value,
                      P(dv), P(dlp), P(dent), P(out), S()))
    o = out.cpu()
    assert abs(o[0] - vl.item()) <= 1e-5 * max(1, abs(vl.item()))
    assert abs(o[1] - al.item()) <= 1e-5 * max(1, abs(al.item()))
    assert abs(o[2] - de.item()) <= 1e-5
    assert abs(o[3] - total.item()) <= 1e-5 * max(1, abs(total.item()))
    r = ratio.detach()
    ref_m = [v.min(), v.mean(), v.max(), r.min(), r.mean(), r.max(), (r > 1.2).float().mean() + (r < 0.8).float().mean()]
    assert torch.allclose(o[4:11], torch.stack([x.detach() for x in ref_m]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(dv.cpu(), v.grad.view(-1), rtol=1e-5, atol=1e-9)
    assert torch.allclose(dlp.cpu(), lp.grad.view(-1), rtol=1e-4, atol=1e-9)
    assert torch.allclose(dent.cpu(), ent.grad.view(-1), rtol=1e-6, atol=1e-12)


def test_clip_adam_vs_torch(L):
    torch.manual_seed(3)
    n = 100003
    p0, g = torch.randn(n + 1)[: n + 1], torch.randn(n + 1) * 0.01
    for max_norm, gscale in ((0.5, 1.0), (1e9, 0.5)):
        p = p0.clone()
        pt = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([pt], lr=2.5e-4, eps=1e-5)
        m, v = torch.zeros(n + 1, device="cuda"), torch.zeros(n + 1, device="cuda")
        pd = p.cuda()
        scratch = torch.zeros(1024, dtype=torch.float64, device="cuda")
        gn = torch.zeros(1, device="cuda")
        for step in range(1, 4):
            gs = g * step
            pt.grad = (gs * gscale).clone()
            norm_ref = torch.nn.utils.clip_grad_norm_([pt], max_norm)
            opt.step()
            ck(L.hab_clip_adam_step(P(pd), P(gs.cuda()), P(m), P(v), n + 1, P(scratch), 1024, gscale, max_norm, 2.5e-4, 0.9, 0.999,
                                    1e-5, step, P(gn), S()))
            assert abs(gn.item() - norm_ref.item()) <= 1e-5 * norm_ref.item()
            assert torch.allclose(pd.cpu(), pt.detach(), rtol=1e-5, atol=2e-7)


def test_sample_actions_bit_exact_vs_multinomial(L):
    torch.manual_seed(4)
    for n, A in ((64, 4), (7, 6), (1000, 4)):
        probs = torch.softmax(torch.randn(n, A) * 2, -1)
        torch.manual_seed(n)
        ref = torch.multinomial(probs, 1, True)
        torch.manual_seed(n)
        q = torch.empty(n, A).exponential_(1)
        act = torch.zeros(n, dtype=torch.int64, device="cuda")
        ck(L.hab_sample_actions(P(probs.cuda()), P(q.cuda()), P(act), n, A, 0, S()))
        assert torch.equal(act.cpu().view(-1, 1), ref)
        ck(L.hab_sample_actions(P(probs.cuda()), None, P(act), n, A, 1, S()))
        assert torch.equal(act.cpu(), probs.argmax(-1))


# ------------------------------------------------------------------------------------------------
def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def repack(L, w, cpad=None):
    Cout, Cin, KH, KW = w.shape
    cpad = cpad or Cin
    wd = w.cuda().contiguous()
    wf = torch.zeros(Cout, KH, KW, cpad, device="cuda")
    wdg = torch.zeros(Cin, KH, KW, Cout, device="cuda") if cpad == Cin else None
    ck(L.hab_repack_conv_weight(P(wd), P(wf), P(wdg), Cout, Cin, KH, KW, cpad, S()))
    return wf, wdg


CONVS = [  # B, H, W, C, Cout, K, stride, pad
    (3, 63, 63, 32, 64, 4, 2, 0),    # SimpleCNN conv2 @256
    (3, 30, 30, 64, 32, 3, 1, 0),    # SimpleCNN conv3 @256
    (2, 32, 32, 32, 32, 3, 1, 1),    # resnet18 layer1
    (2, 32, 32, 32, 64, 3, 2, 1),    # layer2 stride 2
    (2, 32, 32, 32, 64, 1, 2, 0),    # downsample
    (2, 8, 8, 128, 256, 3, 2, 1),    # layer4 (N = 256 tile path)
    (1, 4, 4, 256, 128, 3, 1, 1),    # compression (tiny M)
    (2, 64, 64, 4, 32, 7, 2, 3),     # stem on 4 channels
    (4, 8, 8, 128, 128, 3, 1, 1),    # resnet18 layer3 @256^2: M = 256 (two full 128-row tiles), split-K
    (4, 16, 16, 64, 64, 3, 1, 1),    # layer2 @256^2
    (4, 16, 16, 64, 128, 3, 2, 1),   # layer3.0 stride 2
    (4, 16, 16, 64, 128, 1, 2, 0),   # layer3.0 downsample
    (3, 128, 128, 4, 32, 7, 2, 3),   # stem @256^2
    (2, 16, 18, 32, 32, 4, 2, 1),    # kernel = 2 x stride with padding: merged-class data gradient, N = 128
    (2, 17, 15, 16, 32, 2, 2, 0),    # kernel = stride, odd extent: merged-class data gradient, N = 64
    (65, 63, 63, 32, 64, 4, 2, 0),   # SimpleCNN conv2, tiles that straddle images
]


@pytest.mark.parametrize("B,H,W,Cc,Cout,K,s,p", CONVS)
@pytest.mark.parametrize("use_ws", [0, 1])
def test_conv_fwd_dgrad_wgrad(L, B, H, W, Cc, Cout, K, s, p, use_ws):
    torch.manual_seed(B * 1000 + H + Cout)
    x = torch.randn(B, Cc, H, W, requires_grad=True)
    w = (torch.randn(Cout, Cc, K, K) / np.sqrt(Cc * K * K)).requires_grad_()
    b = torch.randn(Cout)
    y_ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p))
    wf, wdg = repack(L, w.detach())
    xh = nhwc(x.detach()).cuda()
    Ho, Wo = y_ref.shape[2:]
    ws = torch.zeros(1 << 22, device="cuda") if use_ws else None
    wsn = ws.numel() if use_ws else 0
    y = torch.zeros(B, Ho, Wo, Cout, device="cuda")
    ck(L.hab_conv2d_fwd(P(xh), P(wf), P(b.cuda()), P(y), B, H, W, Cc, Cout, K, K, s, p, 1, P(ws), wsn, S()))
    assert torch.allclose(y.cpu(), nhwc(y_ref), atol=2e-5, rtol=1e-4)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dy = nhwc(gy * (y_ref > 0)).cuda()
    if Cc % 4 == 0:
        dx = torch.full((B, H, W, Cc), 7.0, device="cuda")  # every element must be written
        ck(L.hab_conv2d_dgrad(P(dy), P(wdg), None, None, P(dx), B, H, W, Cc, Cout, K, K, s, p, P(ws), wsn, S()))
        assert torch.allclose(dx.cpu(), nhwc(x.grad), atol=2e-5, rtol=1e-4)
        # fused epilogue: residual-gradient add, then the producer's ReLU mask
        m = torch.randn(B, H, W, Cc)
        add = torch.randn(B, H, W, Cc)
        dx2 = torch.full((B, H, W, Cc), 7.0, device="cuda")
        ck(L.hab_conv2d_dgrad(P(dy), P(wdg), P(m.cuda()), P(add.cuda()), P(dx2), B, H, W, Cc, Cout, K, K, s, p, P(ws), wsn, S()))
        assert torch.allclose(dx2.cpu(), (nhwc(x.grad) + add) * (m > 0), atol=2e-5, rtol=1e-4)
        dw = torch.zeros(Cout, Cc, K, K, device="cuda")
        db = torch.full((Cout,), 7.0, device="cuda")
        ck(L.hab_conv2d_wgrad(P(xh), P(dy), P(dw), P(db), B, H, W, Cc, Cout, K, K, s, p, P(ws), wsn, S()))
        scale = w.grad.abs().max().item()
        assert (dw.cpu() - w.grad).abs().max().item() <= 1e-4 * scale + 1e-5
        db_ref = (gy * (y_ref > 0)).sum((0, 2, 3))  # fused bias gradient = column sums of dY
        assert torch.allclose(db.cpu(), db_ref, rtol=1e-4, atol=1e-4 * db_ref.abs().max().item() + 1e-5)


@pytest.mark.parametrize("has_rgb,has_depth,H,W,B", [(1, 1, 256, 256, 2), (0, 1, 84, 84, 4), (1, 0, 64, 96, 3)])
def test_obs_conv(L, has_rgb, has_depth, H, W, B):
    torch.manual_seed(5)
    nrows = B + 3
    rgb = torch.randint(0, 256, (nrows, H, W, 3), dtype=torch.uint8) if has_rgb else None
    depth = torch.rand(nrows, H, W, 1) if has_depth else None
    rows = torch.randperm(nrows)[:B].int()
    obs = {}
    if has_rgb:
        obs["rgb"] = rgb[rows.long()]
    if has_depth:
        obs["depth"] = depth[rows.long()]
    x = O.simple_cnn_input(obs)
    Cin = x.shape[1]
    w = (torch.randn(32, Cin, 8, 8) / np.sqrt(Cin * 64)).requires_grad_()
    b = torch.randn(32)
    y_ref = F.relu(F.conv2d(x, w, b, stride=4))
    wf, _ = repack(L, w.detach())
    Ho, Wo = y_ref.shape[2:]
    y = torch.zeros(B, Ho, Wo, 32, device="cuda")
    ws = torch.zeros(1 << 22, device="cuda")
    rg, dp = (rgb.cuda() if has_rgb else None), (depth.cuda() if has_depth else None)
    ck(L.hab_obs_conv2d_fwd(P(rg), P(dp), P(rows.cuda()), P(wf), P(b.cuda()), P(y), B, H, W, 32, 8, 8, 4, 0, 1, P(ws), ws.numel(), S()))
    assert torch.allclose(y.cpu(), nhwc(y_ref), atol=2e-5, rtol=1e-4)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dy = nhwc(gy * (y_ref > 0)).cuda()
    dw = torch.zeros(32, Cin, 8, 8, device="cuda")
    db = torch.full((32,), 7.0, device="cuda")
    ck(L.hab_obs_conv2d_wgrad(P(rg), P(dp), P(rows.cuda()), P(dy), P(dw), P(db), B, H, W, 32, 8, 8, 4, 0, P(ws), ws.numel(), S()))
    assert (dw.cpu() - w.grad).abs().max().item() <= 1e-4 * w.grad.abs().max().item() + 1e-5
    db_ref = (gy * (y_ref > 0)).sum((0, 2, 3))
    assert torch.allclose(db.cpu(), db_ref, rtol=1e-4, atol=1e-4 * db_ref.abs().max().item() + 1e-5)
    ck(L.hab_obs_conv2d_wgrad(P(rg), P(dp), P(rows.cuda()), P(dy), P(dw), None, B, H, W, 32, 8, 8, 4, 0, None, 0, S()))  # no ws, no bias
    assert (dw.cpu() - w.grad).abs().max().item() <= 1e-4 * w.grad.abs().max().item() + 1e-5


@pytest.mark.parametrize("M,N,K", [(64, 512, 25088), (300, 1536, 514), (4096, 4, 512), (37, 96, 70), (2048, 512, 1568)])
def test_linear_family(L, M, N, K):
    torch.manual_seed(M + N)
    ldk = (K + 3) // 4 * 4
    x = torch.randn(M, K, requires_grad=True)
    w = (torch.randn(N, K) / np.sqrt(K)).requires_grad_()
    b = torch.randn(N)
    y_ref = F.relu(F.linear(x, w, b))
    ws = torch.zeros(1 << 24, device="cuda")
    xd, wd = torch.zeros(M, ldk, device="cuda"), w.detach().cuda().contiguous()
    xd[:, :K] = x.detach()
    ldn = (N + 3) // 4 * 4
    y = torch.zeros(M, ldn, device="cuda")
    ck(L.hab_linear_fwd(P(xd), ldk, P(wd), K, P(b.cuda()), P(y), ldn, M, N, K, 1, 0, P(ws), ws.numel(), S()))
    assert torch.allclose(y[:, :N].cpu(), y_ref, atol=5e-5, rtol=1e-4)
    gy = torch.randn(M, N)
    y_ref.backward(gy)
    dyp = torch.zeros(M, ldn, device="cuda")
    dyp[:, :N] = (gy * (y_ref > 0)).cuda()
    wpad = torch.zeros(N, ldk, device="cuda")
    wpad[:, :K] = w.detach()
    dx = torch.zeros(M, ldk, device="cuda")
    ck(L.hab_linear_dgrad(P(dyp), ldn, P(wpad), ldk, None, 0, P(dx), ldk, M, K, N, 0, P(ws), ws.numel(), S()))
    assert torch.allclose(dx[:, :K].cpu(), x.grad, atol=5e-5, rtol=1e-4)
    dw = torch.zeros(N, K, device="cuda")
    ck(L.hab_linear_wgrad(P(dyp), ldn, P(xd), ldk, P(dw), K, M, N, K, 0, 0, 0, P(ws), ws.numel(), S()))
    assert (dw.cpu() - w.grad).abs().max().item() <= 1e-4 * w.grad.abs().max().item() + 1e-5
    db = torch.zeros(N, device="cuda")
    ck(L.hab_colsum(P(dyp), ldn, M, N, P(db), 0, P(ws), ws.numel(), S()))
    assert torch.allclose(db.cpu(), (gy * (y_ref > 0)).sum(0), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("M,N,K,perm", [(512, 512, 3136, True), (300, 384, 1024, False), (256, 128, 256, False), (1100, 260, 608, False),
                                         (640, 2048, 576, False),
                                         # >= 128 output tiles: the weight gradient runs without split-K and writes the permutation itself
                                         # (8 hw x 16 channel tile columns, full 32-byte sectors)
                                         (512, 512, 8192, True)])
def test_dense_gemm_kernel_vs_float64(L, M, N, K, perm):
    """csrc/dense_bf3.h (matrix-path bits 10 + 11: every applicable shape) through hab_linear_fwd / _dgrad / _wgrad against float64:
    bias + ReLU and split-K slabs (forward), a k-strided weight operand (data gradient), two k-strided operands + the NHWC-flatten ->
    NCHW-flatten column permutation + accumulate (weight gradient), ragged M / N tiles.  Same error bar as the igemm kernels."""
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    dy = torch.randn(M, N, device="cuda", generator=g)
    ws = torch.zeros(1 << 25, device="cuda")
    prev = L.hab_set_matrix_path(4095)
    try:
        ldn = N + 4  # (padded output rows: the vector stores must respect the leading dimension)
        y = torch.full((M, ldn), 7.0, device="cuda")
        ck(L.hab_linear_fwd(P(x), K, P(w), K, P(b), P(y), ldn, M, N, K, 1, 0, P(ws), ws.numel(), S()))
        ref = torch.relu(x.double() @ w.double().t() + b.double())
        assert float((y[:, :N].double() - ref).norm() / ref.norm()) <= 2e-6 and bool((y[:, N:] == 7.0).all())
        dx = torch.zeros(M, K, device="cuda")
        ck(L.hab_linear_dgrad(P(dy), N, P(w), K, None, 0, P(dx), K, M, K, N, 0, P(ws), ws.numel(), S()))
        ref = dy.double() @ w.double()
        assert float((dx.double() - ref).norm() / ref.norm()) <= 2e-6
        old = torch.randn(N, K, device="cuda", generator=g)
        dw = old.clone()
        pc, ph = (32, K // 32) if perm else (0, 0)
        ck(L.hab_linear_wgrad(P(dy), N, P(x), K, P(dw), K, M, N, K, pc, ph, 1, P(ws), ws.numel(), S()))
        ref = dy.double().t() @ x.double()
        if perm:  # column hw * 32 + c of x is column c * HW + hw of the reference weight
            ref = ref.view(N, K // 32, 32).transpose(1, 2).reshape(N, K)
        assert float((dw.double() - old.double() - ref).norm() / ref.norm()) <= 2e-6
        # bitwise reproducible (fixed reduction order, no atomics)
        dw2 = old.clone()
        ck(L.hab_linear_wgrad(P(dy), N, P(x), K, P(dw2), K, M, N, K, pc, ph, 1, P(ws), ws.numel(), S()))
        assert torch.equal(dw, dw2)
    finally:
        L.hab_set_matrix_path(prev)


def test_igemm_transpose_detecting_identity(L):
    """A = I with an asymmetric B: catches swapped fragment rows/cols (cdna guide, rule 16)."""
    M = N = K = 96
    x = torch.eye(M, K)
    w = torch.arange(N * K, dtype=torch.float32).view(N, K) / 100.0  # y = w^T
    y = torch.zeros(M, N, device="cuda")
    ck(L.hab_linear_fwd(P(x.cuda()), K, P(w.cuda()), K, None, P(y), N, M, N, K, 0, 0, None, 0, S()))
    assert torch.equal(y.cpu(), w.t().contiguous())


# ------------------------------------------------------------------------------------------------
# HBM-bound kernels of the GroupNorm-ResNet encoder
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,HW,Cc,groups,relu,res", [(3, 64 * 64, 32, 16, 1, 0), (2, 100, 64, 16, 1, 1), (5, 16, 256, 16, 0, 0),
                                                      (2, 4, 1024, 16, 1, 1), (3, 16, 128, 1, 1, 0), (2, 1, 2048, 1, 1, 0),
                                                      # register-resident kernels at 1024 threads per frame (resnet18 layer1 / layer2 @256^2)
                                                      (3, 32 * 32, 32, 16, 1, 1), (2, 16 * 16, 64, 16, 1, 0), (2, 30 * 30, 32, 16, 0, 1),
                                                      (2, 8 * 8, 128, 16, 1, 1),
                                                      # chunk-parallel path (frames > 128 KB): stem 64x64x32, resnet50 32x32x128 and
                                                      # 16x16x256, a ragged last chunk, one channel group
                                                      (2, 32 * 32, 128, 16, 1, 1), (2, 16 * 16, 256, 16, 0, 0), (3, 50 * 41, 32, 16, 1, 1),
                                                      (2, 40 * 40, 64, 1, 1, 0)])
@pytest.mark.parametrize("with_ws", [1, 0])
def test_groupnorm_fwd_bwd(L, B, HW, Cc, groups, relu, res, with_ws):
    if not with_ws and HW * Cc <= 32768:
        pytest.skip("scratch only matters for frames > 128 KB")
    gws = torch.zeros(1 << 20, device="cuda") if with_ws else None
    torch.manual_seed(B * 10 + Cc)
    x = torch.randn(B, Cc, HW, 1, requires_grad=True)
    g = (1 + 0.1 * torch.randn(Cc)).requires_grad_()
    b = (0.1 * torch.randn(Cc)).requires_grad_()
    r = torch.randn(B, Cc, HW, 1, requires_grad=True) if res else None
    y_ref = F.group_norm(x, groups, g, b, eps=1e-5)
    if res:
        y_ref = y_ref + r
    if relu:
        y_ref = F.relu(y_ref)
    xh, rh = nhwc(x.detach()).cuda(), (nhwc(r.detach()).cuda() if res else None)
    y = torch.zeros(B, HW, Cc, device="cuda")
    mean, rstd = torch.zeros(B, groups, device="cuda"), torch.zeros(B, groups, device="cuda")
    ck(L.hab_groupnorm_fwd(P(xh), P(y), P(g.detach().cuda()), P(b.detach().cuda()), P(rh), P(mean), P(rstd), B, HW, Cc, groups, relu, 1e-5, P(gws), gws.numel() if gws is not None else 0, S()))
    assert torch.allclose(y.cpu().view(B, HW, 1, Cc), nhwc(y_ref), atol=2e-5, rtol=1e-4)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dy = nhwc(gy).cuda().view(B, HW, Cc)
    dx = torch.zeros(B, HW, Cc, device="cuda")
    dym = torch.zeros(B, HW, Cc, device="cuda")
    cs = torch.zeros(B, 2, Cc, device="cuda")
    ck(L.hab_groupnorm_bwd(P(xh), P(dy), P(y) if relu else None, P(dx), P(dym), P(g.detach().cuda()), P(mean), P(rstd), P(cs), B, HW, Cc,
                           groups, P(gws), gws.numel() if gws is not None else 0, S()))
    # without the optional dy' output and in place (dx aliases dy), as the engine calls it for non-residual layers
    dy2 = dy.clone()
    ck(L.hab_groupnorm_bwd(P(xh), P(dy2), P(y) if relu else None, P(dy2), None, P(g.detach().cuda()), P(mean), P(rstd), P(cs), B, HW, Cc,
                           groups, P(gws), gws.numel() if gws is not None else 0, S()))
    assert torch.equal(dy2, dx)
    scale = x.grad.abs().max().item()
    assert (dx.cpu().view(B, HW, 1, Cc) - nhwc(x.grad)).abs().max().item() <= 1e-4 * scale + 1e-6
    assert torch.allclose(cs[:, 0].sum(0).cpu(), b.grad, rtol=1e-4, atol=1e-4 * b.grad.abs().max().item())
    assert torch.allclose(cs[:, 1].sum(0).cpu(), g.grad, rtol=1e-4, atol=1e-4 * g.grad.abs().max().item())
    if res:  # the masked dy is the gradient of the residual branch
        assert torch.allclose(dym.cpu().view(B, HW, 1, Cc), nhwc(r.grad), atol=1e-6)


@pytest.mark.parametrize("B,H,W,Cc", [(2, 64, 64, 32), (3, 9, 13, 32), (1, 7, 7, 64)])
def test_maxpool_fwd_bwd(L, B, H, W, Cc):
    torch.manual_seed(H)
    x = torch.relu(torch.randn(B, Cc, H, W)).requires_grad_()  # post-ReLU input: many exact ties at 0
    y_ref = F.max_pool2d(x, 3, 2, 1)
    Ho, Wo = y_ref.shape[2:]
    y = torch.zeros(B, Ho, Wo, Cc, device="cuda")
    idx = torch.zeros(B, Ho, Wo, Cc, dtype=torch.uint8, device="cuda")
    ck(L.hab_maxpool3x3s2_fwd(P(nhwc(x.detach()).cuda()), P(y), P(idx), B, H, W, Cc, S()))
    assert torch.equal(y.cpu(), nhwc(y_ref))
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dx = torch.zeros(B, H, W, Cc, device="cuda")
    ck(L.hab_maxpool3x3s2_bwd(P(nhwc(gy).cuda()), P(idx), P(dx), B, H, W, Cc, S()))
    assert torch.allclose(dx.cpu(), nhwc(x.grad), atol=1e-6)  # same arg-max choice as ATen on ties


@pytest.mark.parametrize("keys", [("rgb", "depth"), ("depth", "rgb"), ("depth",), ("rgb",), ("rgb", "depth", "semantic"),
                                  ("semantic", "rgb", "depth")])
def test_ingest_and_running_mean_var(L, keys):
    torch.manual_seed(3)
    nrows, B, H, W = 7, 4, 20, 24
    rows = torch.randperm(nrows)[:B].int()
    obs_all = {"rgb": torch.randint(0, 256, (nrows, H, W, 3), dtype=torch.uint8), "depth": torch.rand(nrows, H, W, 1),
               "semantic": torch.randint(0, 40, (nrows, H, W, 1), dtype=torch.int32)}
    obs = {k: obs_all[k][rows.long()] for k in keys}
    x_ref = O.resnet_input(obs, list(keys))
    n_in = x_ref.shape[1]
    cpad = 4 if n_in <= 4 else 8
    off, o = {"rgb": -1, "depth": -1, "semantic": -1}, 0
    for k in keys:
        off[k] = o
        o += 3 if k == "rgb" else 1
    y = torch.full((B, H // 2, W // 2, cpad), 9.0, device="cuda")
    dev = {k: obs_all[k].cuda() for k in keys}
    ck(L.hab_obs_ingest_pool(P(dev.get("rgb")), P(dev.get("depth")), P(dev.get("semantic")), P(rows.cuda()), P(y), B, H, W, cpad,
                             off["rgb"], off["depth"], off["semantic"], S()))
    assert torch.equal(y.cpu()[..., :n_in], nhwc(x_ref)), "uint8 scaling + 2x2 average must be bitwise the reference's arithmetic"
    assert n_in == cpad or float(y[..., n_in:].abs().max()) == 0.0
    if cpad != 4:
        return
    mean0, var0, cnt0 = torch.rand(1, n_in, 1, 1), torch.rand(1, n_in, 1, 1) * 0.1, torch.tensor(5.0)
    xn_ref, m_ref, v_ref, c_ref = O.running_mean_and_var(x_ref, mean0, var0, cnt0, True)
    npix = B * (H // 2) * (W // 2)
    st = torch.zeros(16, device="cuda")
    scratch = torch.zeros(1024 * 4, dtype=torch.float64, device="cuda")
    rm, rv, rc = mean0.view(-1).cuda().contiguous(), var0.view(-1).cuda().contiguous(), cnt0.view(1).cuda().contiguous()
    ck(L.hab_channel_moments(P(y), npix, 4, 0, None, P(st), P(scratch), scratch.numel(), S()))
    ck(L.hab_channel_moments(P(y), npix, 4, 1, P(st), P(st[8:]), P(scratch), scratch.numel(), S()))
    ck(L.hab_running_mean_var_update(P(rm), P(rv), P(rc), P(st), P(st[8:]), float(B), n_in, S()))
    ck(L.hab_running_mean_var_normalize(P(y), npix, 4, n_in, P(rm), P(rv), S()))
    assert torch.allclose(rm.cpu(), m_ref.view(-1), rtol=1e-5, atol=1e-6) and torch.allclose(rv.cpu(), v_ref.view(-1), rtol=1e-5, atol=1e-6)
    assert float(rc.cpu()) == float(c_ref)
    assert torch.allclose(y.cpu()[..., :n_in], nhwc(xn_ref), rtol=1e-5, atol=1e-5)


def test_nav_embeddings_fwd_bwd(L):
    """All five 1-D sensor embedding kinds of PointNavResNetNet.forward (PointNav: goal + previous action; ObjectNav: objectgoal,
    compass, gps + previous action), forward and the deterministic backward reduction."""
    from habitat_amd._lib import EmbedSlot
    torch.manual_seed(4)
    B, nrows, ld, col0, A, ncat = 300, 350, 200, 12, 6, 21
    rows = torch.randperm(nrows)[:B].int()
    ridx = rows.long()
    goal = torch.stack([torch.rand(nrows) * 5, (torch.rand(nrows) - 0.5) * 6], 1)
    objg = torch.randint(0, ncat, (nrows, 1))
    compass = (torch.rand(nrows, 1) - 0.5) * 6
    gps = torch.randn(nrows, 2)
    pa = torch.randint(0, A, (nrows, 1))
    masks = torch.rand(nrows, 1) > 0.3
    w = {"tgt": torch.randn(32, 3), "tgt_b": torch.randn(32), "obj": torch.randn(ncat, 32), "cmp": torch.randn(32, 2), "cmp_b": torch.randn(32),
         "gps": torch.randn(32, 2), "gps_b": torch.randn(32), "emb": torch.randn(A + 1, 32)}
    for v in w.values():
        v.requires_grad_()
    g = goal[ridx]
    tok = torch.where(masks[ridx].view(-1), pa[ridx].view(-1) + 1, torch.zeros(B, dtype=torch.long))
    c = compass[ridx]
    ref = torch.cat([F.linear(torch.stack([g[:, 0], torch.cos(-g[:, 1]), torch.sin(-g[:, 1])], -1), w["tgt"], w["tgt_b"]),
                     F.embedding(objg[ridx], w["obj"]).squeeze(1),
                     F.linear(torch.stack([torch.cos(c), torch.sin(c)], -1).squeeze(1), w["cmp"], w["cmp_b"]),
                     F.linear(gps[ridx], w["gps"], w["gps_b"]), F.embedding(tok, w["emb"])], 1)
    dw = {k: torch.zeros_like(v, device="cuda") for k, v in w.items()}
    wd = {k: v.detach().cuda() for k, v in w.items()}
    spec = [(0, goal, "tgt", "tgt_b", 0), (1, objg, "obj", None, ncat), (2, compass, "cmp", "cmp_b", 0), (3, gps, "gps", "gps_b", 0),
            (4, pa, "emb", None, A + 1)]
    slots = (EmbedSlot * 5)()
    for i, (kind, inp, wk, bk, ntok) in enumerate(spec):
        slots[i].kind, slots[i].num_tokens = kind, ntok
        slots[i].input = P(inp.cuda()).value
        slots[i].weight, slots[i].bias = wd[wk].data_ptr(), (wd[bk].data_ptr() if bk else None)
        slots[i].d_weight, slots[i].d_bias = dw[wk].data_ptr(), (dw[bk].data_ptr() if bk else None)
    out = torch.zeros(B, ld, device="cuda")
    saved = torch.zeros(B, 5, 4, device="cuda")
    ck(L.hab_nav_embed_fwd(slots, 5, P(masks.cuda()), P(rows.cuda()), P(out), ld, col0, B, P(saved), S()))
    assert torch.allclose(out.cpu()[:, col0:col0 + 160], ref, atol=1e-5, rtol=1e-5)
    gy = torch.randn(B, 160)
    ref.backward(gy)
    dout = torch.zeros(B, ld)
    dout[:, col0:col0 + 160] = gy
    ws = torch.zeros(1 << 18, device="cuda")
    ck(L.hab_nav_embed_bwd(slots, 5, P(saved), P(dout.cuda()), ld, col0, B, P(ws), ws.numel(), S()))
    for k, v in w.items():
        assert torch.allclose(dw[k].cpu(), v.grad, atol=2e-4, rtol=1e-4), k


def test_rollout_step_stats(L):
    """Fused per-step episode bookkeeping == the reference's elementwise sequence (ppo_trainer.py:417-446)."""
    torch.manual_seed(11)
    N = 70
    cur, sr, sc = torch.randn(N, 1), torch.randn(N, 1).abs(), torch.randint(0, 5, (N, 1)).float()
    d = [t.clone().cuda() for t in (cur, sr, sc)]
    prev = torch.zeros(N, 1, dtype=torch.long, device="cuda")
    for step in range(5):
        rewards = torch.randn(N, 1)
        not_done = torch.rand(N, 1) > 0.3
        actions = torch.randint(0, 4, (N, 1))
        done = ~not_done
        cur += rewards
        sr += cur.where(done, cur.new_zeros(()))
        sc += done.float()
        cur.masked_fill_(done, 0.0)
        ck(L.hab_rollout_step_stats(P(rewards.cuda()), P(not_done.cuda()), P(d[0]), P(d[1]), P(d[2]), P(actions.cuda()), P(prev), N, 1, S()))
        assert torch.equal(d[0].cpu(), cur) and torch.equal(d[1].cpu(), sr) and torch.equal(d[2].cpu(), sc)
        assert torch.equal(prev.cpu(), actions)
