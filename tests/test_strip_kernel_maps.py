"""CPU: the index maps of the strip-resident kernels (csrc/wgrad3x3_bf3.h, csrc/conv2_fwd_strip.h) replayed in numpy.

Everything that decides WHICH numbers meet in an MFMA is index arithmetic: the LDS image a strip is staged into (pixel-major rows,
channel halves, even / odd column order, XOR-swizzled chunks), the per-lane addresses of the fragment reads (incl. the LDS transpose
read, whose data movement tools/ubench/tr_read_probe.hip pins on the hardware: every 16-lane group reads a [4 rows][16 columns] block,
lane i supplies the address of 8-byte chunk i -- row i // 4, chunk i % 4 -- and receives column i of the four rows), the MFMA operand /
accumulator lane layouts, the wave decomposition, the slab / partial-tile addresses of the epilogues.  This file restates those maps
from the headers (same formulas, same constants) and runs them on small integer-valued tensors, where float64 arithmetic is exact:
the result must equal the convolution's weight gradient / forward output element for element.  The operand split into three bf16
planes is exact arithmetic and orthogonal to the maps, so one "plane" holds the whole value here.  The GPU parity tests
(tests/test_gpu_bf3.py) check the kernels themselves; this test makes the design reviewable without a GPU."""
import numpy as np
import pytest


# ---- hardware models -------------------------------------------------------------------------------------------------------------
def tr_read_b64(lds, addr):
    """ds_read_b64_tr_b16: `addr[64]` element offsets (multiples of 4); returns [64][4]: lane i of a 16-lane group receives element
    (i % 16) of the group's 16-element rows, from the four rows whose chunks the lanes 4 r .. 4 r + 3 of the group addressed."""
    out = np.zeros((64, 4), lds.dtype)
    for g in range(4):
        lanes = np.arange(16) + 16 * g
        rows = [np.concatenate([lds[addr[lanes[4 * r + c]]: addr[lanes[4 * r + c]] + 4] for c in range(4)]) for r in range(4)]
        for i in range(16):
            out[lanes[i]] = [rows[r][i] for r in range(4)]
    return out


def mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag / b_frag: [64][8] (lane l: row / column l % 32, k = 8 (l // 32) .. +7); acc: [64][16] (lane l: column l % 32, rows
    (v & 3) + 8 (v >> 2) + 4 (l // 32))."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        A[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = a_frag[l]
        B[8 * (l // 32): 8 * (l // 32) + 8, l % 32] = b_frag[l]
    D = A @ B
    for l in range(64):
        for v in range(16):
            acc[l, v] += D[(v & 3) + 8 * (v >> 2) + 4 * (l // 32), l % 32]
    return acc


# ---- wgrad3x3_bf3.h ---------------------------------------------------------------------------------------------------------------
def wgrad_strip_model(x, dy, C32, N32, W, PAD, KHW, S, R, KS, NS, strips=None):
    """x [B][H][W][C], dy [B][Ho][Wo][N] -> dW [N][KHW][KHW][C] through the kernel's maps (strips: the strips to run; the others must
    hold zero dY -- they contribute exactly zero and only cost time)."""
    Bn, H = x.shape[0], x.shape[1]
    C, N, WP = 32 * C32, 32 * N32, W + 2 * PAD
    Wo = (WP - KHW) // S + 1
    Ho = dy.shape[1]
    NW = N32 // NS
    XRS = (R - 1) * S + KHW
    XROWS, NPIX = XRS * WP, R * Wo
    NQ = (NPIX + 3) // 4
    NSTEPS = (NQ + 3) // 4
    YROWS = NSTEPS * 16
    X_HALF, Y_HALF = XROWS * 32, YROWS * 32
    WH = (WP + 1) // 2
    xcol = (lambda w: (w & 1) * WH + (w >> 1)) if S == 2 else (lambda w: w)
    tap_col = (lambda kw: (kw & 1) * WH + (kw >> 1)) if S == 2 else (lambda kw: kw)
    nwaves = KHW * KS * NS
    slabs = np.zeros((KS, KHW * KHW * C, N))
    assert Ho % R == 0
    for img in range(Bn):
        for strip in (range(Ho // R) if strips is None else strips):
            ho0 = strip * R
            xs = np.zeros(C32 * X_HALF); ys = np.zeros(N32 * Y_HALF)
            # stage(): unit u = 4 channels of one pixel
            for u in range(XRS * W * C // 4):
                c4, pix = u % (C // 4), u // (C // 4)
                w, hh = pix % W, pix // W
                h = ho0 * S - PAD + hh
                v = x[img, h, w, 4 * c4: 4 * c4 + 4] if 0 <= h < H else np.zeros(4)
                dst = (c4 >> 3) * X_HALF + (hh * WP + xcol(w + PAD)) * 32 + (c4 & 7) * 4
                xs[dst: dst + 4] = v
            for u in range(R * Wo * N // 4):
                c4, pix = u % (N // 4), u // (N // 4)
                dst = (c4 >> 3) * Y_HALF + pix * 32 + (c4 & 7) * 4
                ys[dst: dst + 4] = dy[img, ho0 + pix // Wo, pix % Wo, 4 * c4: 4 * c4 + 4]
            for wave in range(nwaves):
                kh, ns, ks = wave % KHW, (wave // KHW) % NS, wave // (KHW * NS)
                acc = np.zeros((KHW, C32, NW, 64, 16))
                for s in range(ks, NSTEPS, KS):
                    yoff = np.zeros((2, 64), int); xoff = np.zeros((2, 64), int)
                    for lane in range(64):
                        i16, kblk = lane & 15, lane >> 5
                        chunk_off = ((lane >> 4) & 1) * 16 + (i16 & 3) * 4
                        for r in range(2):
                            pix = 4 * (4 * s + 2 * kblk + r) + (i16 >> 2)
                            yoff[r, lane] = pix * 32 + chunk_off
                            pc = pix if pix < NPIX else 0
                            hol, wo = pc // Wo, pc % Wo
                            xoff[r, lane] = ((hol * S + kh) * WP + (wo if S == 2 else wo * S)) * 32 + chunk_off
                    bfr = [np.concatenate([tr_read_b64(ys, (ns * NW + hn) * Y_HALF + yoff[0]),
                                           tr_read_b64(ys, (ns * NW + hn) * Y_HALF + yoff[1])], 1) for hn in range(NW)]
                    for kw in range(KHW):
                        for hc in range(C32):
                            base = hc * X_HALF + tap_col(kw) * 32
                            afr = np.concatenate([tr_read_b64(xs, base + xoff[0]), tr_read_b64(xs, base + xoff[1])], 1)
                            for hn in range(NW):
                                mfma_32x32x16(afr, bfr[hn], acc[kw, hc, hn])
                for kw in range(KHW):
                    for hc in range(C32):
                        for hn in range(NW):
                            for lane in range(64):
                                li, hi = lane & 31, lane >> 5
                                for v in range(16):
                                    ci = (v & 3) + 8 * (v >> 2) + 4 * hi
                                    slabs[ks, (kh * KHW + kw) * C + hc * 32 + ci, (ns * NW + hn) * 32 + li] += acc[kw, hc, hn, lane, v]
    dw = slabs.sum(0).reshape(KHW, KHW, C, N)  # row i = (kh*KHW + kw)*C + ci
    return dw.transpose(3, 0, 1, 2)


def wgrad_reference(x, dy, PAD, KHW, S):
    Bn, H, W, C = x.shape
    _, Ho, Wo, N = dy.shape
    xp = np.zeros((Bn, H + 2 * PAD, W + 2 * PAD, C)); xp[:, PAD: PAD + H, PAD: PAD + W] = x
    dw = np.zeros((N, KHW, KHW, C))
    for kh in range(KHW):
        for kw in range(KHW):
            patch = xp[:, kh: kh + S * (Ho - 1) + 1: S, kw: kw + S * (Wo - 1) + 1: S]  # [B][Ho][Wo][C]
            dw[:, kh, kw] = np.einsum("bhwn,bhwc->nc", dy, patch)
    return dw


@pytest.mark.parametrize("cfg", [dict(C32=2, N32=1, W=30, PAD=0, KHW=3, S=1, R=2, KS=2, NS=1, H=30),    # SimpleCNN conv3
                                 dict(C32=1, N32=1, W=32, PAD=1, KHW=3, S=1, R=2, KS=1, NS=1, H=32),    # ResNet layer1
                                 dict(C32=2, N32=2, W=16, PAD=1, KHW=3, S=1, R=4, KS=1, NS=2, H=16),    # ResNet layer2
                                 dict(C32=1, N32=2, W=63, PAD=0, KHW=4, S=2, R=2, KS=2, NS=1, H=63)])   # SimpleCNN conv2
def test_weight_gradient_strip_maps(cfg):
    rng = np.random.default_rng(0)
    H, W, PAD, KHW, S = cfg["H"], cfg["W"], cfg["PAD"], cfg["KHW"], cfg["S"]
    C, N = 32 * cfg["C32"], 32 * cfg["N32"]
    Ho = (H + 2 * PAD - KHW) // S + 1
    Wo = (W + 2 * PAD - KHW) // S + 1
    R = cfg["R"]
    # two strips of one image are enough to exercise every map (first strip: top border / padding row; the other: interior or bottom)
    x = rng.integers(-3, 4, (1, H, W, C)).astype(np.float64)
    dy = np.zeros((1, Ho, Wo, N))
    rows = list(range(0, R)) + list(range(Ho - R, Ho))
    dy[:, rows] = rng.integers(-3, 4, (1, len(rows), Wo, N))
    kw = {k: v for k, v in cfg.items() if k != "H"}
    got = wgrad_strip_model(x, dy, strips=(0, Ho // R - 1), **kw)
    ref = wgrad_reference(x, dy, PAD, KHW, S)
    assert np.array_equal(got, ref)


# ---- conv2_fwd_strip.h ------------------------------------------------------------------------------------------------------------
def conv2_fwd_strip_model(x, wf, bias, relu, R=2, strips=None):
    """x [B][H][W][32] (W <= 63), wf [64][4][4][32] -> y [B][Ho][Wo][64] through the kernel's maps (strips: subset of strip indices).
    63 x 63 is the compile-time instantiation; other geometries run the same maps with H, W, Ho, Wo from the arguments."""
    Bn, H, W = x.shape[0], x.shape[1], x.shape[2]
    Wo, Ho, WH = (W - 4) // 2 + 1, (H - 4) // 2 + 1, (W + 1) // 2
    XRS = 2 * (R - 1) + 4
    XROWS, NPIX = XRS * W, R * Wo
    NPT = (NPIX + 31) // 32
    RED_LD = 68
    xcol = lambda w: (w & 1) * WH + (w >> 1)
    y = np.full((Bn, Ho, Wo, 64), np.nan)
    for img in range(Bn):
        for strip in (range((Ho + R - 1) // R) if strips is None else strips):
            ho0 = strip * R
            xs = np.full(XROWS * 32, np.nan)
            in_units = min(XRS, H - 2 * ho0) * W * 8  # whole input rows inside the image; the rest is staged as zeros
            for u in range(XRS * W * 8):
                c4, pix = u & 7, u >> 3
                w, hh = pix % W, pix // W
                row = hh * W + xcol(w)
                dst = row * 32 + (((c4 >> 1) ^ ((row >> 2) & 3)) << 3) + (c4 & 1) * 4
                xs[dst: dst + 4] = x[img, ho0 * 2 + hh, w, 4 * c4: 4 * c4 + 4] if u < in_units else 0.0
            npix_here = min(R, Ho - ho0) * Wo
            for pt in range(NPT):
                red = np.zeros(8 * 32 * RED_LD)
                for wave in range(8):
                    kh, kwp = wave >> 1, wave & 1
                    acc = np.zeros((2, 64, 16))
                    for j in range(4):
                        kw = 2 * kwp + (j >> 1)
                        af = np.zeros((64, 8)); bw = np.zeros((2, 64, 8))
                        for lane in range(64):
                            li, hi = lane & 31, lane >> 5
                            p = pt * 32 + li
                            pc = p if p < NPIX else 0
                            hol, wo = pc // Wo, pc % Wo
                            row = (hol * 2 + kh) * W + wo + (kw & 1) * WH + (kw >> 1)
                            chunk = (j & 1) * 2 + hi
                            src = row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3)
                            af[lane] = xs[src: src + 8]
                            for ct in range(2):
                                ci = (j & 1) * 16 + hi * 8
                                bw[ct, lane] = wf[ct * 32 + li, kh, kw, ci: ci + 8]
                        for ct in range(2):  # operands swapped: A = filter (rows = output channels), B = input (columns = pixels)
                            mfma_32x32x16(bw[ct], af, acc[ct])
                    for lane in range(64):
                        li, hi = lane & 31, lane >> 5
                        mine = (wave * 32 + li) * RED_LD
                        for ct in range(2):
                            for g in range(4):
                                red[mine + ct * 32 + 8 * g + 4 * hi: mine + ct * 32 + 8 * g + 4 * hi + 4] = acc[ct, lane, 4 * g: 4 * g + 4]
                for t in range(512):
                    rpix, rco = t >> 4, (t & 15) * 4
                    s = sum(red[(w8 * 32 + rpix) * RED_LD + rco: (w8 * 32 + rpix) * RED_LD + rco + 4] for w8 in range(8))
                    if bias is not None:
                        s = s + bias[rco: rco + 4]
                    if relu:
                        s = np.maximum(s, 0)
                    op = pt * 32 + rpix
                    if op < npix_here:
                        y[img, ho0 + op // Wo, op % Wo, rco: rco + 4] = s
    return y


def test_conv2_forward_strip_maps():
    rng = np.random.default_rng(1)
    x = rng.integers(-3, 4, (1, 63, 63, 32)).astype(np.float64)
    wf = rng.integers(-2, 3, (64, 4, 4, 32)).astype(np.float64)
    bias = rng.integers(-5, 6, 64).astype(np.float64)
    strips = (0, 7, 14)
    y = conv2_fwd_strip_model(x, wf, bias, relu=True, strips=strips)
    ref = np.zeros((30, 30, 64))
    for kh in range(4):
        for kw in range(4):
            ref += np.einsum("hwc,nc->hwn", x[0, kh: kh + 59: 2, kw: kw + 59: 2], wf[:, kh, kw])
    ref = np.maximum(ref + bias, 0)
    for s in strips:
        assert np.array_equal(y[0, 2 * s: 2 * s + 2], ref[2 * s: 2 * s + 2]), s
    assert np.isnan(y[0, 2]).all()  # strips that were not run stay untouched


@pytest.mark.parametrize("H,W", [(20, 20), (31, 31), (21, 14), (4, 5)])
def test_conv2_forward_strip_maps_at_runtime_geometries(H, W):
    """The runtime-geometry instantiation (84^2 / 128^2 observations, non-square, the smallest input): odd Ho leaves a one-row last strip
    whose second output row reads input rows below the image -- staged as zeros, never stored."""
    rng = np.random.default_rng(H * 64 + W)
    x = rng.integers(-3, 4, (1, H, W, 32)).astype(np.float64)
    wf = rng.integers(-2, 3, (64, 4, 4, 32)).astype(np.float64)
    Ho, Wo = (H - 4) // 2 + 1, (W - 4) // 2 + 1
    strips = sorted({0, (Ho + 1) // 2 - 1})
    y = conv2_fwd_strip_model(x, wf, None, relu=False, strips=strips)
    ref = np.zeros((Ho, Wo, 64))
    for kh in range(4):
        for kw in range(4):
            ref += np.einsum("hwc,nc->hwn", x[0, kh: kh + 2 * Ho - 1: 2, kw: kw + 2 * Wo - 1: 2], wf[:, kh, kw])
    for s_ in strips:
        rows = slice(2 * s_, min(2 * s_ + 2, Ho))
        assert np.array_equal(y[0, rows], ref[rows]), s_


def test_lds_read_patterns_are_conflict_free():
    """Bank model: 64 banks x 4 bytes.  (a) conv2_fwd_strip's ds_read_b128 fragment reads: a phase = 16 lanes x 16 bytes must cover all
    64 banks once; (b) wgrad3x3's transpose reads at 64-byte pixel rows: the two 16-lane groups of a 32-lane half (same 4 pixel rows,
    both channel halves) cover 256 contiguous bytes."""
    W, WH, Wo = 63, 32, 30
    for kh in range(4):
        for kw in range(4):
            for chunk in range(4):
                for p0 in range(0, 60 - 15):  # 16 consecutive pixels of the strip, any alignment, incl. an output-row crossing
                    banks = []
                    for li in range(16):
                        p = p0 + li
                        hol, wo = p // Wo, p % Wo
                        row = (hol * 2 + kh) * W + wo + (kw & 1) * WH + (kw >> 1)
                        byte = row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)
                        banks += [((byte >> 2) + q) & 63 for q in range(4)]
                    if p0 // Wo == (p0 + 15) // Wo:  # inside one output row: consecutive LDS rows -> every bank exactly once
                        assert sorted(banks) == list(range(64)), (kh, kw, chunk, p0)
    for base_row in range(0, 40):
        banks = []
        for lane in range(32):
            i16 = lane & 15
            byte = (base_row + (i16 >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (i16 & 3) * 8
            banks += [((byte >> 2) + q) & 63 for q in range(2)]
        assert sorted(banks) == list(range(64)), base_row


# ---- conv2_dgrad_strip.h ----------------------------------------------------------------------------------------------------------
def conv2_dgrad_strip_model(dy, wd, mask, R2=2, strips=None, H=63, W=63):
    """dy [B][Ho][Wo][64], wd [32 ci][4][4][64 co], mask [B][H][W][32] or None -> dx [B][H][W][32] through the kernel's maps
    (H, W <= 63; 63 x 63 is the compile-time instantiation)."""
    Bn = dy.shape[0]
    Ho, Wo, YCOLS, RED_LD = dy.shape[1], dy.shape[2], 33, 68
    assert Ho == (H - 4) // 2 + 1 and Wo == (W - 4) // 2 + 1
    YRS = R2 + 1
    dx = np.full((Bn, H, W, 32), np.nan)
    for img in range(Bn):
        for strip in (range(((H + 1) // 2 + R2 - 1) // R2) if strips is None else strips):
            h20 = strip * R2
            ys = np.zeros(YRS * YCOLS * 64)
            for u in range(YRS * Wo * 16):
                r = u // (Wo * 16)
                rem = u - r * (Wo * 16)
                w, c4 = rem >> 4, rem & 15
                pixidx = r * YCOLS + w + 1
                dst = pixidx * 64 + (((c4 >> 1) ^ ((pixidx >> 1) & 7)) << 3) + (c4 & 1) * 4
                row = h20 - 1 + r
                ys[dst: dst + 4] = dy[img, row, w, 4 * c4: 4 * c4 + 4] if 0 <= row < Ho else 0.0
            for tr in range(R2):
                red = np.zeros(8 * 32 * RED_LD)
                for wave in range(8):
                    tap, ph = wave >> 1, wave & 1
                    ta, tb = tap >> 1, tap & 1
                    acc = np.zeros((2, 64, 16))
                    for j in range(4):
                        yf = np.zeros((64, 8)); bw = np.zeros((2, 64, 8))
                        for lane in range(64):
                            li, hi = lane & 31, lane >> 5
                            pixidx = (tr + 1 - ta) * YCOLS + li + 1 - tb
                            src = pixidx * 64 + (((2 * j + hi) ^ ((pixidx >> 1) & 7)) << 3)
                            yf[lane] = ys[src: src + 8]
                            for pw in range(2):
                                kh, kw, co = ph + 2 * ta, pw + 2 * tb, j * 16 + hi * 8
                                bw[pw, lane] = wd[li, kh, kw, co: co + 8]
                        for pw in range(2):
                            mfma_32x32x16(bw[pw], yf, acc[pw])
                    for lane in range(64):
                        li, hi = lane & 31, lane >> 5
                        mine = (wave * 32 + li) * RED_LD
                        for pw in range(2):
                            for g in range(4):
                                red[mine + pw * 32 + 8 * g + 4 * hi: mine + pw * 32 + 8 * g + 4 * hi + 4] = acc[pw, lane, 4 * g: 4 * g + 4]
                h2 = h20 + tr
                for t in range(512):
                    for q in range(2):
                        idx = t + q * 512
                        rph, cell, cq = idx >> 9, (idx >> 4) & 31, idx & 15
                        s = sum(red[((2 * t4 + rph) * 32 + cell) * RED_LD + cq * 4: ((2 * t4 + rph) * 32 + cell) * RED_LD + cq * 4 + 4]
                                for t4 in range(4))
                        h, w = 2 * h2 + rph, 2 * cell + (cq >> 3)
                        if h < H and w < W:
                            c0 = (cq & 7) * 4
                            if mask is not None:
                                s = np.where(mask[img, h, w, c0: c0 + 4] > 0, s, 0.0)
                            dx[img, h, w, c0: c0 + 4] = s
    return dx


def test_conv2_data_gradient_strip_maps():
    rng = np.random.default_rng(2)
    dy = rng.integers(-3, 4, (1, 30, 30, 64)).astype(np.float64)
    w = rng.integers(-2, 3, (64, 32, 4, 4)).astype(np.float64)  # OIHW
    wd = np.ascontiguousarray(w.transpose(1, 2, 3, 0))          # [ci][kh][kw][co]
    mask = rng.integers(-1, 2, (1, 63, 63, 32)).astype(np.float64)
    strips = (0, 9, 14, 15)  # top border, interior, the rows that read dY rows 28 / 29 / outside, the half-empty last cell row
    dx = conv2_dgrad_strip_model(dy, wd, mask, strips=strips)
    ref = np.zeros((63, 63, 32))
    for kh in range(4):
        for kw in range(4):
            ref[kh: kh + 59: 2, kw: kw + 59: 2] += np.einsum("hwn,nc->hwc", dy[0], w[:, :, kh, kw])
    ref = np.where(mask[0] > 0, ref, 0.0)
    for s in strips:
        rows = slice(4 * s, min(4 * s + 4, 63))
        assert np.array_equal(dx[0, rows], ref[rows]), s
    assert np.isnan(dx[0, 4]).all()
    # bank pattern of the fragment reads: 16 consecutive cells, any tap / chunk -> every bank once
    for base in range(0, 3 * 33 - 17):
        for chunk in range(8):
            banks = []
            for i in range(16):
                pixidx = base + i
                byte = pixidx * 128 + ((chunk ^ ((pixidx >> 1) & 7)) << 4)
                banks += [((byte >> 2) + q) & 63 for q in range(4)]
            assert sorted(banks) == list(range(64)), (base, chunk)


@pytest.mark.parametrize("H,W", [(20, 20), (31, 31), (21, 14), (4, 5)])
def test_conv2_data_gradient_strip_maps_at_runtime_geometries(H, W):
    rng = np.random.default_rng(H * 64 + W + 1)
    Ho, Wo = (H - 4) // 2 + 1, (W - 4) // 2 + 1
    dy = rng.integers(-3, 4, (1, Ho, Wo, 64)).astype(np.float64)
    w = rng.integers(-2, 3, (64, 32, 4, 4)).astype(np.float64)
    wd = np.ascontiguousarray(w.transpose(1, 2, 3, 0))
    dx = conv2_dgrad_strip_model(dy, wd, None, H=H, W=W)
    ref = np.zeros((H, W, 32))
    for kh in range(4):
        for kw in range(4):
            ref[kh: kh + 2 * Ho - 1: 2, kw: kw + 2 * Wo - 1: 2] += np.einsum("hwn,nc->hwc", dy[0], w[:, :, kh, kw])
    assert np.array_equal(dx[0], ref)


# ---- conv1x1_gn_stream.h ----------------------------------------------------------------------------------------------------------
def conv1x1_gn_stream_model(x, w, gamma, beta, groups, stride, relu, residual=None, eps=1e-5):
    """x [H][W][C] (one frame), w [Cout][C] -> y [Ho*Wo][Cout] through the kernel's maps: workgroup = slab of 32 output channels, wave =
    MT tiles of 32 pixels, activations staged per (tile, channel chunk) into a wave-private [32][CH + 4] LDS tile by units u = lane + 64 j,
    fragments = 8 consecutive channels of the lane's pixel, weights in the fragment order of cgs_split_weights, statistics by per-lane
    entries -> half-wave totals -> groups."""
    H, W, C = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    HoWo = Ho * Wo
    MT = 1 if HoWo <= 256 else (2 if HoWo <= 512 else 4)
    CH = 128 if C >= 128 else C
    NCH = C // CH
    CP, QPP, KSC = CH + 4, CH // 4, CH // 16
    NU, KS = QPP // 2, (CH // 16) * NCH
    gs = Cout // groups
    gsh = gs.bit_length() - 1
    ng = 32 >> gsh
    # fragment-ordered weight planes (cgs_split_weights_kernel): [ct][s][lane][8]
    wq = np.zeros((Cout // 32) * KS * 512)
    for co in range(Cout):
        for k in range(C):
            ct, s_, lane, e = co // 32, k // 16, ((k % 16) // 8) * 32 + co % 32, k % 8
            wq[(ct * KS + s_) * 512 + lane * 8 + e] = w[co, k]
    xf = x.reshape(H * W, C)
    y = np.full((HoWo, Cout), np.nan)
    for slab in range(Cout // 32):
        acc = np.zeros((8, MT, 64, 16))
        for wave in range(8):
            offs = np.zeros((MT, 32), dtype=np.int64)
            for m in range(MT):
                for li in range(32):
                    p = (wave * MT + m) * 32 + li
                    pc = p if p < HoWo else HoWo - 1
                    offs[m, li] = (pc // Wo * stride) * W + (pc % Wo) * stride
            for st in range(MT * NCH):
                m, kc = st // NCH, st % NCH
                xs = np.full(32 * CP, np.nan)
                for lane in range(64):
                    for j in range(NU):
                        u = lane + 64 * j
                        pix, quad = u // QPP, u % QPP
                        xs[pix * CP + quad * 4: pix * CP + quad * 4 + 4] = xf[offs[m, pix], kc * CH + quad * 4: kc * CH + quad * 4 + 4]
                for ks in range(KSC):
                    af = np.zeros((64, 8)); bw = np.zeros((64, 8))
                    for lane in range(64):
                        li, hi = lane & 31, lane >> 5
                        af[lane] = xs[li * CP + 16 * ks + 8 * hi: li * CP + 16 * ks + 8 * hi + 8]
                        base = ((slab * KS + kc * KSC + ks) * 64 + lane) * 8
                        bw[lane] = wq[base: base + 8]
                    mfma_32x32x16(bw, af, acc[wave, m])  # operands swapped: rows = output channels, columns = pixels
        # statistics: lane entries -> half-wave totals -> groups
        pairs = gs == 2
        nent = 8 if pairs else 4
        okm = np.zeros((8, MT, 64), dtype=bool)
        for wave in range(8):
            for m in range(MT):
                for lane in range(64):
                    okm[wave, m, lane] = (wave * MT + m) * 32 + (lane & 31) < HoWo

        def entry_val(vals16, j):
            if pairs:
                return vals16[4 * (j >> 1) + 2 * (j & 1)] + vals16[4 * (j >> 1) + 2 * (j & 1) + 1]
            return vals16[4 * j: 4 * j + 4].sum()

        def group_total(red, grp):
            if pairs:
                e = ((grp >> 1) & 1) * 8 + 2 * (grp >> 2) + (grp & 1)
                return sum(red[w_][e] for w_ in range(8))
            nq, q0 = gs >> 2, grp * (gs >> 2)
            return sum(red[w_][((q0 + i) & 1) * 8 + ((q0 + i) >> 1)] for w_ in range(8) for i in range(nq))

        def entry_group(h, j):
            return 4 * (j >> 1) + 2 * h + (j & 1) if pairs else (8 * j + 4 * h) >> gsh

        red = np.zeros((8, 16))
        for wave in range(8):
            for hi in range(2):
                for j in range(nent):
                    red[wave, hi * 8 + j] = sum(entry_val(acc[wave, m, 32 * hi + li], j) for m in range(MT) for li in range(32)
                                                if okm[wave, m, 32 * hi + li])
        n = HoWo * gs
        mu = np.array([group_total(red, g_) / n for g_ in range(ng)])
        red2 = np.zeros((8, 16))
        for wave in range(8):
            for hi in range(2):
                for j in range(nent):
                    t_ = 0.0
                    for m in range(MT):
                        for li in range(32):
                            if not okm[wave, m, 32 * hi + li]:
                                continue
                            v = acc[wave, m, 32 * hi + li]
                            idx = [4 * (j >> 1) + 2 * (j & 1), 4 * (j >> 1) + 2 * (j & 1) + 1] if pairs else list(range(4 * j, 4 * j + 4))
                            t_ += sum((v[i] - mu[entry_group(hi, j)]) ** 2 for i in idx)
                    red2[wave, hi * 8 + j] = t_
        rs = np.array([1.0 / np.sqrt(group_total(red2, g_) / n + eps) for g_ in range(ng)])
        for wave in range(8):
            for m in range(MT):
                for lane in range(64):
                    if not okm[wave, m, lane]:
                        continue
                    li, hi = lane & 31, lane >> 5
                    p = (wave * MT + m) * 32 + li
                    for g_ in range(4):
                        for k in range(4):
                            c0 = 8 * g_ + 4 * hi + k
                            grp = c0 >> gsh
                            sc = rs[grp] * gamma[slab * 32 + c0]
                            o = acc[wave, m, lane, 4 * g_ + k] * sc + (beta[slab * 32 + c0] - mu[grp] * sc)
                            if residual is not None:
                                o += residual[p, slab * 32 + c0]
                            y[p, slab * 32 + c0] = max(o, 0.0) if relu else o
    return y


@pytest.mark.parametrize("H,W,C,Cout,groups,stride", [(8, 8, 32, 32, 16, 1), (6, 7, 64, 64, 16, 1), (18, 18, 32, 64, 8, 1), (32, 32, 32, 32, 2, 1),
                                                       (9, 10, 256, 32, 1, 2), (5, 5, 128, 96, 24, 1)])
def test_conv1x1_gn_stream_maps(H, W, C, Cout, groups, stride):
    """Every group size the kernel takes (2 .. 32), one / two / four tiles per wave, two channel chunks (C = 256), stride 2, a ragged
    last tile: conv + GroupNorm (+ residual, ReLU) through the kernel's maps == the direct evaluation."""
    rng = np.random.default_rng(H * 1000 + C + Cout)
    x = rng.integers(-3, 4, (H, W, C)).astype(np.float64)
    w = rng.integers(-2, 3, (Cout, C)).astype(np.float64)
    gamma, beta = rng.uniform(0.5, 1.5, Cout), rng.uniform(-0.3, 0.3, Cout)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.standard_normal((Ho * Wo, Cout))
    y = conv1x1_gn_stream_model(x, w, gamma, beta, groups, stride, relu=True, residual=res)
    raw = np.einsum("hwc,nc->hwn", x[::stride, ::stride], w).reshape(Ho * Wo, Cout)
    g = raw.reshape(Ho * Wo, groups, Cout // groups)
    mean, var = g.mean(axis=(0, 2)), g.var(axis=(0, 2))
    ref = ((g - mean[None, :, None]) / np.sqrt(var[None, :, None] + 1e-5)).reshape(Ho * Wo, Cout) * gamma + beta + res
    ref = np.maximum(ref, 0.0)
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("CH", [32, 64, 128])
def test_conv1x1_gn_stream_lds_bank_pattern(CH):
    """64 banks x 4 bytes; a 16-byte access is served in phases of 16 lanes.  (a) staging writes: lanes l .. l + 15 of an instruction write
    consecutive units -- [pixel][CH + 4] rows, quads of a pixel contiguous (conflict-free except a 2-way overlap on 4 banks at CH = 32);
    (b) fragment reads (the accesses of the main loop): the 16 lanes of a phase are 16 consecutive
    pixels at one channel offset, row stride CH + 4 floats = 4 banks (mod 64) -> every bank exactly once."""
    CP, QPP = CH + 4, CH // 4
    for phase in range(4):
        banks = []
        for lane in range(16 * phase, 16 * phase + 16):
            u = lane  # j = 0; j > 0 shifts every lane by the same multiple of 64 units
            a = (u // QPP) * CP + (u % QPP) * 4
            banks += [(a + i) % 64 for i in range(4)]
        # CH = 32: a phase covers two 32-float pixel rows 36 floats apart -> 4 of the 64 banks are hit twice (a 2-way conflict on a
        # write that happens once per tile); CH = 64 / 128: one pixel row or part of one -> conflict-free
        assert max(banks.count(b) for b in set(banks)) == (2 if CH == 32 else 1), ("write", CH, phase)
        assert len(set(banks)) == (60 if CH == 32 else 64), ("write", CH, phase)
    for ks in range(CH // 16):
        for hi in range(2):
            for half in range(2):  # the two 16-byte reads of a fragment
                for phase in range(2):
                    banks = []
                    for li in range(16 * phase, 16 * phase + 16):
                        a = li * CP + 16 * ks + 8 * hi + 4 * half
                        banks += [(a + i) % 64 for i in range(4)]
                    assert len(set(banks)) == 64, ("read", CH, ks, hi, half, phase)
