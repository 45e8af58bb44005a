"""CPU: the arithmetic identities behind three round-4 fusions, restated in numpy (the kernels themselves are checked on the GPU against
float64 and the reference goldens; this file makes the formulas reviewable without one).

  * RunningMeanAndVar moments inside the ingest pass (csrc/resnet_ops.hip `ingest_pool_kernel<true>`, `moment_finish_mean/var`): per channel
    S1 = sum (x - p), S2 = sum (x - p)^2 about a pivot p, fp32 over runs of 32 values flushed into doubles; mean = p + S1 / n and the
    variance about ANY mean m -- under DD-PPO the cross-rank mean (running_mean_and_var.py:38-49) -- is (S2 - 2 (m - p) S1 + n (m - p)^2) / n.
  * GroupNorm statistics from per-strip partials (csrc/stem_conv_strip.h epilogue -> `gn_chunk_apply_pool_kernel`): (mean_k, M2_k) of
    strips of 8 output rows merged by Chan's formula == the two-pass statistics of the whole frame, for a ragged last strip too.
  * the layer wavefront of the packed recurrence (csrc/rnn.hip `rnn_seq_wave_forward` / `_backward`): launch w runs step w - l of layer l
    (forward) / step max_len - 1 - (w - (L - 1 - l)) (backward); every cell's inputs were written by an EARLIER launch."""
import numpy as np
import pytest


def pivot_moments(x, p):
    """x: [n] float32 of one channel; the kernel's accumulation: d = x - p in fp32, fp32 sums over runs of 32, doubles across runs."""
    s1 = s2 = 0.0
    for i in range(0, len(x), 32):
        d = (x[i: i + 32] - np.float32(p)).astype(np.float32)
        f1 = np.float32(0)
        f2 = np.float32(0)
        for v in d:
            f1 = np.float32(f1 + v)
            f2 = np.float32(np.float32(v * v) + f2)  # (the kernel uses an fma: one rounding less)
        s1 += float(f1)
        s2 += float(f2)
    return s1, s2


@pytest.mark.parametrize("scale,pivot", [(1.0, 0.0), (1.0, 0.43), (900.0, 0.0), (900.0, 300.0)])
def test_pivot_form_moments_give_mean_and_variance_about_any_mean(scale, pivot):
    rng = np.random.default_rng(int(scale) + int(pivot * 100))
    x = (rng.random(8192) ** 2 * scale).astype(np.float32)  # rgb / depth in [0, 1]; semantic ids up to ~1e3 (pivot 0 on the first update)
    n = len(x)
    s1, s2 = pivot_moments(x, pivot)
    mean = pivot + s1 / n
    x64 = x.astype(np.float64)
    assert abs(mean - x64.mean()) <= 2e-7 * max(1.0, abs(x64.mean()))
    for m in (np.float32(mean), np.float32(mean * 1.01 + 0.003)):  # this rank's mean; a cross-rank mean that differs from it
        d = float(m) - pivot
        var = (s2 - 2.0 * d * s1 + n * d * d) / n
        ref = ((x64 - float(m)) ** 2).mean()
        assert abs(var - ref) <= 2e-6 * ref, (m, var, ref)


@pytest.mark.parametrize("Ho,Wo,groups", [(64, 64, 16), (12, 33, 8), (9, 5, 32), (8, 64, 16)])
def test_strip_partials_merge_to_the_frame_statistics(Ho, Wo, groups):
    rng = np.random.default_rng(Ho * Wo)
    C = 32
    cpg = C // groups
    y = rng.standard_normal((Ho, Wo, C)) * rng.uniform(0.1, 3.0, C) + rng.uniform(-2, 2, C)
    nstrips = (Ho + 7) // 8
    part = np.zeros((nstrips, groups, 2))
    for k in range(nstrips):  # the convolution's epilogue: exact local two-pass per strip and group
        blk = y[8 * k: 8 * k + 8].reshape(-1, groups, cpg)
        mean_k = blk.mean(axis=(0, 2))
        part[k, :, 0] = mean_k
        part[k, :, 1] = ((blk - mean_k[None, :, None]) ** 2).sum(axis=(0, 2))
    # gn_chunk_apply_pool_kernel's merge: chunk k holds min(chunk4, F4 - k chunk4) float4s = whole pixels
    C4 = C // 4
    F4, chunk4 = Ho * Wo * C4, 8 * Wo * C4
    n = Ho * Wo * cpg
    mean = np.zeros(groups)
    for k in range(nstrips):
        nk = (min(chunk4, F4 - k * chunk4) // C4) * cpg
        mean += nk * part[k, :, 0]
    mean /= n
    m2 = np.zeros(groups)
    for k in range(nstrips):
        nk = (min(chunk4, F4 - k * chunk4) // C4) * cpg
        m2 += part[k, :, 1] + nk * (part[k, :, 0] - mean) ** 2
    g = y.reshape(-1, groups, cpg)
    assert np.allclose(mean, g.mean(axis=(0, 2)), rtol=0, atol=1e-12)
    assert np.allclose(m2 / n, g.var(axis=(0, 2)), rtol=1e-12)


@pytest.mark.parametrize("L,max_len", [(2, 1), (2, 7), (3, 5), (4, 2)])
def test_layer_wavefront_schedule_respects_the_dependencies(L, max_len):
    # forward: cell (l, s) needs (l, s - 1) [its own state] and (l - 1, s) [the layer below's output of the same step]
    done = {}
    for w in range(max_len + L - 1):
        cells = [(l, w - l) for l in range(L) if 0 <= w - l < max_len]
        for l, s in cells:
            assert s == 0 or done[(l, s - 1)] < w
            assert l == 0 or done[(l - 1, s)] < w
        for c in cells:
            assert c not in done
            done[c] = w
    assert len(done) == L * max_len
    # BPTT: cell (l, s) needs (l, s + 1) [the carry] and (l + 1, s) [the layer above's dgi of the same step]
    done = {}
    for w in range(max_len + L - 1):
        cells = [(l, max_len - 1 - (w - (L - 1 - l))) for l in range(L) if 0 <= max_len - 1 - (w - (L - 1 - l)) < max_len]
        for l, s in cells:
            assert s == max_len - 1 or done[(l, s + 1)] < w
            assert l == L - 1 or done[(l + 1, s)] < w
        for c in cells:
            assert c not in done
            done[c] = w
    assert len(done) == L * max_len
