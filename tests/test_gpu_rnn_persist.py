"""GPU: the persistent time-major recurrence (csrc/rnn_persist.h: ONE launch per layer and time chunk, state exchanged between the
workgroups of a row tile through write-through stores + an agent-scope arrival counter) against the step-per-launch form it replaces
(csrc/rnn.hip, selected by matrix-path bit 12): the two run the same MFMA sequence and the same summation trees, so every output of the
engine's evaluate -- values, log-probs, entropy, final hidden state -- and every gradient must be EQUAL bit for bit.  The oracle parity of
either form then is the other's (tests/test_gpu_policy.py::test_engine_lstm_gru_multilayer_vs_oracle runs through the persistent form at
hidden 512 / 256 / 128; the golden updates at hidden 512).

Cases: GRU and LSTM, 1 and 2 layers, hidden 128 / 256 / 512, one row tile with fewer than 16 environments, several row tiles with a ragged
last one, chunk lengths that do and do not divide T (HAB_RNN_CHUNKS, default 4, is read once per process), episode starts in every step,
all-zero masks (every frame starts an episode)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOAL = "pointgoal_with_gps_compass"
STEP_LAUNCHES = 4096  # matrix-path bit 12


def run_engine(eng, inputs, T, n, hidden, Lh):
    from habitat_amd.engine import DevicePackInfo
    rgb, depth, goal, masks, actions, h0, gv, glp, gent = inputs
    B = T * n
    pack = DevicePackInfo(np.logical_not(masks.view(T, n).cpu().numpy()), "cuda")
    dv, dl, de = (torch.zeros(B, device="cuda") for _ in range(3))
    rows = torch.arange(B, dtype=torch.int32, device="cuda")  # (frame f at arena row f: the rows-indirected call is what takes the time-major form)
    eng.evaluate(rgb, depth, goal, rows, h0, masks, actions, pack, B, n, value=dv, log_prob=dl, entropy=de)
    hf = torch.zeros(n, Lh, hidden, device="cuda")
    eng.final_hidden(hf)
    eng.backward(rgb, depth, goal, rows, actions, pack, gv, glp, gent)
    torch.cuda.synchronize()
    return (dv.clone(), dl.clone(), de.clone(), hf.clone()), {k: g.detach().clone() for k, g in eng.grad_views.items()}


@pytest.mark.parametrize("rnn_type,layers,hidden,T,n,p_start", [
    ("GRU", 1, 512, 32, 16, 0.06),    # the C2 minibatch geometry: one full row tile, 4 chunks of 8 steps
    ("GRU", 1, 512, 13, 5, 0.3),      # fewer than 16 environments, chunks of 4 + 4 + 4 + 1 steps
    ("GRU", 1, 128, 9, 37, 0.2),      # three row tiles, ragged last one
    ("LSTM", 1, 256, 10, 20, 0.25),
    ("LSTM", 2, 512, 8, 16, 0.15),    # two layers, layer by layer
    ("GRU", 2, 256, 6, 3, 1.0),       # every frame starts an episode
    ("LSTM", 1, 512, 6, 240, 0.1),    # 15 row tiles x 32 = 480 workgroups of a kernel that fits once per CU: row tiles run in rounds
])
def test_persistent_recurrence_bit_identical_to_step_launches(rnn_type, layers, hidden, T, n, p_start):
    from habitat_amd import _lib
    from habitat_amd.engine import PolicyEngine
    from oracle.fixtures import baseline_param_shapes, det_params
    L = _lib.lib()
    H = W = 44
    B = T * n
    params = det_params(baseline_param_shapes(4, H, W, hidden, rnn_type=rnn_type, layers=layers), 5)
    rng = np.random.default_rng(T * 100 + n)
    Lh = layers * (2 if rnn_type == "LSTM" else 1)
    inputs = (torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda(),
              torch.from_numpy(rng.random((B, H, W, 1), dtype=np.float32)).cuda(),
              torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32)).cuda(),
              torch.from_numpy(rng.random((B, 1)) >= p_start).cuda(),
              torch.from_numpy(rng.integers(0, 4, (B, 1))).cuda(),
              torch.from_numpy(rng.standard_normal((n, Lh, hidden)).astype(np.float32)).cuda(),
              *(torch.from_numpy(rng.standard_normal(B).astype(np.float32)).cuda() for _ in range(3)))
    results = {}
    prev = L.hab_set_matrix_path(-1)
    try:
        for name, mode in (("persistent", prev & ~STEP_LAUNCHES), ("steps", prev | STEP_LAUNCHES), ("persistent_again", prev & ~STEP_LAUNCHES)):
            L.hab_set_matrix_path(mode)
            eng = PolicyEngine(arch="simple_cnn", rnn_type=rnn_type, rnn_layers=layers, hidden=hidden, H=H, W=W, max_frames=B, max_envs=n)
            eng.load({k: v.cuda() for k, v in params.items()})
            results[name] = run_engine(eng, inputs, T, n, hidden, Lh)
            del eng
    finally:
        L.hab_set_matrix_path(prev)
    ref_out, ref_g = results["steps"]
    assert all(torch.isfinite(x).all() for x in ref_out) and float(ref_g["net.state_encoder.rnn.weight_ih_l0"].abs().max()) > 0
    for name in ("persistent", "persistent_again"):
        out, grads = results[name]
        for a, b, what in zip(out, ref_out, ("value", "log_prob", "entropy", "final_hidden")):
            assert torch.equal(a, b), (name, what, float((a - b).abs().max()))
        bad = [k for k in ref_g if not torch.equal(grads[k], ref_g[k])]
        assert not bad, (name, bad[:8])
