"""Not collected by `pytest tests/`: the engine-vs-oracle parity test of tests/test_gpu_policy.py at the odd / non-square observation
sizes of the reference's test/test_baseline_resnet.py.  On the CPU these geometries are pinned for the parameter table and the oracle
(tests/test_host_logic.py, tests/test_oracle_golden.py); the HIP engine has only been run at even sizes.  First GPU call of the next
round:  python -m pytest tools/experiments/test_gpu_odd_geometry.py -q   -- then move the cases that pass into RESNET_VARIANTS."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))

import test_gpu_policy as tgp  # noqa: E402

ODD = [
    ("resnet18", "GRU", 1, 62, 30, ("rgb", "depth"), True),
    ("resnet18", "LSTM", 2, 63, 84, ("rgb", "depth"), False),
    ("resnet50", "GRU", 1, 65, 30, ("rgb", "depth"), True),
    ("resnet18", "GRU", 1, 66, 64, ("depth",), False),
    ("resnet18", "GRU", 1, 100, 180, ("rgb", "depth"), True),
]


@pytest.mark.parametrize("backbone,rnn_type,layers,H,W,keys,normalize", ODD)
def test_resnet_engine_vs_oracle_odd_geometry(backbone, rnn_type, layers, H, W, keys, normalize):
    tgp.test_resnet_engine_vs_oracle(backbone, rnn_type, layers, H, W, keys, normalize)
