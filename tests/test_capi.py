"""CPU: the C-ABI library loads without a GPU and exports exactly the symbols that include/habitat_amd.h declares;
host-only entry points (pack-info builder, version, error strings) are exercised."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from test_oracle_golden import G, canon_pack, check_pack_consistent

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "habitat_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hab_[a-z0-9_]+)\s*\(", src)))


def test_header_binding_and_library_agree():
    from habitat_amd import _lib
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 30
    assert sorted(_lib.SIGNATURES.keys()) == syms, set(syms) ^ set(_lib.SIGNATURES.keys())
    for s in syms:
        assert hasattr(L, s), f"{s} declared in habitat_amd.h but not exported by libhabitat_amd.so"
    assert L.hab_abi_version() == 1
    assert b"invalid argument" in L.hab_error_string(-1)


def test_missing_library_fails_loudly(monkeypatch):
    from habitat_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libhabitat_amd.so")
    with pytest.raises(_lib.HabError):
        _lib.lib()


def test_argument_validation_returns_error_codes():
    from habitat_amd import _lib
    L = _lib.lib()
    assert L.hab_compute_returns(None, None, None, None, None, 4, 4, 0.99, 0.95, 1, 0, None) == -1
    assert L.hab_sample_actions(None, None, None, 0, 4, 0, None) == -1
    assert L.hab_policy_create(None, None) == -1
    d = _lib.PolicyDesc(0, 18, 32, 0, 0, 1, 100, 4, 64, 64, 1, 1, 2, 16, 4)  # hidden not a multiple of 64
    h = C.c_void_p()
    assert L.hab_policy_create(C.byref(d), C.byref(h)) == -1


def test_policy_engine_parameter_table_matches_reference_state_dict():
    """Names / shapes / order of the flat arena = PointNavBaselinePolicy.state_dict() of the reference
    (fixture list in oracle.fixtures.baseline_param_shapes, itself checked against the live reference)."""
    from habitat_amd import _lib
    from oracle.fixtures import baseline_param_shapes
    L = _lib.lib()
    for (cin, H, W, hidden, rgb, depth) in ((4, 256, 256, 512, 1, 1), (1, 84, 84, 512, 0, 1), (4, 44, 44, 64, 1, 1)):
        d = _lib.PolicyDesc(0, 18, 32, 0, 0, 1, hidden, 4, H, W, rgb, depth, 2, 64, 8)
        h = C.c_void_p()
        assert L.hab_policy_create(C.byref(d), C.byref(h)) == 0
        n = L.hab_policy_num_params(h)
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd, off = C.c_int(0), C.c_int64(0)
        got, prev_end = [], 0
        for i in range(n):
            assert L.hab_policy_param_info(h, i, name, 256, shape, C.byref(nd), C.byref(off)) == 0
            shp = tuple(int(shape[k]) for k in range(nd.value))
            got.append((name.value.decode(), shp))
            assert off.value % 4 == 0 and off.value >= prev_end
            prev_end = off.value + int(np.prod(shp))
        assert got == baseline_param_shapes(cin, H, W, hidden)
        assert L.hab_policy_param_floats(h) >= prev_end
        assert L.hab_policy_work_floats(h) > 0 and L.hab_policy_packed_floats(h) > 0
        L.hab_policy_destroy(h)


def test_pack_info_cpp_builder_vs_reference_golden():
    from habitat_amd.engine import DevicePackInfo
    z = np.load(os.path.join(G, "pack_info.npz"))
    for i in range(int(z["num_cases"])):
        dones = z[f"c{i}_dones"]
        T, N = dones.shape
        ref = {k[len(f"c{i}_"):]: z[k] for k in z.files if k.startswith(f"c{i}_") and k != f"c{i}_dones"}
        mine = DevicePackInfo(dones).arrays
        check_pack_consistent(mine, T, N)
        assert canon_pack(ref, N) == canon_pack(mine, N)
        assert np.array_equal(np.asarray(mine["sequence_lengths"]), np.asarray(ref["sequence_lengths"]))
        assert np.array_equal(np.asarray(mine["num_seqs_at_step"]), np.asarray(ref["num_seqs_at_step"]))


@pytest.mark.parametrize("T,N", [(1, 1), (2, 1), (128, 32), (7, 64)])
def test_pack_info_edge_cases(T, N):
    from habitat_amd.engine import DevicePackInfo
    from oracle import functional as O
    for dones in (np.zeros((T, N), bool), np.ones((T, N), bool), np.random.default_rng(T + N).random((T, N)) < 0.3):
        mine = DevicePackInfo(dones).arrays
        check_pack_consistent(mine, T, N)
        assert canon_pack(O.build_pack_info_from_dones(dones), N) == canon_pack(mine, N)
