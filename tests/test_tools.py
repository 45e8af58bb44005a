"""CPU checks of the measurement tooling: the roofline model bench.py reports, the rollout / learner phase split of a kernel trace and the
SQ counter table (profiles/ are produced by these)."""
import csv
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_site_roofline_picks_the_bound_that_takes_longer():
    import bench
    # conv1 forward: 966 784 B and 65.0 MFLOP per frame -> 0.121 us at 8 TB/s vs 0.098 us at the 666.7 TFLOP/s-eq MFMA ceiling: HBM
    fl = bench.PROBES["conv1_fwd"][1]
    r = bench.site_roofline("conv1_fwd", fl, 2048, 1.0)  # 2048 frames in 1 ms
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.PEAK_HBM_GBS
    assert abs(r["achieved"] - 2048 * 966784 / 1e-3 / 1e9) < 1.0
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert abs(r["fp32_equiv_tflops"] - 2048 * fl / 1e-3 / 1e12) < 0.1
    assert abs(r["mfma_ceiling_fp32_equiv_tflops"] - 2500.0 / 3.75) < 0.1
    # conv2 forward: 738 432 B (0.092 us) vs 59.0 MFLOP at 416.7 TFLOP/s-eq (0.142 us): MFMA
    r2 = bench.site_roofline("conv2_fwd", bench.PROBES["conv2_fwd"][1], 2048, 1.0)
    assert r2["bound"] == "mfma" and r2["unit"] == "TFLOP/s" and abs(r2["peak"] - 416.7) < 0.1
    assert abs(r2["frac"] - r2["achieved"] / r2["peak"]) < 1e-3
    # conv2 data gradient: written to read the fp32 activation for its sign (1 246 464 B -> HBM-bound, 0.156 us); the computation needs
    # a 1-bit mask (754 308 B -> 0.094 us, below the 0.142 us MFMA floor): both positions are on the record
    r4 = bench.site_roofline("conv2_dgrad", bench.PROBES["conv2_dgrad"][1], 512, 0.2)
    assert r4["bound"] == "hbm" and r4["algorithmic_bytes_per_frame"] == 1246464
    assert r4["algorithmic_bytes_needed"] == 30 * 30 * 64 * 4 + 63 * 63 * 32 * 4 + 63 * 63 * 4 and r4["bound_with_bytes_needed"] == "mfma"
    assert abs(r4["frac_of_bytes_needed"] - (bench.PROBES["conv2_dgrad"][1] / (2500e12 / 6)) / (0.2e-3 / 512)) < 1e-3
    assert r4["frac_of_bytes_needed"] < r4["frac"]
    assert "algorithmic_bytes_needed" not in r2
    # a call site without a model (recurrent steps) is priced against the fp32 MFMA peak
    r3 = bench.site_roofline("rnn_fwd", bench.PROBES["rnn_fwd"][1], 2048, 1.0)
    assert r3["peak"] == bench.PEAK_FP32_MFMA_TFLOPS and r3["bound"] == "mfma"


def test_committed_traffic_lookup_matches_the_probed_kernels():
    import bench
    for site in ("conv1_wgrad", "conv3_fwd", "conv3_dgrad", "fc_wgrad"):
        traffic, src = bench.hbm_traffic("c2", site)
        assert traffic and src.startswith("profiles/r0"), (site, traffic, src)
    # kernels that first appeared in round 3 (conv1 forward patch kernel, strip-resident weight gradients): only a counter pass that
    # ran them can describe them -- never an older file's entry for the kernel they replaced
    for site in ("conv1_fwd", "conv2_fwd", "conv2_dgrad", "conv2_wgrad", "conv3_wgrad"):
        traffic, src = bench.hbm_traffic("c2", site)
        assert traffic is None or not src.startswith(("profiles/r02_", "profiles/r01_")), (site, traffic, src)
    assert bench.hbm_traffic("c3", "conv1_fwd") == (None, None)


def test_trace_phases_splits_rollout_and_learner(tmp_path):
    db = tmp_path / "t_results.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer)")
    t = 0
    rows = []
    for step in range(3):  # three rollout steps: environments stepped, conv, rnn, heads
        for nm, d in (("hab::synth_images_kernel(unsigned char*)", 7000), ("void hab::igemm_bf3_kernel<ConvFwdProb>", 20000),
                      ("void hab::rnn_step_kernel<3, 8>", 6000), ("hab::heads_fwd_kernel(HeadsArgs)", 10000)):
            rows.append((nm, t, t + d)); t += d + 1000
    # learner: [GAE + first minibatch forward] (no environment kernels, no loss yet), then [loss + backward + next forward]
    for nm, d in (("hab::gae_kernel(GaeArgs)", 5000), ("void hab::igemm_bf3_kernel<ConvFwdProb>", 700000), ("hab::heads_fwd_kernel(HeadsArgs)", 10000),
                  ("hab::ppo_loss_kernel(LossArgs)", 5000), ("void hab::igemm_bf3_kernel<ConvWgradProb>", 900000), ("hab::heads_fwd_kernel(HeadsArgs)", 10000)):
        rows.append((nm, t, t + d)); t += d + 1000
    con.executemany("insert into kernels values (?, ?, ?)", rows)
    con.commit(); con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_phases.py"), str(db)], capture_output=True, text=True, check=True).stdout
    assert "3 rollout steps, 2 windows containing learner work" in out
    assert "rollout run of 3 steps" in out and "learner window" in out
    assert "launches/window median 4" in out


def test_pmc_sq_table(tmp_path):
    d = tmp_path / "p1"
    d.mkdir()
    with open(d / "p1_counter_collection.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for _ in range(2):
            for c, v in (("SQ_WAVES", 1024), ("SQ_WAVE_CYCLES", 4.0e6), ("SQ_BUSY_CU_CYCLES", 1.0e6), ("SQ_VALU_MFMA_BUSY_CYCLES", 1.2e6),
                         ("SQ_WAIT_ANY", 1.0e6), ("SQ_INSTS_VALU", 2048000), ("SQ_INSTS_MFMA", 102400)):
                w.writerow(["void hab::igemm_bf3_kernel<ConvFwdProb, 1, 2, 4, 1>(...)", c, v])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_sq.py"), str(tmp_path)], capture_output=True, text=True, check=True).stdout
    assert "igemm_bf3_kernel<ConvFwdProb, 1, 2, 4, 1>" in out and "(2 launches" in out
    assert "0.300" in out  # MFMA busy = 1.2e6 / 1.0e6 / 4
    assert "SQ_WAIT_ANY / SQ_WAVE_CYCLES" in out and "0.250" in out
    assert "SQ_INSTS_VALU per wave" in out and "2000.0" in out


def test_staged_wave_specialised_schedule_model():
    """tools/ws_schedule_model.py: the barrier schedule of the producer / consumer kernels (csrc/igemm_bf3_ws.h, obs_conv_bf3_ws.h) is replayed on the
    CPU for every k-tile count: equal barrier counts in both roles, no image written while read, register sets / key buffers hold
    what their readers expect."""
    import runpy
    mod = runpy.run_path(os.path.join(ROOT, "tools", "ws_schedule_model.py"))
    for ksh in (False, True):
        for ntk in range(0, 24):
            mod["check"](ntk, ksh)
    for n in range(1, 8):
        mod["check_obs"](n)


def test_no_kernel_outside_the_allow_list_spills_registers():
    """tools/kernel_resources.py on the built library: the code objects' metadata must show no scratch memory and no spilled VGPRs for any
    kernel except `ppo_loss_kernel` (one 1024-thread workgroup per minibatch, 15 us) -- a spill in a contraction or recurrent kernel is a
    silent 2-10x on that kernel.  Also pins the register budget that lets the recurrent backward steps share a CU with the strip kernels."""
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "habitat-lab_amd", "habitat_amd", "libhabitat_amd.so")
    if not (os.path.exists(lib) and os.path.exists("/opt/rocm/lib/llvm/bin/clang-offload-bundler")):
        pytest.skip("library not built or LLVM tools absent")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), lib], capture_output=True, text=True, check=True).stdout
    rows = [ln.split() for ln in out.splitlines() if ln and not ln.startswith("#")]
    assert len(rows) > 250
    col = lambda r, i: int(r[i - 8])  # the eight numeric columns are the last eight fields (kernel names contain spaces)
    spilling = sorted({r[0] for r in rows if col(r, 4) or col(r, 5)})
    assert spilling == ["ppo_loss_kernel"], spilling
    by = {" ".join(r[:-8]): r for r in rows}
    for k, r in by.items():
        if k.startswith("rnn_bwd_step"):
            assert col(r, 0) <= 64, (k, r)  # 8 waves per SIMD
        if k.startswith("conv2_fwd_strip_kernel") or k.startswith("conv2_dgrad_strip_kernel"):
            assert col(r, 0) <= 256, (k, r)  # 2 waves per SIMD at 512-thread workgroups
