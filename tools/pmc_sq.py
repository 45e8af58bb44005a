#!/usr/bin/env python3
"""Per-kernel SQ counter table from rocprofv3 --pmc csv passes (counters only, one pass per <= 8 SQ counters).
usage: tools/pmc_sq.py <dir containing *_counter_collection.csv> [out.txt] [kernel-substring ...]
Derived columns (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES and
SQ_BUSY_CYCLES count cycles):  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES or per-SE SQ_BUSY_CYCLES scaled)."""
import collections
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    subs = sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = re.sub(r"\bhab::", "", r["Kernel_Name"])
                if subs and not any(s in k for s in subs):
                    continue
                a = agg[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    lines = []
    for k, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 0])[1]):
        m = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
        n = max(v[0] for v in cs.values())
        lines.append(f"== {k[:150]}   ({n} launches; per-launch means)")
        for c in sorted(m):
            lines.append(f"   {c:32s} {m[c]:16.0f}")
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if c in m:
                    lines.append(f"   {c + ' / SQ_WAVE_CYCLES':42s} {m[c] / wc:8.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"]:
            lines.append(f"   {'MFMA busy = MFMA_BUSY_CYCLES / BUSY_CU_CYCLES / 4 SIMDs':42s} {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CU_CYCLES'] / 4:8.3f}")
        if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"]:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"):
                if c in m:
                    lines.append(f"   {c + ' per wave':42s} {m[c] / m['SQ_WAVES']:10.1f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
