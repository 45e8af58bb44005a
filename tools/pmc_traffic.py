#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output), as prescribed by
/opt/skills/guides/MI355X_MICROARCH.md (separate passes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads -> doubled here; both counters are reported in KB).
usage: tools/pmc_traffic.py <dir with FETCH_SIZE_counter_collection.csv, WRITE_SIZE_counter_collection.csv> out.json out.txt"""
import collections
import csv
import json
import re
import sys


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != name:
                continue
            # one row per (kernel, grid): a kernel that serves the rollout (64 frames) and the update (512 .. 2048 frames) is not averaged
            # over the two
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
            a = agg[re.sub(r"\bhab::", "", r["Kernel_Name"]) + (f" [grid {grid}]" if grid else "")]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    d, out_json, out_txt = sys.argv[1:4]
    fe = load(f"{d}/FETCH_SIZE_counter_collection.csv", "FETCH_SIZE")
    wr = load(f"{d}/WRITE_SIZE_counter_collection.csv", "WRITE_SIZE")
    rows = {}
    for k, (n, v) in fe.items():
        w = wr.get(k, [n, 0.0])
        rows[k] = dict(calls=n, fetch_bytes_per_call=2.0 * v / n * 1024.0, write_bytes_per_call=w[1] / max(w[0], 1) * 1024.0)
    json.dump(rows, open(out_json, "w"), indent=1)
    lines = ["# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 gfx950 correction)",
             f"{'fetch MB':>10} {'write MB':>10} {'calls':>7}  kernel"]
    for k, r in sorted(rows.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_call"] + kv[1]["write_bytes_per_call"]) * kv[1]["calls"]):
        lines.append(f"{r['fetch_bytes_per_call'] / 1e6:10.1f} {r['write_bytes_per_call'] / 1e6:10.1f} {r['calls']:7d}  {k[:140]}")
    open(out_txt, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
