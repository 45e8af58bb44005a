#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace result (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel table:
calls, total ms, avg us, share -- and, for kernels launched at more than one grid size (rollout-size and update-size calls of the same
kernel), a second table per (kernel, grid): the first table's avg mixes them (NOTEBOOK R6.12).
usage: tools/rocprof_summary.py <results.db|kernel_trace.csv> [out.txt]"""
import csv
import re
import sqlite3
import sys


def rows_from_db(path):
    con = sqlite3.connect(path)
    return con.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()


def grid_rows_from_db(path):
    """(name, grid, calls, total, min, max) when the kernels view carries a grid column; [] otherwise."""
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    gcol = next((c for c in cols if re.fullmatch(r"grid(_size)?(_x)?", c, re.I)), None)
    if gcol is None:
        return []
    return con.execute(f"select name, {gcol}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name, {gcol}").fetchall()


def grid_rows_from_csv(path):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            g = r.get("Grid_Size") or r.get("Grid_Size_X")
            if g is None:
                return []
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = agg.setdefault((r["Kernel_Name"], int(g)), [0, 0, 1 << 62, 0])
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    return [(k[0], k[1], *v) for k, v in agg.items()]


def rows_from_csv(path):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    return [(k, *v) for k, v in agg.items()]


def main():
    src = sys.argv[1]
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    out = [f"# rocprofv3 --kernel-trace summary of {src}", f"# total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches",
           f"{'total_ms':>10} {'share':>6} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel"]
    for name, n, t, mn, mx in rows:
        name = re.sub(r"\bhab::", "", name)
        out.append(f"{t / 1e6:10.3f} {100.0 * t / tot:5.1f}% {n:7d} {t / n / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f}  {name[:150]}")
    try:
        grows = grid_rows_from_db(src) if src.endswith(".db") else grid_rows_from_csv(src)
    except Exception:  # noqa: BLE001 -- an unknown schema only costs the second table
        grows = []
    multi = {}
    for name, grid, n, t, mn, mx in grows:
        multi.setdefault(name, []).append((grid, n, t, mn, mx))
    multi = {k: v for k, v in multi.items() if len(v) > 1}
    if multi:
        out.append("")
        out.append("# kernels launched at several grid sizes (threads), per size")
        out.append(f"{'total_ms':>10} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'grid':>10}  kernel")
        for name in sorted(multi, key=lambda k: -sum(v[2] for v in multi[k])):
            for grid, n, t, mn, mx in sorted(multi[name], key=lambda v: -v[2])[:6]:
                short = re.sub(r"\bhab::", "", name)[:130]
                out.append(f"{t / 1e6:10.3f} {n:7d} {t / n / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {int(grid):10d}  {short}")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
