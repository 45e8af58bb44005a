#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace result (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel table:
calls, total ms, avg us, share.  usage: tools/rocprof_summary.py <results.db|kernel_trace.csv> [out.txt]"""
import csv
import re
import sqlite3
import sys


def rows_from_db(path):
    con = sqlite3.connect(path)
    return con.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()


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
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
