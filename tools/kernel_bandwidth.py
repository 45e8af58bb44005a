#!/usr/bin/env python3
"""Joins a round's per-kernel times (profiles/<tag>_<w>_kernel_stats.txt, incl. its per-grid table) with the counter traffic of the same
kernels (profiles/<tag>_<w>_hbm_traffic.json, per kernel and grid): achieved HBM TB/s per kernel, largest total time first -- the list of
streaming kernels that run far below the memory rate.   usage: python tools/kernel_bandwidth.py [tag] [c2|c3|c5 ...]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
for w in (sys.argv[2:] or ["c2", "c3", "c5"]):
    tr = json.load(open(f"{ROOT}/profiles/{tag}_{w}_hbm_traffic.json"))
    main, bygrid, sec = {}, {}, 0
    for line in open(f"{ROOT}/profiles/{tag}_{w}_kernel_stats.txt").read().splitlines():
        if line.startswith("# kernels launched at several"):
            sec = 1
            continue
        if line.startswith("#") or not line.strip() or line.strip().startswith("total_ms"):
            continue
        p = line.split(None, 6)
        try:
            if sec == 0:
                main[p[6].strip()] = (float(p[0]), int(p[2]), float(p[3]))
            else:
                bygrid[(p[6].strip(), int(p[5]))] = (float(p[0]), int(p[1]), float(p[2]))
        except (ValueError, IndexError):
            pass
    rows = []
    for name, v in tr.items():
        m = re.match(r"(.*) \[grid (\d+)\]$", name)
        kn, grid = (m.group(1), int(m.group(2))) if m else (name, None)
        multi = any(k[0] == kn[:130] for k in bygrid)   # listed per grid: the main table's average mixes its sizes
        t = bygrid.get((kn[:130], grid)) if multi else main.get(kn[:150])
        mb = (v["fetch_bytes_per_call"] + v["write_bytes_per_call"]) / 1e6
        if t and mb >= 20:
            rows.append((t[0], t[2], mb, kn[:90], grid))
    rows.sort(reverse=True)
    print(f"== {w}: total ms in the trace, avg us per call, counter MB per call, achieved TB/s")
    for tot, avg, mb, kn, grid in rows[:24]:
        print(f"{tot:9.1f} {avg:9.1f} {mb:9.1f} {mb / avg:6.2f}  {kn} [grid {grid}]")
