#!/usr/bin/env python3
"""One learner minibatch of a rocprofv3 --kernel-trace (rocpd sqlite) of bench.py as a timeline: every kernel with start offset, duration and
queue, long kernels by name, short ones folded into runs -- who is on the critical path, where the streams wait for each other.
usage: tools/minibatch_timeline.py <results.db> [which ppo_loss window, default 6] [out.txt]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = con.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    idx = [i for i, r in enumerate(rows) if "ppo_loss" in r[0]]
    if len(idx) <= which + 1:
        print("not enough minibatches in the trace", len(idx))
        return
    a, b = idx[which], idx[which + 1]
    t0 = rows[a][1]
    out = [f"# minibatch window between ppo_loss #{which} and #{which + 1}: {(rows[b][1] - t0) / 1e3:.1f} us, {b - a} kernels"]
    qs = sorted({r[3] for r in rows[a:b]}) if qcol else [0]
    out.append(f"# queues: {qs}")
    run = None
    def flush():
        nonlocal run
        if run:
            out.append(f"{run['s'] / 1e3:9.1f} us  +{(run['e'] - run['s']) / 1e3:8.1f}  q{run['q']}  {run['n']:4d} x {run['name'][:90]}  (busy {run['busy'] / 1e3:.1f} us)")
            run = None
    for name, s, e, *q in rows[a:b]:
        q = q[0] if q else 0
        short = name.split("(")[0][:90]
        if run and run["name"] == short and run["q"] == q:
            run["e"] = e - t0; run["n"] += 1; run["busy"] += e - s
        else:
            flush()
            run = {"name": short, "q": q, "s": s - t0, "e": e - t0, "n": 1, "busy": e - s}
    flush()
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
