#!/usr/bin/env python3
"""Splits a rocprofv3 --kernel-trace (rocpd sqlite .db) of bench.py into rollout steps and learner phases and reports, for each,
wall time, GPU-busy time (union of kernel intervals) and launches: the evidence for "launch-bound" vs "kernel-bound".

Windows end with the action-head kernel (heads_fwd_kernel / gauss_heads_fwd_kernel).  A window is a ROLLOUT STEP when it holds the synthetic
environments' step kernels (synth_images / rollout_step_stats: action of step t sampled, environments stepped, policy of step t + 1) and no
ppo_loss; every other window is learner time: [GAE + first minibatch forward], [loss + backward + Adam + next minibatch forward] ..., and the
last one [loss + backward + Adam of the last minibatch + the first policy call of the next rollout].  (Up to round 4's first collection the
window [GAE + first minibatch forward] was counted into the rollout run: "87.5 ms" there is rollout + one 15 ms minibatch forward.)
usage: tools/trace_phases.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


ENV_STEP = re.compile(r"synth_images_kernel|rollout_step_stats_kernel")


def busy(iv):
    iv.sort()
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    windows, cur = [], []
    for name, s, e in rows:
        cur.append((name, s, e))
        if "heads_fwd_kernel" in name:
            windows.append(cur)
            cur = []
    tail = cur
    roll, learn = [], []
    for w in windows:
        is_step = any(ENV_STEP.search(n) for n, _, _ in w) and not any("ppo_loss" in n for n, _, _ in w)
        (roll if is_step else learn).append(w)
    out = [f"# {sys.argv[1]}: {len(rows)} dispatches, {len(roll)} rollout steps, {len(learn)} windows containing learner work"]

    def stats(ws, label):
        if not ws:
            return
        wall = [w[-1][2] - w[0][1] for w in ws]
        gaps = []
        for a, b in zip(ws[:-1], ws[1:]):
            gaps.append(b[0][1] - a[-1][2])
        b_ = [busy([(s, e) for _, s, e in w]) for w in ws]
        n = [len(w) for w in ws]
        med = lambda v: sorted(v)[len(v) // 2]
        out.append(f"{label}: windows {len(ws)}  launches/window median {med(n)}  first-start->last-end median {med(wall) / 1e3:.1f} us  "
                   f"GPU-busy median {med(b_) / 1e3:.1f} us  busy/wall {sum(b_) / max(1, sum(wall)):.3f}")
        return sum(wall), sum(b_)

    stats(roll, "rollout step")
    if len(roll) > 2:
        # rollout steps are consecutive between learner windows: whole-rollout wall = from first start to last end of each run
        runs, run = [], [roll[0]]
        idx = {id(w): i for i, w in enumerate(windows)}
        for w in roll[1:]:
            if idx[id(w)] == idx[id(run[-1])] + 1:
                run.append(w)
            else:
                runs.append(run); run = [w]
        runs.append(run)
        for r in runs:
            wall = r[-1][-1][2] - r[0][0][1]
            b_ = busy([(s, e) for w in r for _, s, e in w])
            out.append(f"  rollout run of {len(r)} steps: wall {wall / 1e6:.2f} ms, GPU busy {b_ / 1e6:.2f} ms ({b_ / wall:.3f}), "
                       f"{sum(len(w) for w in r)} launches, {wall / len(r) / 1e3:.1f} us/step")
    for w in learn:
        wall = w[-1][2] - w[0][1]
        b_ = busy([(s, e) for _, s, e in w])
        out.append(f"  learner window: wall {wall / 1e6:.2f} ms, GPU busy {b_ / 1e6:.2f} ms ({b_ / wall:.3f}), {len(w)} launches")
    # per-kernel table of the rollout steps
    agg = {}
    for w in roll:
        for nme, s, e in w:
            a = agg.setdefault(re.sub(r"\bhab::", "", nme)[:110], [0, 0])
            a[0] += 1; a[1] += e - s
    tot = sum(a[1] for a in agg.values()) or 1
    out.append("# kernels inside rollout steps: total_ms share calls avg_us name")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        out.append(f"{t / 1e6:9.3f} {100 * t / tot:5.1f}% {c:7d} {t / c / 1e3:8.1f}  {k}")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
