"""A/B on the GPU: the device-path rollout as plain stream launches vs ONE captured hipGraph of the whole rollout (T steps x launches per step),
same library, same box, same state ("MEASURE hipGraph capture").

  python tools/rollout_graph_ab.py [c2|c3|c5 ...] > gpurun_out/r06_rollout_graph_ab.txt

Per workload: 2 warm-up update cycles; then, from one snapshot of (rollout arena, env source state, episode statistics):
  plain   : collect_rollout() x R, HIP events around each
  graph   : the same T steps captured once (torch.cuda.graph on the engine's launch stream), replayed x R with fresh noise copied into the
            captured buffer, HIP events around each replay
and the arena after ONE plain rollout vs after ONE replay from the same snapshot with the same noise, compared bit for bit.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
import bench  # noqa: E402


def leaves(td, prefix=""):
    for k, v in td.items():
        if isinstance(v, torch.Tensor):
            yield prefix + k, v
        else:
            yield from leaves(v, prefix + k + ".")


def snapshot(trainer):
    st = trainer._agent.rollouts
    snap = {"buf": {k: v.clone() for k, v in leaves(st.buffers)}, "t": trainer.envs._t.clone(), "since": trainer.envs._since.clone(),
            "cer": trainer.current_episode_reward.clone(), "stats": {k: v.clone() for k, v in trainer.running_episode_stats.items()},
            "idx": list(st.current_rollout_step_idxs)}
    return snap


def restore(trainer, snap):
    st = trainer._agent.rollouts
    for k, v in leaves(st.buffers):
        v.copy_(snap["buf"][k])
    trainer.envs._t.copy_(snap["t"])
    trainer.envs._since.copy_(snap["since"])
    trainer.current_episode_reward.copy_(snap["cer"])
    for k, v in trainer.running_episode_stats.items():
        v.copy_(snap["stats"][k])
    st.current_rollout_step_idxs = list(snap["idx"])


def run(workload, reps=5):
    trainer, cfg = bench.make_trainer(workload, 8)
    trainer._init_train()
    for _ in range(2):
        trainer.run_update_cycle()
    torch.cuda.synchronize()
    T = trainer._ppo_cfg.num_steps
    st = trainer._agent.rollouts
    trainer._agent.eval()
    snap = snapshot(trainer)
    torch.manual_seed(1234)
    noise0 = trainer._draw_rollout_noise(T).clone()

    def plain(noise):
        for step in range(T):
            trainer._device_rollout_step(step, noise)

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        h1 = time.perf_counter()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1), (h1 - h0) * 1e3

    # reference result: one plain rollout from the snapshot
    restore(trainer, snap)
    plain(noise0)
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in leaves(st.buffers)}
    ref_t = trainer.envs._t.clone()
    # plain timing
    t_plain = []
    for _ in range(reps):
        restore(trainer, snap)
        t_plain.append(timed(lambda: plain(noise0)))
    # capture
    static_noise = noise0.clone()
    restore(trainer, snap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    c0 = time.perf_counter()
    try:
        with torch.cuda.graph(g):
            plain(static_noise)
    except Exception as exc:  # noqa: BLE001
        print(f"[{workload}] capture FAILED: {exc!r}")
        trainer.envs.close()
        return
    torch.cuda.synchronize()
    capture_s = time.perf_counter() - c0
    # one replay from the snapshot with the same noise: bit-identical arena?
    restore(trainer, snap)
    static_noise.copy_(noise0)
    g.replay()
    torch.cuda.synchronize()
    diffs = [k for k, v in leaves(st.buffers) if not torch.equal(v, ref[k])]
    same_env = torch.equal(trainer.envs._t, ref_t)
    t_graph = []
    for _ in range(reps):
        restore(trainer, snap)
        t_graph.append(timed(g.replay))
    launches = None
    try:
        import re
        dbg = g.debug_dump  # noqa: F841  (node count is not exposed by torch; counted from the per-step figure below instead)
    except Exception:  # noqa: BLE001
        pass
    mp = sum(a for a, _ in t_plain[1:]) / (reps - 1)
    mg = sum(a for a, _ in t_graph[1:]) / (reps - 1)
    hp = sum(b for _, b in t_plain[1:]) / (reps - 1)
    hg = sum(b for _, b in t_graph[1:]) / (reps - 1)
    print(f"[{workload}] T={T} envs={trainer.envs.num_envs}  plain: {mp:.2f} ms GPU ({hp:.2f} ms host enqueue)   graph replay: {mg:.2f} ms GPU "
          f"({hg:.2f} ms host)   ratio {mg / mp:.3f}   capture+instantiate {capture_s:.2f} s   "
          f"arena bit-identical: {not diffs and same_env}{'' if not diffs else ' DIFF ' + ','.join(diffs[:6])}")
    print(f"[{workload}]   plain runs (GPU ms): {[round(a, 2) for a, _ in t_plain]}   graph runs: {[round(a, 2) for a, _ in t_graph]}")
    trainer.envs.close()
    del g, trainer
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["c2", "c3"]):
        run(w)
