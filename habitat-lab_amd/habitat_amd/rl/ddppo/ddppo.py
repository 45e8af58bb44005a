"""DD-PPO updater: decentralised synchronous gradient averaging (habitat_baselines/rl/ddppo/algo/ddppo.py:59-157).

The reference wraps `evaluate_actions` in DistributedDataParallel so that backward all-reduces ~60-170
parameter tensors in 25 MiB buckets, overlapped with the rest of backward.  Here every gradient already lives in
ONE flat fp32 arena, backward fills it from the end, and the overlap needs two to four messages: the engine reports
(`hab_policy_set_grad_ready`) each time a longer TAIL of the arena is final -- first visual fc + recurrent encoder + heads
(99.5 % of the SimpleCNN policy's bytes, 63 % of ResNet18's), BEFORE the convolution stack's backward starts; for the ResNets
again when layer4 (+ compression: 28 % of ResNet18) and when layer3 (7 %) are done.  Each new segment is all-reduced
asynchronously on RCCL's stream while the earlier stages' backward runs, and the small head of the arena (layer2, layer1,
stem: 2 %) follows after backward.
The 1/world_size is folded into the fused clip+Adam kernel (`grad_scale`); the initial weight broadcast (DDP ctor)
is one broadcast of the parameter arena.  HAB_NO_GRAD_OVERLAP=1 restores the single blocking all-reduce.

Device-side exchange (round 4, csrc/comm.hip): on the `nccl` backend the library opens its own RCCL communicator (unique id broadcast
once through the process group), and the engine enqueues the all-reduce of every finished gradient tail on that communicator's stream
from INSIDE backward, and the two RunningMeanAndVar sums inside the training forward -- no ctypes -> Python -> torch.distributed round
trip in the middle of a pass (what DistributedDataParallel's C++ reducer gives the reference).  Opt-in (HAB_NATIVE_COMM=1) until it
has run on more than one GPU: `negotiate_native_comm` opens it on every rank or on none (availability, creation under a watchdog and a
two-stream self-test against torch.distributed each end in a MIN vote), otherwise the callback form below carries the exchange and a
warning says why.  gloo (the CPU / shared-GPU tests) always uses the callbacks."""
from __future__ import annotations

import os

import torch
import torch.distributed as distrib

from habitat_amd.common.baseline_registry import baseline_registry
from habitat_amd.rl.ppo.ppo import PPO


def _vote(ok: bool, device) -> bool:
    """True only if EVERY rank says ok (MIN over the process group): all ranks take the same branch."""
    t = torch.tensor([1.0 if ok else 0.0], device=device if distrib.get_backend() == "nccl" else "cpu")
    distrib.all_reduce(t, op=distrib.ReduceOp.MIN)
    return float(t.item()) == 1.0


def _run_with_timeout(fn, timeout_s: float):
    """(finished, result, exception) of fn() run on a daemon thread: a call that never returns (ncclCommInitRank waiting for a rank that
    failed before it) is abandoned -- its thread and whatever it holds are leaked on purpose, the caller goes on without it."""
    import threading
    box = {}

    def body():
        try:
            box["result"] = fn()
        except BaseException as exc:  # noqa: BLE001
            box["exc"] = exc

    th = threading.Thread(target=body, daemon=True, name="hab-native-comm")
    th.start()
    th.join(timeout_s)
    return (not th.is_alive()), box.get("result"), box.get("exc")


def negotiate_native_comm(world: int, rank: int, device, *, available=None, make_id=None, create=None, selftest=None, vote=None,
                          bcast=None, timeout_s: float = None):
    """Opens the library-owned RCCL communicator (csrc/comm.hip) on every rank or on none.  Each stage that can fail or hang on a
    SUBSET of the ranks ends in a MIN vote over the process group, so a rank never sits in ncclCommInitRank / an all-reduce waiting
    for a peer that has already fallen back:

      1. librccl symbols found?                                                            -> vote
      2. rank 0 creates the unique id; (ok, id) is broadcast: a failure there reaches every rank as data, not as an exception
      3. ncclCommInitRank on a watchdog thread (HAB_NATIVE_COMM_TIMEOUT, default 60 s)       -> vote (a rank still inside votes no and
         abandons the call; ranks that got a communicator destroy it)
      4. self-test on a SIDE stream, polled with a deadline: one all-reduce against torch.distributed bit for bit, then the real
         pattern -- communicator-stream and compute-stream all-reduces of one communicator alternating with torch's own RCCL
         all-reduces for HAB_NATIVE_COMM_SELFTEST (default 64) rounds                        -> vote

    Returns (communicator or None, reason).  The keyword hooks exist for the CPU test of the protocol (gloo, injected failures)."""
    from habitat_amd import _lib
    from habitat_amd.engine import NativeComm
    timeout_s = float(os.environ.get("HAB_NATIVE_COMM_TIMEOUT", "60")) if timeout_s is None else timeout_s
    vote = vote or (lambda ok: _vote(ok, device))
    bcast = bcast or (lambda obj: distrib.broadcast_object_list(obj, src=0))
    available = available or (lambda: bool(_lib.lib().hab_comm_available()))
    make_id = make_id or NativeComm.unique_id
    def _create_on_device(ident):
        # the watchdog thread starts with HIP's per-thread default device (0): bind the rank's GPU before ncclCommInitRank and the
        # communicator's stream are created there (every rank of a torchrun job sees all GPUs)
        if getattr(device, "type", None) == "cuda":
            torch.cuda.set_device(device)
        return NativeComm(world, rank, ident=ident)

    create = create or _create_on_device
    selftest = selftest or (lambda comm: _native_comm_selftest(comm, device, timeout_s))
    try:
        have = bool(available())
    except Exception:  # noqa: BLE001
        have = False
    if not vote(have):
        return None, "librccl is not loadable on every rank"
    ident, err0 = [None], ""
    if rank == 0:
        try:
            ident[0] = make_id()
        except Exception as exc:  # noqa: BLE001
            ident[0] = None
            err0 = repr(exc)
    bcast(ident)
    if ident[0] is None:
        return None, "rank 0 could not create the unique id" + (f" ({err0})" if err0 else "")
    finished, comm, exc = _run_with_timeout(lambda: create(ident[0]), timeout_s)
    mine = finished and exc is None and comm is not None
    if not vote(mine):
        if mine:
            comm.close()
        return None, ("ncclCommInitRank did not return within the deadline here" if not finished else
                      f"communicator creation failed here ({exc!r})" if exc is not None else "communicator creation failed on another rank")
    try:
        ok, detail = selftest(comm)
    except Exception as exc:  # noqa: BLE001
        ok, detail = False, repr(exc)
    if not vote(ok):
        if ok:
            comm.close()
        # (a communicator whose self-test hung or failed here is abandoned, not destroyed: ncclCommDestroy may wait for the stuck work)
        return None, f"self-test: {detail if not ok else 'failed on another rank'}"
    return comm, "ok"


def _native_comm_selftest(comm, device, timeout_s: float):
    """See negotiate_native_comm stage 4.  Everything the communicator does runs on side streams and is awaited by POLLING an event
    against a deadline, so a hung collective costs a timeout, not the process."""
    import time
    rounds = int(os.environ.get("HAB_NATIVE_COMM_SELFTEST", "64"))
    rank = distrib.get_rank()
    base = torch.arange(1, 257, device=device, dtype=torch.float32)
    side_a, side_b = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    torch.cuda.current_stream(device).synchronize()
    refs, gots_a, gots_b = [], [], []
    for r in range(rounds + 1):  # round 0: the single all-reduce; then the alternating pattern
        x = base * float((rank + 1) * (r + 1))  # integers in fp32: the sum is exact, any order
        ref = x.clone()
        distrib.all_reduce(ref)  # torch's own RCCL communicator, its own stream
        refs.append(ref)
        a, b = x.clone(), x.clone()
        side_a.wait_stream(torch.cuda.current_stream(device))
        side_b.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side_a):   # the gradient tails' role
            comm.all_reduce_sum_(a)
        with torch.cuda.stream(side_b):   # the RunningMeanAndVar sums' role (same communicator, another stream)
            side_b.wait_stream(side_a)    # one communicator: its collectives must be ordered the same way on every rank
            comm.all_reduce_sum_(b)
        gots_a.append(a)
        gots_b.append(b)
    done = torch.cuda.Event()
    with torch.cuda.stream(side_b):
        done.record()
    deadline = time.monotonic() + timeout_s
    while not done.query():
        if time.monotonic() > deadline:
            return False, f"the communicator's all-reduces did not finish within {timeout_s:.0f} s"
        time.sleep(0.001)
    torch.cuda.current_stream(device).synchronize()
    for r, (ref, a, b) in enumerate(zip(refs, gots_a, gots_b)):
        if not (torch.equal(ref, a) and torch.equal(ref, b)):
            return False, f"round {r}: sums differ from torch.distributed's"
    return True, f"{rounds + 1} rounds bit-identical to torch.distributed"


class DecentralizedDistributedMixin:
    def _world_size(self) -> int:
        return distrib.get_world_size() if distrib.is_initialized() else 1

    def init_distributed(self, find_unused_params: bool = True) -> None:
        """Broadcast rank 0's parameters (what the DDP constructor does in the reference, ddppo.py:110-140)."""
        assert distrib.is_initialized(), "Distributed must be initialized"
        eng = self.actor_critic.engine
        distrib.broadcast(eng.params_flat, src=0)
        eng.repack()
        # every tensor of the policy OUTSIDE the arena too (auxiliary-loss modules: each rank seeded them differently,
        # ppo_trainer.py:208-211) -- the DDP constructor broadcasts all of actor_critic's parameters and buffers
        for t in DecentralizedDistributedMixin._tensors_outside_arena(self.actor_critic):
            distrib.broadcast(t.data, src=0)
        world = distrib.get_world_size()
        self._native_comm = None
        if DecentralizedDistributedMixin._want_native_comm(world):
            self._native_comm = DecentralizedDistributedMixin._open_native_comm(eng, world)
        if self._native_comm is not None:
            eng.set_comm(self._native_comm)  # gradient tails + RunningMeanAndVar sums are enqueued by the engine itself
        elif world > 1:
            # RunningMeanAndVar batch moments + frame count are summed over ranks inside the engine's forward
            # (rl/ddppo/policy/running_mean_and_var.py:38-41,47-49); the engine divides the means / variances by world_size
            # where it consumes them and merges with the summed count
            def _avg(view: torch.Tensor, scale: float) -> None:
                distrib.all_reduce(view)
                if scale != 1.0:
                    view.mul_(scale)

            eng.set_allreduce(_avg, world)
        self._grad_works = []      # handles of the early all-reduces of this backward
        self._grad_first = None    # grads_flat[_grad_first:] is covered by them
        overlap = (world > 1 and os.environ.get("HAB_NO_GRAD_OVERLAP") is None) or os.environ.get("HAB_FORCE_GRAD_OVERLAP") is not None
        if self._native_comm is None and overlap and hasattr(eng, "set_grad_ready"):
            def _tail_ready(first: int, count: int) -> None:
                g = eng.grads_flat
                assert first + count == g.numel()
                end = g.numel() if self._grad_first is None else self._grad_first
                assert first < end, "the engine reports growing tails"
                self._grad_works.append(distrib.all_reduce(g[first:end], async_op=True))
                self._grad_first = first

            eng.set_grad_ready(_tail_ready)

        def _dense_sync() -> None:  # autograd bridge (PPO._evaluate_actions + loss.backward()): average, as DDP does in backward
            self._all_reduce_grads()
            if world > 1:
                eng.grads_flat.div_(world)

        self.actor_critic._dense_grad_sync = _dense_sync
        self._distributed = True

    @staticmethod
    def _tensors_outside_arena(actor_critic):
        """Parameters and buffers of actor_critic that are not views of the engine's arena (aux_loss_modules, foreign modules hung on the
        policy), in a rank-independent order (registration order)."""
        if not hasattr(actor_critic, "named_parameters"):
            return []
        eng = actor_critic.engine
        lo = eng.params_flat.data_ptr()
        hi = lo + eng.params_flat.numel() * eng.params_flat.element_size()
        out = []
        for _, t in list(actor_critic.named_parameters()) + list(actor_critic.named_buffers()):
            if t is None or t.numel() == 0:
                continue
            if t.device == eng.params_flat.device and lo <= t.data_ptr() < hi:
                continue
            out.append(t)
        return out

    @staticmethod
    def _want_native_comm(world: int) -> bool:
        """OPT-IN (HAB_NATIVE_COMM=1) until a multi-GPU run of the bit-identity test against the callback form has passed: the pattern
        -- one library-owned communicator driven from two streams beside torch's own RCCL process group -- has only ever run with one
        rank.  RCCL process groups only (a gloo group means CPU tensors or several ranks on one GPU, which RCCL refuses)."""
        if os.environ.get("HAB_NATIVE_COMM") != "1" or distrib.get_backend() != "nccl" or not torch.cuda.is_available():
            return False
        return True

    @staticmethod
    def _open_native_comm(eng, world: int):
        from habitat_amd.utils.logging import logger
        comm, why = negotiate_native_comm(world, distrib.get_rank(), eng.params_flat.device)
        if comm is None:
            logger.warning(f"device-side DD-PPO exchange not used ({why}): the torch.distributed callbacks carry the exchange")
        return comm

    def _all_reduce_grads(self) -> None:
        if not distrib.is_initialized():
            return
        if getattr(self, "_native_comm", None) is not None:
            self.actor_critic.engine.grad_sync()  # head of the arena + wait for the tails the engine enqueued during backward
            return
        g = self.actor_critic.engine.grads_flat
        works, first = getattr(self, "_grad_works", []), getattr(self, "_grad_first", None)
        self._grad_works, self._grad_first = [], None
        if first is None:
            if distrib.get_world_size() > 1:
                distrib.all_reduce(g)
            return
        if first > 0:
            distrib.all_reduce(g[:first])  # the head of the arena, produced after the last reported tail
        for w in works:
            w.wait()                       # the current stream now waits for the early all-reduces

    def _all_reduce_scalar_stats(self, t: torch.Tensor) -> None:
        if distrib.is_initialized() and distrib.get_world_size() > 1:
            distrib.all_reduce(t)


@baseline_registry.register_updater
class DDPPO(DecentralizedDistributedMixin, PPO):
    pass
