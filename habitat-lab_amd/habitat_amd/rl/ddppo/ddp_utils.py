"""Distributed helpers with the surface of habitat_baselines/rl/ddppo/ddp_utils.py
(get_distrib_size :247-264, init_distrib_slurm :271-309, rank0_only :100-138, resume-state files :74-87,182-224,
preemption signal flags :141-179).  One process per GPU; backend "nccl" is RCCL over xGMI on ROCm."""
from __future__ import annotations

import contextlib
import functools
import os
import signal
import threading
import time
from typing import Any, Callable, Optional, Tuple

import torch
import torch.distributed as distrib

EXIT = threading.Event()
EXIT.clear()
REQUEUE = threading.Event()
REQUEUE.clear()
SAVE_STATE = threading.Event()
SAVE_STATE.clear()

DEFAULT_PORT = 8738
DEFAULT_PORT_RANGE = 127
DEFAULT_MAIN_ADDR = "127.0.0.1"
SLURM_JOBID = os.environ.get("SLURM_JOB_ID", None)
RESUME_STATE_BASE_NAME = ".habitat-resume-state"


def is_slurm_job() -> bool:
    return SLURM_JOBID is not None


def is_slurm_batch_job() -> bool:
    return is_slurm_job() and os.environ.get("SLURM_JOB_NAME", None) not in (None, "bash")


def _clean_exit_handler(signum, frame):
    EXIT.set()


def _requeue_handler(signum, frame):
    EXIT.set()
    REQUEUE.set()


def _save_state_handler(signum, frame):
    SAVE_STATE.set()


def add_signal_handlers() -> None:
    signal.signal(signal.SIGINT, _clean_exit_handler)
    signal.signal(signal.SIGTERM, _clean_exit_handler)
    signal.signal(signal.SIGUSR2, _requeue_handler)
    signal.signal(signal.SIGUSR1, _save_state_handler)


def resume_state_filename(config, filename_key: str = "") -> str:
    fname = RESUME_STATE_BASE_NAME
    if is_slurm_job() and config.habitat_baselines.rl.preemption.append_slurm_job_id:
        fname += f"-{SLURM_JOBID}"
    return os.path.join(config.habitat_baselines.checkpoint_folder, fname + filename_key + ".pth")


def save_resume_state(state: Any, filename_or_config, filename_key: str = ""):
    fn = filename_or_config if isinstance(filename_or_config, str) else resume_state_filename(filename_or_config, filename_key)
    os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
    torch.save(state, fn)


def load_resume_state(filename_or_config, filename_key: str = "") -> Optional[Any]:
    fn = filename_or_config if isinstance(filename_or_config, str) else resume_state_filename(filename_or_config, filename_key)
    if not os.path.exists(fn):
        return None
    return torch.load(fn, map_location="cpu", weights_only=False)


def requeue_job():
    """SLURM requeue (ddp_utils.py:227-240); outside SLURM it only synchronises the ranks."""
    if not REQUEUE.is_set():
        return
    if distrib.is_initialized():
        distrib.barrier()
    if SLURM_JOBID is not None and (not distrib.is_initialized() or distrib.get_rank() == 0):
        import shlex
        import subprocess
        subprocess.check_call(shlex.split(f"scontrol requeue {SLURM_JOBID}"))


def get_distrib_size() -> Tuple[int, int, int]:
    """(local_rank, world_rank, world_size) from torchrun (LOCAL_RANK/RANK/WORLD_SIZE) or SLURM variables."""
    if os.environ.get("LOCAL_RANK", None) is not None:
        return int(os.environ["LOCAL_RANK"]), int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("SLURM_JOBID", None) is not None:
        return int(os.environ["SLURM_LOCALID"]), int(os.environ["SLURM_PROCID"]), int(os.environ["SLURM_NTASKS"])
    return 0, 0, 1


def get_main_addr() -> str:
    return os.environ.get("MAIN_ADDR", os.environ.get("MASTER_ADDR", DEFAULT_MAIN_ADDR))


def init_distrib_slurm(backend: str = "nccl"):
    """Rendezvous + process group (ddp_utils.py:271-309).  Returns (local_rank, store); the store also carries the
    `num_done` counter of the preemptive straggler synchronisation (ppo_trainer.py:641-653).
      * under torchrun (MASTER_ADDR / MASTER_PORT set, no MAIN_PORT): the launcher's own store is used
        (`env://`), no extra port is opened;
      * otherwise (SLURM / MAIN_ADDR + MAIN_PORT, as in the reference): an explicit TCPStore hosted by rank 0."""
    assert distrib.is_available(), "torch.distributed must be available"
    local_rank, world_rank, world_size = get_distrib_size()
    global _STORE
    if distrib.is_initialized():  # a second trainer in this process (bench.py's sub-records at N > 1): same group, same store
        from torch.distributed import distributed_c10d
        return local_rank, (_STORE if _STORE is not None else distributed_c10d._get_default_store())
    if "MASTER_PORT" in os.environ and "MAIN_PORT" not in os.environ and SLURM_JOBID is None:
        os.environ.setdefault("MASTER_ADDR", DEFAULT_MAIN_ADDR)
        distrib.init_process_group(backend.lower(), init_method="env://", rank=world_rank, world_size=world_size)
        from torch.distributed import distributed_c10d
        return local_rank, distributed_c10d._get_default_store()
    main_port = int(os.environ.get("MAIN_PORT", DEFAULT_PORT))
    if SLURM_JOBID is not None:
        main_port += int(SLURM_JOBID) % int(os.environ.get("MAIN_PORT_RANGE", DEFAULT_PORT_RANGE))
    main_addr = get_main_addr()
    tcp_store = distrib.TCPStore(main_addr, main_port, world_size, world_rank == 0)
    distrib.init_process_group(backend.lower(), store=tcp_store, rank=world_rank, world_size=world_size)
    _STORE = tcp_store
    return local_rank, tcp_store


_STORE = None  # the explicit TCPStore of the SLURM-style rendezvous (kept for a second trainer of the same process)


def rank_cpu_block(cpus, local_rank: int, local_world: int):
    """The CPUs of local rank r of w: the r-th of w contiguous blocks of the sorted CPU list (contiguous ranges share a NUMA node and,
    on SMT hosts, lie on one side of the sibling split).  Falls back to the whole list when there are fewer CPUs than ranks."""
    cpus = sorted(cpus)
    if local_world <= 1 or len(cpus) < local_world:
        return cpus
    per = len(cpus) // local_world
    return cpus[local_rank * per:(local_rank + 1) * per]


_AFFINITY_CALLS = [0]  # per-process count of mask exchanges: every rank makes the same calls, so call k reads call k's keys


def _local_sharers(mask, store=None):
    """World ranks ON THIS HOST whose inherited CPU mask is identical to `mask` (this rank included), ascending -- or None when the ranks
    cannot be asked.  The masks travel through the rendezvous STORE (plain key / value traffic with rank 0's server), not through a
    collective: this runs right after init_process_group, BEFORE the rank has selected its GPU, and an object collective on the nccl
    backend would stage its payload on the current device -- cuda:0 on every rank.  Without a store: a gloo group may use
    all_gather_object; an nccl group without a store returns None (the caller falls back to the whole-host test)."""
    if not (distrib.is_available() and distrib.is_initialized()):
        return None
    import json
    import socket
    mine = [socket.gethostname(), list(mask)]
    world, rank = distrib.get_world_size(), distrib.get_rank()
    if store is not None:
        k = _AFFINITY_CALLS[0]
        _AFFINITY_CALLS[0] += 1
        store.set(f"hab_affinity/{k}/{rank}", json.dumps(mine))
        everyone = [json.loads(bytes(store.get(f"hab_affinity/{k}/{r}")).decode()) for r in range(world)]  # get() waits for the key
    elif distrib.get_backend() == "gloo":
        everyone = [None] * world
        distrib.all_gather_object(everyone, mine)
    else:
        return None
    return [r for r, other in enumerate(everyone) if other == mine]


def pin_rank_affinity(local_rank: int, local_world: Optional[int] = None, store=None):
    """One process per GPU, each enqueueing ~7 500 launches per update cycle: without pinning, 8 ranks' host threads migrate across the
    sockets of the node and share cores with each other's environment workers.  Restricts this process (and the threads / workers it
    starts afterwards) to its block of the CPUs it may run on -- but ONLY when the mask it inherited is SHARED with other local ranks
    (torchrun on a whole node).  A launcher that already confined every task to its own CPUs -- SLURM task affinity / cgroups: the
    reference's README launch gives each of 4 tasks --cpus-per-task 10 -- is left alone: slicing those 10 CPUs by 4 again would leave
    each rank, and the environment workers that inherit its mask, 2 of them.  Sharing is established by comparing the masks of the
    ranks on this host through the rendezvous `store` (see _local_sharers; without any way to ask: the mask covers every CPU of the host).
    HAB_NO_AFFINITY=1 leaves the affinity alone.  Returns the chosen CPUs, or None when nothing was changed."""
    if os.environ.get("HAB_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None
    mask = sorted(os.sched_getaffinity(0))
    sharers = _local_sharers(mask, store)
    if sharers is not None:
        if len(sharers) <= 1:
            return None  # nobody else on this host runs on these CPUs: they are this rank's already
        me = distrib.get_rank()
        block = rank_cpu_block(mask, sharers.index(me), len(sharers))
    else:
        if len(mask) < (os.cpu_count() or 0):
            return None  # already confined by the launcher
        if local_world is None:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("SLURM_NTASKS_PER_NODE", "1")) or 1)
        block = rank_cpu_block(mask, local_rank, local_world)
    if not block or len(block) == len(mask):
        return None
    try:
        os.sched_setaffinity(0, block)
    except OSError:
        return None
    return block


def rank0_only(fn: Optional[Callable] = None):
    """Predicate (`rank0_only()`) and decorator (`@rank0_only`) -- ddp_utils.py:100-138."""
    if fn is None:
        return (not distrib.is_available()) or (not distrib.is_initialized()) or distrib.get_rank() == 0

    @functools.wraps(fn)
    def _wrapper(*args, **kwargs):
        if rank0_only():
            return fn(*args, **kwargs)
        return None

    return _wrapper


class StoreCounterPoller:
    """Cached view of an integer key of the rendezvous store, refreshed by a daemon thread while `polling()` is active.

    The reference reads the `num_done` counter with one blocking `store.get` per rollout step (ppo_trainer.py:641-653); its steps
    take milliseconds of simulator time.  Here a rollout step is ~200 us of GPU work enqueued by the host slightly ahead of the
    device, and a TCP round trip per step (tens of microseconds, 75 % of the steps, every rank against rank 0's server) would put the
    host on the critical path of multi-rank rollouts.  While `polling()` is active the trainer reads the cached value instead:
    at most `interval_s` (0.5 ms, ~2 rollout steps) older than a direct query.  Outside `polling()` every read is a direct query."""

    def __init__(self, store, key: str, interval_s: float = 0.0005):
        self._store, self._key, self._interval = store, key, interval_s
        self._active = threading.Event()
        self._fresh = threading.Event()   # set once the cached value was read after the current activation
        self._stop = threading.Event()    # stop(): the thread leaves its loop and is joined (teardown before the store goes away)
        self._lock = threading.Lock()     # guards (_gen, _fresh, _value): an activation and a straddling query never interleave
        self._value = 0
        self._gen = 0
        self._error: Optional[BaseException] = None
        self._thread: Optional[threading.Thread] = None

    def _run(self) -> None:
        while not self._stop.is_set():
            self._active.wait()
            if self._stop.is_set():
                return
            try:
                with self._lock:
                    gen = self._gen
                value = int(self._store.get(self._key))
                with self._lock:
                    if gen == self._gen:  # a query that straddles a new activation does not count as fresh for it
                        self._value = value
                        self._fresh.set()
            except BaseException as e:  # store gone (teardown) or transport error: fall back to direct queries
                self._error = e
                self._active.clear()
                self._fresh.set()
                return
            self._stop.wait(self._interval)

    def stop(self, timeout_s: float = 2.0) -> None:
        """Ends the polling thread and joins it.  Call before the process group / store is destroyed: a daemon thread left inside a
        blocking TCPStore call at interpreter shutdown can hang or crash the teardown."""
        self._stop.set()
        self._active.set()  # wake the thread if it is parked between activations
        t = self._thread
        if t is not None and t.is_alive() and t is not threading.current_thread():
            t.join(timeout_s)
        self._active.clear()

    def read(self) -> int:
        if self._active.is_set() and self._error is None:
            self._fresh.wait()
            if self._error is None:
                return self._value
        return int(self._store.get(self._key))

    @contextlib.contextmanager
    def polling(self):
        if self._error is None and not self._stop.is_set() and (self._thread is None or not self._thread.is_alive()):
            self._thread = threading.Thread(target=self._run, name="num_done_poller", daemon=True)
            self._thread.start()
        if self._stop.is_set():  # stopped pollers answer every read with a direct query
            yield self
            return
        with self._lock:
            self._gen += 1
            self._fresh.clear()
        self._active.set()
        try:
            yield self
        finally:
            self._active.clear()
