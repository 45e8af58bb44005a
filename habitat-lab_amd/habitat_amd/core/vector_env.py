"""Process-per-environment VectorEnv with a shared-memory observation plane (N1).

API of the reference's `habitat.core.vector_env.VectorEnv` (habitat-lab/habitat/core/vector_env.py:135-620): one worker process
per environment driven by (command, data) messages -- step / reset / call / count_episodes / close -- with `auto_reset_done`,
`async_step_at` / `wait_step_at`, `pause_at` / `resume_all`, `call` / `call_at` and the `num_envs`, `observation_spaces`,
`action_spaces`, `orig_action_spaces`, `number_of_episodes` attributes the trainer reads.

What differs is the data plane.  The reference pickles every observation dict through the worker's pipe
(habitat/utils/pickle5_multiprocessing.py:51-79: 458 752 B per env-step at 256x256 RGB-D), rebuilds a list of dicts in the
trainer and re-stacks it sensor by sensor in `batch_obs` (habitat_baselines/utils/common.py:244-310) before the upload.  Here the
array sensors never travel through a pipe: after the first handshake the parent allocates ONE slab per sensor, shaped
`(num_envs, *sensor_shape)` -- i.e. already the layout of a rollout-storage row -- in POSIX shared memory, and every worker
writes its observation straight into its own row.  The pipe carries only `(reward, done, info)`.  The slab is page-locked
(`hipHostRegister` through torch's runtime handle) so `batched_obs(env_slice, device)` is one asynchronous H2D copy per sensor
into the rollout row, with no per-env Python work and no intermediate stack.  A HIP event fences the slots: the next
`async_step_at` of an env waits until the upload that read its row has finished.

Sensors that are not fixed-shape arrays (none on the PPO path) still go through the pipe, so any env works unchanged."""
from __future__ import annotations

import multiprocessing as mp
import signal
from multiprocessing import shared_memory
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

STEP_COMMAND, RESET_COMMAND, CLOSE_COMMAND, CALL_COMMAND, COUNT_EPISODES_COMMAND, ATTACH_COMMAND = (
    "step", "reset", "close", "call", "count_episodes", "attach_shm")


def _slab_specs(observation_space) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """Sensors that can live in a slab: anything with a static shape and a numeric dtype."""
    specs = {}
    for k, sp in observation_space.spaces.items():
        shape, dtype = getattr(sp, "shape", None), getattr(sp, "dtype", None)
        if shape is None or dtype is None:
            continue
        dt = np.dtype(dtype)
        if dt.kind in "uifb":
            specs[k] = (tuple(int(s) for s in shape), dt.str)
    return specs


def _worker(conn, parent_conn, env_fn: Callable, env_fn_args: Tuple, auto_reset_done: bool, mask_signals: bool) -> None:
    if mask_signals:
        for s in (signal.SIGINT, signal.SIGTERM, signal.SIGUSR1, signal.SIGUSR2):
            signal.signal(s, signal.SIG_IGN)
    if parent_conn is not None:
        parent_conn.close()
    env = env_fn(*env_fn_args)
    slabs: Dict[str, np.ndarray] = {}
    shms: List[shared_memory.SharedMemory] = []
    row = -1
    episodes_done = 0

    def publish(obs):
        """Array sensors -> this env's slab row; returns what still has to travel through the pipe."""
        if not slabs:
            return obs
        rest = {}
        for k, v in obs.items():
            dst = slabs.get(k)
            if dst is None:
                rest[k] = v
            else:
                np.copyto(dst[row], np.asarray(v).reshape(dst.shape[1:]), casting="same_kind")
        return rest

    try:
        while True:
            command, data = conn.recv()
            if command == STEP_COMMAND:
                obs, reward, done, info = env.step(data)
                if done:
                    episodes_done += 1
                    if auto_reset_done:
                        obs = env.reset()
                conn.send((publish(obs), reward, done, info))
            elif command == RESET_COMMAND:
                conn.send(publish(env.reset()))
            elif command == CALL_COMMAND:
                name, kwargs = data
                attr = getattr(env, name)
                conn.send(attr(**(kwargs or {})) if callable(attr) else attr)
            elif command == COUNT_EPISODES_COMMAND:
                conn.send(len(getattr(env, "episodes", ())))
            elif command == ATTACH_COMMAND:
                row, specs = data
                for k, (name, shape, dtype) in specs.items():
                    shm = shared_memory.SharedMemory(name=name)
                    shms.append(shm)
                    slabs[k] = np.ndarray(shape, dtype=np.dtype(dtype), buffer=shm.buf)
                conn.send(True)
            elif command == CLOSE_COMMAND:
                break
            else:
                raise NotImplementedError(f"Unknown command {command}")
    except KeyboardInterrupt:
        pass
    finally:
        slabs.clear()
        for shm in shms:
            shm.close()
        try:
            env.close()
        finally:
            conn.close()


class VectorEnv:
    """`VectorEnv(make_env_fn, env_fn_args, ...)` -- see the module docstring.  `env_fn_args[i]` are the arguments of env i."""

    def __init__(self, make_env_fn: Callable[..., Any], env_fn_args: Sequence[Tuple], auto_reset_done: bool = True,
                 multiprocessing_start_method: str = "forkserver", workers_ignore_signals: bool = False,
                 shared_obs: bool = True) -> None:
        assert len(env_fn_args) > 0, "number of environments to be created should be greater than 0"
        assert multiprocessing_start_method in {"forkserver", "spawn", "fork"}
        self._is_closed = True
        self._auto_reset_done = auto_reset_done
        self._mp_ctx = mp.get_context(multiprocessing_start_method)
        self._conns: List[Any] = []
        self._workers: List[Any] = []
        self._paused: List[Tuple[int, Any, Any, int]] = []
        self._waiting: List[bool] = []
        for args in env_fn_args:
            parent, child = self._mp_ctx.Pipe(duplex=True)
            p = self._mp_ctx.Process(target=_worker, args=(child, parent, make_env_fn, tuple(args), auto_reset_done,
                                                           workers_ignore_signals), daemon=True)
            p.start()
            child.close()
            self._conns.append(parent)
            self._workers.append(p)
            self._waiting.append(False)
        self._is_closed = False
        self._rows = list(range(len(self._conns)))  # slab row of the env at each (unpaused) index
        self.observation_spaces = self._call_all("observation_space")
        self.action_spaces = self._call_all("action_space")
        self.orig_action_spaces = self._call_all("original_action_space")
        self.number_of_episodes = self._call_all("number_of_episodes")
        # ---- shared observation plane ----
        self._shms: List[shared_memory.SharedMemory] = []
        self._slabs: Dict[str, np.ndarray] = {}
        self._registered: List[int] = []
        self._fence = None
        if shared_obs:
            specs = _slab_specs(self.observation_spaces[0])
            wire = {}
            n = len(self._conns)
            for k, (shape, dtype) in specs.items():
                nbytes = max(1, int(np.prod((n,) + shape)) * np.dtype(dtype).itemsize)
                shm = shared_memory.SharedMemory(create=True, size=nbytes)
                self._shms.append(shm)
                self._slabs[k] = np.ndarray((n,) + shape, dtype=np.dtype(dtype), buffer=shm.buf)
                wire[k] = (shm.name, (n,) + shape, dtype)
            for i, c in enumerate(self._conns):
                c.send((ATTACH_COMMAND, (i, wire)))
            for c in self._conns:
                assert c.recv() is True

    # ---- plumbing --------------------------------------------------------------------------------------------------
    def _call_all(self, name: str, kwargs: Optional[dict] = None) -> List[Any]:
        for c in self._conns:
            c.send((CALL_COMMAND, (name, kwargs)))
        return [c.recv() for c in self._conns]

    def _wait_fence(self) -> None:
        if self._fence is not None:  # an upload is still reading the slab rows: finish it before any worker writes again
            self._fence.synchronize()
            self._fence = None

    def _obs_of(self, index_env: int, rest) -> Dict[str, np.ndarray]:
        if not self._slabs:
            return rest
        row = self._rows[index_env]
        obs = {k: slab[row] for k, slab in self._slabs.items()}
        obs.update(rest)
        return obs

    @property
    def num_envs(self) -> int:
        return len(self._conns)

    @property
    def shared_obs_keys(self) -> List[str]:
        return list(self._slabs.keys())

    # ---- queries ----------------------------------------------------------------------------------------------------
    def current_episodes(self):
        return self._call_all("current_episode")

    def count_episodes(self):
        for c in self._conns:
            c.send((COUNT_EPISODES_COMMAND, None))
        return [c.recv() for c in self._conns]

    def episode_over(self):
        return self._call_all("episode_over")

    def get_metrics(self):
        return self._call_all("get_metrics")

    def call_at(self, index_env: int, function_name: str, function_args: Optional[Dict[str, Any]] = None) -> Any:
        self._conns[index_env].send((CALL_COMMAND, (function_name, function_args)))
        return self._conns[index_env].recv()

    def call(self, function_names: List[str], function_args_list: Optional[List[Any]] = None) -> List[Any]:
        if function_args_list is None:
            function_args_list = [None] * len(function_names)
        assert len(function_names) == len(function_args_list)
        for c, name, kw in zip(self._conns, function_names, function_args_list):
            c.send((CALL_COMMAND, (name, kw)))
        return [c.recv() for c in self._conns]

    # ---- stepping ---------------------------------------------------------------------------------------------------
    def reset(self):
        self._wait_fence()
        for c in self._conns:
            c.send((RESET_COMMAND, None))
        return [self._obs_of(i, c.recv()) for i, c in enumerate(self._conns)]

    def reset_at(self, index_env: int):
        self._wait_fence()
        self._conns[index_env].send((RESET_COMMAND, None))
        return self._obs_of(index_env, self._conns[index_env].recv())

    def async_step_at(self, index_env: int, action) -> None:
        self._wait_fence()
        if isinstance(action, np.ndarray) and action.ndim == 0:
            action = action.item()
        self._conns[index_env].send((STEP_COMMAND, action))
        self._waiting[index_env] = True

    def wait_step_at(self, index_env: int):
        obs, reward, done, info = self._conns[index_env].recv()
        self._waiting[index_env] = False
        return self._obs_of(index_env, obs), reward, done, info

    def poll_steps(self, timeout: float = 0.0, max_messages: Optional[int] = None) -> List[Tuple[int, Tuple[Any, float, bool, dict]]]:
        """Non-blocking counterpart of wait_step_at for ANY environment (VER's inference queue, rl/ver/inference_worker.py:458-470):
        results of the outstanding steps that have arrived within `timeout` seconds, as (index_env, (obs, reward, done, info))."""
        from multiprocessing.connection import wait
        pending = {self._conns[i]: i for i in range(len(self._conns)) if self._waiting[i]}
        if not pending:
            return []
        out = []
        for conn in wait(list(pending.keys()), timeout):
            i = pending[conn]
            obs, reward, done, info = conn.recv()
            self._waiting[i] = False
            out.append((i, (self._obs_of(i, obs), reward, done, info)))
            if max_messages is not None and len(out) >= max_messages:
                break
        return out

    def gather_obs(self, env_indices: Sequence[int], device) -> Dict[str, Any]:
        """Observations of an arbitrary list of environments (whose steps have been received) as device tensors: one gather per
        slab sensor on the host, one upload."""
        import torch
        assert self._slabs and not self._paused
        rows = [self._rows[i] for i in env_indices]
        dev = torch.device(device)
        return {k: torch.from_numpy(self._slabs[k][rows]).to(dev, non_blocking=False) for k in
                sorted(self._slabs, key=lambda k: -self._slabs[k][0].nbytes)}

    def step_at(self, index_env: int, action):
        self.async_step_at(index_env, action)
        return self.wait_step_at(index_env)

    def async_step(self, data: Sequence[Any]) -> None:
        for i, a in enumerate(data):
            self.async_step_at(i, a)

    def wait_step(self) -> List[Any]:
        return [self.wait_step_at(i) for i in range(self.num_envs)]

    def step(self, data: Sequence[Any]) -> List[Any]:
        self.async_step(data)
        return self.wait_step()

    def post_step(self, observations):
        return observations

    # ---- batched upload (replaces batch_obs for the slab sensors) --------------------------------------------------------
    def _pin(self) -> None:
        """Page-lock the slabs once so the uploads are true asynchronous DMA (hipHostRegister via torch's runtime handle)."""
        import torch
        if self._registered or not torch.cuda.is_available():
            return
        rt = torch.cuda.cudart()
        for slab in self._slabs.values():
            ptr = slab.ctypes.data
            if int(rt.cudaHostRegister(ptr, slab.nbytes, 0)) == 0:
                self._registered.append(ptr)

    def batched_obs(self, env_slice: slice, device, out: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        """Observations of the envs in `env_slice` (all of them stepped and waited for) as device tensors: one H2D copy per
        slab sensor, largest first (utils/common.py:279-290), straight from the worker-written rows.  With `out`, copies into
        the given tensors (rollout-storage rows) instead of allocating."""
        import torch
        dev = torch.device(device)
        assert self._slabs and not self._paused, "batched_obs needs the shared observation plane and no paused envs"
        if dev.type == "cuda":
            self._pin()
        res = {}
        for k in sorted(self._slabs, key=lambda k: -self._slabs[k][0].nbytes):
            src = torch.from_numpy(self._slabs[k][env_slice])
            if out is not None and k in out:
                out[k].copy_(src, non_blocking=True)
                res[k] = out[k]
            else:
                res[k] = src.to(dev, non_blocking=True) if dev.type == "cuda" else src.clone()
        if dev.type == "cuda":
            self._fence = torch.cuda.Event()
            self._fence.record()
        return res

    # ---- pausing ----------------------------------------------------------------------------------------------------
    def pause_at(self, index: int) -> None:
        if self._waiting[index]:
            self._conns[index].recv()
        self._paused.append((index, self._conns.pop(index), self._workers.pop(index), self._rows.pop(index)))
        self._waiting.pop(index)

    def resume_all(self) -> None:
        for index, conn, worker, row in reversed(self._paused):
            self._conns.insert(index, conn)
            self._workers.insert(index, worker)
            self._rows.insert(index, row)
            self._waiting.insert(index, False)
        self._paused = []

    # ---- teardown ---------------------------------------------------------------------------------------------------
    def close(self) -> None:
        if self._is_closed:
            return
        self._wait_fence()
        for i, c in enumerate(self._conns):
            if self._waiting[i]:
                c.recv()
        conns = self._conns + [c for _, c, _, _ in self._paused]
        workers = self._workers + [w for _, _, w, _ in self._paused]
        for c in conns:
            try:
                c.send((CLOSE_COMMAND, None))
            except (BrokenPipeError, OSError):
                pass
        for w in workers:
            w.join(10)
            if w.is_alive():
                w.terminate()
        for c in conns:
            c.close()
        if self._registered:
            import torch
            rt = torch.cuda.cudart()
            for ptr in self._registered:
                rt.cudaHostUnregister(ptr)
            self._registered = []
        self._slabs = {}
        for shm in self._shms:
            try:
                shm.close()
            except BufferError:  # a caller still holds an observation view of the slab: the mapping dies with the last view
                pass
            try:
                shm.unlink()
            except FileNotFoundError:
                pass
        self._shms = []
        self._is_closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()
