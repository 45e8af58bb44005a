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
trip in the middle of a pass (what DistributedDataParallel's C++ reducer gives the reference).  The communicator is self-tested against
torch.distributed at start-up; if librccl is missing, creation fails or the self-test disagrees, the callback form below is used and a
warning says so.  HAB_NATIVE_COMM=0 selects the callback form; gloo (the CPU / shared-GPU tests) always uses it."""
from __future__ import annotations

import os

import torch
import torch.distributed as distrib

from habitat_amd.common.baseline_registry import baseline_registry
from habitat_amd.rl.ppo.ppo import PPO


class DecentralizedDistributedMixin:
    def _world_size(self) -> int:
        return distrib.get_world_size() if distrib.is_initialized() else 1

    def init_distributed(self, find_unused_params: bool = True) -> None:
        """Broadcast rank 0's parameters (what the DDP constructor does in the reference, ddppo.py:110-140)."""
        assert distrib.is_initialized(), "Distributed must be initialized"
        eng = self.actor_critic.engine
        distrib.broadcast(eng.params_flat, src=0)
        eng.repack()
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
    def _want_native_comm(world: int) -> bool:
        """RCCL process groups only (a gloo group means CPU tensors or several ranks on one GPU, which RCCL refuses); a single rank
        takes it only when forced (HAB_NATIVE_COMM=1: the 1-GPU test of the plumbing)."""
        flag = os.environ.get("HAB_NATIVE_COMM")
        if flag == "0" or distrib.get_backend() != "nccl" or not torch.cuda.is_available():
            return False
        return world > 1 or flag == "1"

    @staticmethod
    def _open_native_comm(eng, world: int):
        from habitat_amd.engine import NativeComm
        from habitat_amd.utils.logging import logger
        try:
            comm = NativeComm(world, distrib.get_rank(), exchange=lambda obj: distrib.broadcast_object_list(obj, src=0))
            # self-test against torch.distributed: the same sum, bit for bit (integers in fp32)
            probe = torch.arange(1, 257, device=eng.params_flat.device, dtype=torch.float32) * float(distrib.get_rank() + 1)
            ref = probe.clone()
            distrib.all_reduce(ref)
            comm.all_reduce_sum_(probe)
            torch.cuda.current_stream().synchronize()
            ok = torch.tensor([1.0 if torch.equal(probe, ref) else 0.0], device=probe.device)
            distrib.all_reduce(ok, op=distrib.ReduceOp.MIN)  # every rank takes the same decision
            if float(ok.item()) != 1.0:
                raise RuntimeError("the communicator's all-reduce disagrees with torch.distributed")
            return comm
        except Exception as exc:  # noqa: BLE001 -- fall back loudly, on every rank alike (a failure here is symmetric or fatal)
            logger.warning(f"device-side DD-PPO exchange unavailable ({exc!r}): using the torch.distributed callbacks")
            return None

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
