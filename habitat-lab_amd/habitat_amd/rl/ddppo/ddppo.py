"""DD-PPO updater: decentralised synchronous gradient averaging (habitat_baselines/rl/ddppo/algo/ddppo.py:59-157).

The reference wraps `evaluate_actions` in DistributedDataParallel so that backward all-reduces ~60-170
parameter tensors in 25 MiB buckets, overlapped with the rest of backward.  Here every gradient already lives in
ONE flat fp32 arena and the overlap needs exactly two messages: the engine reports (`hab_policy_set_grad_ready`)
when the TAIL of the arena -- visual fc, recurrent encoder, heads: 99.5 % of the SimpleCNN policy's bytes, 63 % of
ResNet18's -- is final, which is BEFORE the convolution stack's backward starts; that range is all-reduced
asynchronously on RCCL's stream while the convolutions run, and the small head of the arena follows after backward.
The 1/world_size is folded into the fused clip+Adam kernel (`grad_scale`); the initial weight broadcast (DDP ctor)
is one broadcast of the parameter arena.  HAB_NO_GRAD_OVERLAP=1 restores the single blocking all-reduce."""
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
        if world > 1:
            # RunningMeanAndVar batch moments + frame count are summed over ranks inside the engine's forward
            # (rl/ddppo/policy/running_mean_and_var.py:38-41,47-49); the engine divides the means / variances by world_size
            # where it consumes them and merges with the summed count
            def _avg(view: torch.Tensor, scale: float) -> None:
                distrib.all_reduce(view)
                if scale != 1.0:
                    view.mul_(scale)

            eng.set_allreduce(_avg, world)
        self._grad_work = None  # (work handle, first) of the early all-reduce of grads_flat[first:]
        overlap = (world > 1 and os.environ.get("HAB_NO_GRAD_OVERLAP") is None) or os.environ.get("HAB_FORCE_GRAD_OVERLAP") is not None
        if overlap and hasattr(eng, "set_grad_ready"):
            def _tail_ready(first: int, count: int) -> None:
                g = eng.grads_flat
                assert first + count == g.numel()
                self._grad_work = (distrib.all_reduce(g[first:], async_op=True), first)

            eng.set_grad_ready(_tail_ready)
        self._distributed = True

    def _all_reduce_grads(self) -> None:
        if not distrib.is_initialized():
            return
        g = self.actor_critic.engine.grads_flat
        pending, self._grad_work = getattr(self, "_grad_work", None), None
        if pending is None:
            if distrib.get_world_size() > 1:
                distrib.all_reduce(g)
            return
        work, first = pending
        if first > 0:
            distrib.all_reduce(g[:first])  # the convolution stack's gradients, produced after the tail
        work.wait()                        # the current stream now waits for the early all-reduce

    def _all_reduce_scalar_stats(self, t: torch.Tensor) -> None:
        if distrib.is_initialized() and distrib.get_world_size() > 1:
            distrib.all_reduce(t)


@baseline_registry.register_updater
class DDPPO(DecentralizedDistributedMixin, PPO):
    pass
