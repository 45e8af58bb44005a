"""DD-PPO updater: decentralised synchronous gradient averaging (habitat_baselines/rl/ddppo/algo/ddppo.py:59-157).

The reference wraps `evaluate_actions` in DistributedDataParallel so that backward all-reduces ~60-170
parameter tensors in 25 MiB buckets.  Here every gradient already lives in ONE flat fp32 arena, so the
exchange is a single RCCL all-reduce (sum) of that arena per minibatch; the 1/world_size is folded into the
fused clip+Adam kernel (`grad_scale`), and the initial weight broadcast (DDP ctor, C2 in SURVEY.md) is one
broadcast of the parameter arena.  Message size 34-58 MB: latency-bound on xGMI, <1% of a minibatch."""
from __future__ import annotations

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
            # RunningMeanAndVar batch moments are averaged over ranks inside the engine's forward
            # (rl/ddppo/policy/running_mean_and_var.py:38-41,47-49): all_reduce(sum) then / world_size
            def _avg(view: torch.Tensor, scale: float) -> None:
                distrib.all_reduce(view)
                view.mul_(scale)

            eng.set_allreduce(_avg, world)
        self._distributed = True

    def _all_reduce_grads(self) -> None:
        if distrib.is_initialized() and distrib.get_world_size() > 1:
            distrib.all_reduce(self.actor_critic.engine.grads_flat)

    def _all_reduce_scalar_stats(self, t: torch.Tensor) -> None:
        if distrib.is_initialized() and distrib.get_world_size() > 1:
            distrib.all_reduce(t)


@baseline_registry.register_updater
class DDPPO(DecentralizedDistributedMixin, PPO):
    pass
