"""RolloutStorage: device-resident (T+1, N, ...) rollout arena.

Public surface = habitat_baselines/common/rollout_storage.py:24-275 (+ the Storage ABC,
common/storage.py:10-56): same constructor, `buffers` TensorDict with the same keys, insert /
advance_rollout / after_update / compute_returns / data_generator / get_current_step / get_last_step.

MI355X-native differences (behaviour-preserving):
  * compute_returns is one HIP launch (hab_compute_returns) instead of T x ~8 tiny tensor ops;
  * data_generator does NOT copy the minibatch (the reference gathers ~1.9 GB of observations per
    minibatch, rollout_storage.py:237-246): it yields a `MiniBatch` carrying `rows` (frame -> arena
    row) and the pack-info; the kernels gather through `rows`.  Reference-style consumers that index
    the batch like a dict still work -- entries are materialised lazily on first access.
"""
from __future__ import annotations

import warnings
from typing import Any, Dict, Iterator, Optional

import numpy as np
import torch

from habitat_amd import _lib
from habitat_amd._lib import check, ptr, stream_ptr
from habitat_amd.common.baseline_registry import baseline_registry
from habitat_amd.common.spaces import get_action_space_info
from habitat_amd.common.tensor_dict import TensorDict
from habitat_amd.engine import DevicePackInfo

GAE_VARIANTS = {"exact": 0, "scan": 1}


class Storage:
    """Interface of habitat_baselines/common/storage.py:10-56."""

    def insert(self, *a, **k): raise NotImplementedError
    def to(self, device): raise NotImplementedError
    def insert_first_observations(self, batch): raise NotImplementedError
    def advance_rollout(self, buffer_index: int = 0): raise NotImplementedError
    def compute_returns(self, next_value, use_gae, gamma, tau): raise NotImplementedError
    def after_update(self): raise NotImplementedError
    def get_last_step(self): raise NotImplementedError
    def get_current_step(self, env_slice, buffer_index): raise NotImplementedError


_LAZY = ("observations", "recurrent_hidden_states", "prev_actions", "masks", "actions", "action_log_probs", "value_preds",
         "returns", "rewards", "advantages", "rnn_build_seq_info")


class MiniBatch(dict):
    """One PPO minibatch = env columns `inds` x rows 0..T-1 of the arena, flattened time-major.
    Zero-copy for the fused updater (`rows`, `pack`); dict-style access materialises copies."""

    def __init__(self, storage: "RolloutStorage", inds: torch.Tensor, T: int, advantages: Optional[torch.Tensor],
                 dones_cpu: np.ndarray):
        super().__init__()
        self.storage, self.inds, self.T, self.n = storage, inds, T, len(inds)
        self.advantages_full = advantages
        N = storage._num_envs
        inds_np = inds.numpy().astype(np.int64)
        rows = (np.arange(T, dtype=np.int64)[:, None] * N + inds_np[None, :]).reshape(-1)
        self.rows_cpu = torch.from_numpy(rows.astype(np.int32))
        self.rows = self.rows_cpu.to(storage.device, non_blocking=True)
        self._dones = np.ascontiguousarray(dones_cpu[0:T, inds_np].reshape(-1, len(inds)))
        self._pack = None

    @property
    def pack(self) -> DevicePackInfo:
        if self._pack is None:
            self._pack = DevicePackInfo(self._dones, self.storage.device)
        return self._pack

    def __missing__(self, key):
        if key not in _LAZY:
            raise KeyError(key)
        b, T, inds = self.storage.buffers, self.T, self.inds.to(self.storage.device)
        if key == "advantages":
            v = self.advantages_full[0:T, inds].flatten(0, 1)
        elif key == "recurrent_hidden_states":
            v = b[key][0:1, inds].flatten(0, 1)
        elif key == "observations":
            v = b[key][0:T, inds].map(lambda t: t.flatten(0, 1))
        elif key == "rnn_build_seq_info":
            v = TensorDict()
            for k, arr in self.pack.arrays.items():
                t = torch.from_numpy(np.ascontiguousarray(arr))
                dict.__setitem__(v, f"cpu_{k}", t)
                dict.__setitem__(v, k, t.to(self.storage.device))
        else:
            v = b[key][0:T, inds].flatten(0, 1)
        dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in _LAZY

    def to_tree(self):
        return self


@baseline_registry.register_storage
class RolloutStorage(Storage):
    r"""Class for storing rollout information for RL trainers."""

    def __init__(self, numsteps, num_envs, observation_space, action_space, actor_critic, is_double_buffered: bool = False,
                 device=None, gae_variant: str = "scan"):
        action_shape, discrete_actions = get_action_space_info(action_space)
        if device is None:
            device = getattr(actor_critic, "device", None) or torch.device("cpu")
        self.device = torch.device(device)
        dev = self.device
        self.buffers = TensorDict()
        obs = TensorDict()
        for sensor in observation_space.spaces:
            sp = observation_space.spaces[sensor]
            dict.__setitem__(obs, sensor, torch.zeros((numsteps + 1, num_envs, *sp.shape),
                                                      dtype=getattr(torch, np.dtype(sp.dtype).name), device=dev))
        dict.__setitem__(self.buffers, "observations", obs)
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
        B = self.buffers
        dict.__setitem__(B, "recurrent_hidden_states",
                         z(numsteps + 1, num_envs, actor_critic.num_recurrent_layers, actor_critic.recurrent_hidden_size))
        for k in ("rewards", "value_preds", "returns", "action_log_probs"):
            dict.__setitem__(B, k, z(numsteps + 1, num_envs, 1))
        adt = torch.long if discrete_actions else torch.float32
        dict.__setitem__(B, "actions", z(numsteps + 1, num_envs, *action_shape, dtype=adt))
        dict.__setitem__(B, "prev_actions", z(numsteps + 1, num_envs, *action_shape, dtype=adt))
        dict.__setitem__(B, "masks", z(numsteps + 1, num_envs, 1, dtype=torch.bool))
        self.is_double_buffered = is_double_buffered
        self._nbuffers = 2 if is_double_buffered else 1
        self._num_envs = num_envs
        assert (self._num_envs % self._nbuffers) == 0
        self.num_steps = numsteps
        self.current_rollout_step_idxs = [0 for _ in range(self._nbuffers)]
        self.gae_variant = gae_variant
        self._adv = None

    @property
    def current_rollout_step_idx(self) -> int:
        assert all(s == self.current_rollout_step_idxs[0] for s in self.current_rollout_step_idxs)
        return self.current_rollout_step_idxs[0]

    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self.buffers.map_in_place(lambda v: v.to(device))
            self.device = device

    def _env_slice(self, buffer_index):
        return slice(int(buffer_index * self._num_envs / self._nbuffers), int((buffer_index + 1) * self._num_envs / self._nbuffers))

    def insert(self, next_observations=None, next_recurrent_hidden_states=None, actions=None, action_log_probs=None,
               value_preds=None, rewards=None, next_masks=None, buffer_index: int = 0, **kwargs):
        if not self.is_double_buffered:
            assert buffer_index == 0
        next_step = dict(observations=next_observations, recurrent_hidden_states=next_recurrent_hidden_states,
                         prev_actions=actions, masks=next_masks)
        current_step = dict(actions=actions, action_log_probs=action_log_probs, value_preds=value_preds, rewards=rewards)
        next_step = {k: v for k, v in next_step.items() if v is not None}
        current_step = {k: v for k, v in current_step.items() if v is not None}
        env_slice = self._env_slice(buffer_index)
        idx = self.current_rollout_step_idxs[buffer_index]
        if next_step:
            self.buffers.set((idx + 1, env_slice), next_step, strict=False)
        if current_step:
            self.buffers.set((idx, env_slice), current_step, strict=False)

    def advance_rollout(self, buffer_index: int = 0):
        self.current_rollout_step_idxs[buffer_index] += 1

    def after_update(self):
        self.buffers[0] = self.buffers[self.current_rollout_step_idx]
        self.current_rollout_step_idxs = [0 for _ in self.current_rollout_step_idxs]

    def compute_returns(self, next_value, use_gae, gamma, tau):
        """rollout_storage.py:174-205 as ONE kernel launch on the (T+1, N) buffers."""
        if self.device.type != "cuda":
            raise _lib.HabError("RolloutStorage.compute_returns needs the arena on a GPU (no CPU fallback)")
        B = self.buffers
        T, N = self.current_rollout_step_idx, self._num_envs
        nv = next_value.reshape(-1).contiguous().float()
        check(_lib.lib().hab_compute_returns(ptr(B["rewards"]), ptr(B["value_preds"]), ptr(B["masks"]), ptr(B["returns"]),
                                             ptr(nv), T, N, float(gamma), float(tau), int(bool(use_gae)),
                                             GAE_VARIANTS[self.gae_variant], stream_ptr()), "hab_compute_returns")

    def data_generator(self, advantages: Optional[torch.Tensor], num_mini_batch: int) -> Iterator[MiniBatch]:
        num_environments = self._num_envs
        assert num_environments >= num_mini_batch, (
            "Trainer requires the number of environments ({}) to be greater than or equal to the number of "
            "trainer mini batches ({}).".format(num_environments, num_mini_batch))
        if num_environments % num_mini_batch != 0:
            warnings.warn("Number of environments ({}) is not a multiple of the number of mini batches ({}).  This results in "
                          "mini batches of different sizes, which can harm training performance.".format(
                              num_environments, num_mini_batch))
        dones_cpu = torch.logical_not(self.buffers["masks"]).cpu().view(-1, self._num_envs).numpy()
        for inds in torch.randperm(num_environments).chunk(num_mini_batch):
            yield MiniBatch(self, inds, self.current_rollout_step_idx, advantages, dones_cpu)

    def insert_first_observations(self, batch):
        self.buffers["observations"][0] = batch  # type: ignore

    def get_current_step(self, env_slice, buffer_index):
        return self.buffers[self.current_rollout_step_idxs[buffer_index], env_slice]

    def get_last_step(self):
        return self.buffers[self.current_rollout_step_idx]

    def __getstate__(self) -> Dict[str, Any]:
        return self.__dict__

    def __setstate__(self, state: Dict[str, Any]):
        self.__dict__.update(state)
