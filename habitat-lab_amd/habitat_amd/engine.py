"""Thin Python owner of a `hab_policy` engine handle (include/habitat_amd.h): allocates the flat
parameter / gradient / packed-weight / workspace arenas with torch (device memory plumbing only) and
forwards act / evaluate / backward to the C-ABI.  No arithmetic happens in this file."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import Obs, PackInfo, PolicyDesc, check, ptr, stream_ptr

ARCH = {"simple_cnn": 0, "resnet": 1}
RNN = {"GRU": 0, "LSTM": 1}


class DevicePackInfo:
    """build_pack_info_from_dones (rl/models/rnn_state_encoder.py:155-168) through the C++ host builder,
    plus int32 device copies for the kernels."""

    env_first = None  # set by from_ids: first_step_for_env (frames in any order)

    @classmethod
    def from_ids(cls, episode_ids: np.ndarray, environment_ids: np.ndarray, step_ids: np.ndarray, device=None) -> "DevicePackInfo":
        """build_pack_info_from_episode_ids (rnn_state_encoder.py:35-150) for P frames in ANY order -- a VER minibatch
        (rl/ver/ver_rollout_storage.py:586-617)."""
        self = cls.__new__(cls)
        ep, env, st = (np.ascontiguousarray(a, dtype=np.int64).reshape(-1) for a in (episode_ids, environment_ids, step_ids))
        P = ep.size
        a = {"select_inds": np.empty(P, np.int64), "num_seqs_at_step": np.empty(P, np.int64), "sequence_starts": np.empty(P, np.int64),
             "sequence_lengths": np.empty(P, np.int64), "rnn_state_batch_inds": np.empty(P, np.int64),
             "last_sequence_in_batch_mask": np.zeros(P, np.uint8), "first_sequence_in_batch_mask": np.zeros(P, np.uint8),
             "first_step_for_env": np.empty(P, np.int64)}
        nf, ml, ne = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(_lib.lib().hab_build_pack_info_from_ids(ep.ctypes.data, env.ctypes.data, st.ctypes.data, P, *[v.ctypes.data for v in a.values()],
                                                      C.byref(nf), C.byref(ml), C.byref(ne)), "hab_build_pack_info_from_ids")
        F, max_len, n = nf.value, ml.value, ne.value
        self.T, self.N, self.P, self.F, self.max_len = None, n, P, F, max_len
        last, first = a["last_sequence_in_batch_mask"][:F].astype(bool), a["first_sequence_in_batch_mask"][:F].astype(bool)
        self.arrays = {"select_inds": a["select_inds"], "num_seqs_at_step": a["num_seqs_at_step"][:max_len],
                       "sequence_starts": a["sequence_starts"][:F], "sequence_lengths": a["sequence_lengths"][:F],
                       "rnn_state_batch_inds": a["rnn_state_batch_inds"][:F], "last_sequence_in_batch_mask": last,
                       "first_sequence_in_batch_mask": first, "last_sequence_in_batch_inds": np.nonzero(last)[0],
                       "first_episode_in_batch_inds": np.nonzero(first)[0], "first_step_for_env": a["first_step_for_env"][:n]}
        self.env_first = np.ascontiguousarray(self.arrays["first_step_for_env"], dtype=np.int32)
        self._nseq = np.ascontiguousarray(self.arrays["num_seqs_at_step"], dtype=np.int32)
        self._off = np.zeros(max_len + 1, np.int32)
        self._off[1:] = np.cumsum(self._nseq)
        self.struct = None
        if device is not None:
            self.to(device)
        return self

    def __init__(self, dones: np.ndarray, device=None):
        dones = np.ascontiguousarray(dones, dtype=np.uint8)
        T, N = dones.shape
        P = T * N
        L = _lib.lib()
        a = {
            "select_inds": np.empty(P, np.int64), "num_seqs_at_step": np.empty(T, np.int64),
            "sequence_starts": np.empty(P, np.int64), "sequence_lengths": np.empty(P, np.int64),
            "rnn_state_batch_inds": np.empty(P, np.int64), "last_sequence_in_batch_mask": np.zeros(P, np.uint8),
            "first_sequence_in_batch_mask": np.zeros(P, np.uint8), "last_sequence_in_batch_inds": np.empty(N, np.int64),
            "first_episode_in_batch_inds": np.empty(N, np.int64), "first_step_for_env": np.empty(N, np.int64),
        }
        nf, ml = C.c_int32(0), C.c_int32(0)
        check(L.hab_build_pack_info(dones.ctypes.data, T, N, *[v.ctypes.data for v in a.values()], C.byref(nf), C.byref(ml)),
              "hab_build_pack_info")
        F, max_len = nf.value, ml.value
        self.T, self.N, self.P, self.F, self.max_len = T, N, P, F, max_len
        self.arrays = {
            "select_inds": a["select_inds"], "num_seqs_at_step": a["num_seqs_at_step"][:max_len],
            "sequence_starts": a["sequence_starts"][:F], "sequence_lengths": a["sequence_lengths"][:F],
            "rnn_state_batch_inds": a["rnn_state_batch_inds"][:F],
            "last_sequence_in_batch_mask": a["last_sequence_in_batch_mask"][:F].astype(bool),
            "first_sequence_in_batch_mask": a["first_sequence_in_batch_mask"][:F].astype(bool),
            "last_sequence_in_batch_inds": a["last_sequence_in_batch_inds"],
            "first_episode_in_batch_inds": a["first_episode_in_batch_inds"], "first_step_for_env": a["first_step_for_env"],
        }
        self._nseq = np.ascontiguousarray(self.arrays["num_seqs_at_step"], dtype=np.int32)
        self._off = np.zeros(max_len + 1, np.int32)
        self._off[1:] = np.cumsum(self._nseq)
        self.struct = None
        if device is not None:
            self.to(device)

    def to(self, device):
        parts = [self.arrays["select_inds"], self.arrays["rnn_state_batch_inds"], self.arrays["sequence_starts"]]
        if self.env_first is not None:
            parts.append(self.env_first)
        packed = np.concatenate(parts).astype(np.int32)
        self._dev = torch.from_numpy(packed).to(device, non_blocking=True)
        P, F = self.P, self.F
        s = PackInfo()
        base = self._dev.data_ptr()
        s.select_inds = base
        s.frag_env = base + 4 * P
        s.frag_start = base + 4 * (P + F)
        s.env_first_frame = (base + 4 * (P + 2 * F)) if self.env_first is not None else None
        s.step_offsets_host = self._off.ctypes.data
        s.num_seqs_at_step_host = self._nseq.ctypes.data
        s.P, s.F, s.max_len = P, F, self.max_len
        self.struct = s
        return self


class PolicyEngine:
    def __init__(self, *, arch="simple_cnn", backbone=18, baseplanes=32, normalize_visual_inputs=False, rnn_type="GRU",
                 rnn_layers=1, hidden=512, num_actions=4, H=256, W=256, has_rgb=True, has_depth=True, goal_dim=2,
                 max_frames=4096, max_envs=64, device="cuda", with_grads=True, visual_order=("rgb", "depth", "semantic"),
                 has_semantic=False, num_object_categories=0, has_compass=False, has_gps=False, action_dist="categorical",
                 gauss_flags=0, gauss_min_std=0.0, gauss_max_std=0.0, pointgoal_dim=0, proximity_dim=0):
        L = _lib.lib()
        self.L = L
        d = PolicyDesc(ARCH[arch], backbone, baseplanes, int(normalize_visual_inputs), RNN[rnn_type.upper()], rnn_layers, hidden,
                       num_actions, H, W, int(has_rgb), int(has_depth), goal_dim, max_frames, max_envs,
                       sum({"rgb": 1, "depth": 2, "semantic": 3}[k] << (2 * i) for i, k in enumerate(visual_order)), int(has_semantic),
                       int(num_object_categories), int(has_compass), int(has_gps), {"categorical": 0, "gaussian": 1}[action_dist],
                       int(gauss_flags), float(gauss_min_std), float(gauss_max_std), int(pointgoal_dim), int(proximity_dim))
        self.action_dist = action_dist
        self.desc = d
        h = C.c_void_p()
        check(L.hab_policy_create(C.byref(d), C.byref(h)), "hab_policy_create")
        self.h = h
        self.device = torch.device(device)
        self.hidden, self.rnn_layers, self.rnn_type = hidden, rnn_layers, rnn_type.upper()
        self.Lh = rnn_layers * (2 if self.rnn_type == "LSTM" else 1)
        self.num_actions = num_actions
        n = L.hab_policy_num_params(h)
        self.param_floats = L.hab_policy_param_floats(h)
        self.specs = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd, off = C.c_int(0), C.c_int64(0)
        for i in range(n):
            check(L.hab_policy_param_info(h, i, name, 256, shape, C.byref(nd), C.byref(off)), "hab_policy_param_info")
            self.specs.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), int(off.value)))
        self.buffer_names = {self.specs[i][0] for i in range(n) if L.hab_policy_param_is_buffer(h, i) == 1}
        dev = self.device
        self.params_flat = torch.zeros(self.param_floats, dtype=torch.float32, device=dev)
        self.grads_flat = torch.zeros(self.param_floats, dtype=torch.float32, device=dev) if with_grads else None
        self.packed = torch.zeros(L.hab_policy_packed_floats(h), dtype=torch.float32, device=dev)
        self.work_floats = L.hab_policy_work_floats(h)
        self.work = torch.empty(self.work_floats, dtype=torch.float32, device=dev)
        check(L.hab_policy_bind(h, ptr(self.params_flat), ptr(self.grads_flat), ptr(self.packed), ptr(self.work), self.work_floats),
              "hab_policy_bind")
        self.views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.grad_views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for nm, shp, o in self.specs:
            numel = int(np.prod(shp))
            self.views[nm] = self.params_flat[o:o + numel].view(shp)
            if with_grads:
                self.grad_views[nm] = self.grads_flat[o:o + numel].view(shp)
        self._packed_version = -1

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.hab_policy_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------
    def load(self, state: Dict[str, torch.Tensor]):
        with torch.no_grad():
            for nm, v in self.views.items():
                v.copy_(state[nm])
        self.repack()

    def state(self) -> Dict[str, torch.Tensor]:
        return OrderedDict((k, v.detach().clone()) for k, v in self.views.items())

    def set_training(self, mode: bool):
        check(self.L.hab_policy_set_training(self.h, int(bool(mode))), "hab_policy_set_training")

    def set_allreduce(self, fn, world_size: int):
        """fn(tensor_view, scale): in-place all-reduce of a small fp32 view of the workspace (DD-PPO RunningMeanAndVar)."""
        base = self.work.data_ptr()

        def _cb(buf, n, scale, _ctx):
            try:  # an exception must not vanish inside the ctypes trampoline: it is re-raised when the engine call returns
                off = (buf - base) // 4
                fn(self.work[off:off + n], scale)
            except BaseException as exc:  # noqa: BLE001
                self._cb_error = self._cb_error or exc

        self._allreduce_cb = _lib.ALLREDUCE_FN(_cb)  # keep the trampoline alive
        check(self.L.hab_policy_set_allreduce(self.h, self._allreduce_cb, None, int(world_size)), "hab_policy_set_allreduce")

    def set_grad_ready(self, fn):
        """fn(first, count): called inside backward() once grads_flat[first:first + count] (the tail from the visual fc weight on)
        has been produced -- DD-PPO starts its all-reduce of that range there, overlapped with the conv stack's backward."""
        if fn is None:
            self._grad_ready_cb = _lib.GRAD_READY_FN(0)
        else:
            def _cb(first, count, _ctx):
                try:
                    fn(int(first), int(count))
                except BaseException as exc:  # noqa: BLE001
                    self._cb_error = self._cb_error or exc

            self._grad_ready_cb = _lib.GRAD_READY_FN(_cb)
        check(self.L.hab_policy_set_grad_ready(self.h, self._grad_ready_cb, None), "hab_policy_set_grad_ready")

    # ---- device-side exchange (csrc/comm.hip) ----
    def set_comm(self, comm: "NativeComm | None"):
        """From now on backward() enqueues the all-reduce of every finished tail of the gradient arena on the communicator's stream and
        the training forward sums the RunningMeanAndVar moments on the compute stream; the Python callbacks are no longer called."""
        self._comm = comm  # keeps the communicator alive as long as the engine uses it
        check(self.L.hab_policy_set_comm(self.h, comm.h if comm is not None else None), "hab_policy_set_comm")

    def grad_sync(self):
        """After backward(): exchange the rest of the gradient arena and make the current stream wait for all of it (sums over ranks)."""
        check(self.L.hab_policy_grad_sync(self.h, stream_ptr()), "hab_policy_grad_sync")

    _cb_error = None
    _comm = None

    def _raise_cb_error(self):
        if self._cb_error is not None:
            exc, self._cb_error = self._cb_error, None
            raise _lib.HabError("a collective callback failed inside the engine call") from exc

    def _call(self, code_fn, what: str):
        """Runs one engine entry point that may invoke the collective callbacks.  A callback exception is attributed to THIS call
        (it takes precedence over the status code the interrupted call returns) and never survives into a later one."""
        self._cb_error = None
        try:
            code = code_fn()
        finally:
            self._raise_cb_error()
        check(code, what)

    def repack(self):
        check(self.L.hab_policy_repack(self.h, stream_ptr()), "hab_policy_repack")
        self._packed_version = self.params_flat._version

    def _fresh(self):
        if self._packed_version != self.params_flat._version:
            self.repack()

    # ---- calls ------------------------------------------------------------------------------
    @staticmethod
    def _obs(rgb, depth, goal, prev_actions=None, extra=None):
        """extra: optional dict with the ObjectNav sensors `semantic` (int32), `objectgoal` (int64), `compass`, `gps`, and the further
        1-D goal sensors `pointgoal`, `proximity` (float)."""
        dp = lambda t: t.data_ptr() if t is not None else None
        e = extra or {}
        return Obs(dp(rgb), dp(depth), dp(goal), dp(prev_actions), dp(e.get("semantic")), dp(e.get("objectgoal")), dp(e.get("compass")),
                   dp(e.get("gps")), dp(e.get("visual_features")), dp(e.get("pointgoal")), dp(e.get("proximity")))

    def visual_feature_shape(self):
        """(C, Hf, Wf) = ResNetEncoder.output_shape (resnet_policy.py:235-253)."""
        c, h, w = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self.L.hab_policy_visual_feature_shape(self.h, C.byref(c), C.byref(h), C.byref(w)), "hab_policy_visual_feature_shape")
        return (c.value, h.value, w.value)

    def encode(self, rgb, depth, n, out, extra=None):
        """The visual encoder alone on n frames -> out (n, C, Hf, Wf)."""
        self._fresh()
        o = self._obs(rgb, depth, None, None, extra)
        self._call(lambda: self.L.hab_policy_encode(self.h, C.byref(o), n, ptr(out), stream_ptr()), "hab_policy_encode")

    def act(self, rgb, depth, goal, hidden_in, masks, n, *, exp_noise=None, deterministic=False, values, actions=None,
            action_log_probs=None, hidden_out=None, probs_out=None, prev_actions=None, extra=None):
        self._fresh()
        o = self._obs(rgb, depth, goal, prev_actions, extra)
        self._call(lambda: self.L.hab_policy_act(self.h, C.byref(o), ptr(hidden_in), ptr(masks), ptr(exp_noise), int(deterministic), n,
                                                 ptr(values), ptr(actions), ptr(action_log_probs), ptr(hidden_out), ptr(probs_out),
                                                 stream_ptr()), "hab_policy_act")

    def evaluate(self, rgb, depth, goal, rows, hidden0, masks, actions, pack: DevicePackInfo, B, n, *, value=None,
                 log_prob=None, entropy=None, prev_actions=None, extra=None):
        self._fresh()
        o = self._obs(rgb, depth, goal, prev_actions, extra)
        self._call(lambda: self.L.hab_policy_evaluate(self.h, C.byref(o), ptr(rows), ptr(hidden0), self.Lh * self.hidden, ptr(masks),
                                                      ptr(actions), C.byref(pack.struct), B, n, ptr(value), ptr(log_prob), ptr(entropy),
                                                      stream_ptr()), "hab_policy_evaluate")

    def final_hidden(self, out):
        check(self.L.hab_policy_final_hidden(self.h, ptr(out), stream_ptr()), "hab_policy_final_hidden")

    def backward(self, rgb, depth, goal, rows, actions, pack: DevicePackInfo, d_value, d_log_prob, d_entropy, prev_actions=None,
                 extra=None):
        o = self._obs(rgb, depth, goal, prev_actions, extra)
        self._call(lambda: self.L.hab_policy_backward(self.h, C.byref(o), ptr(rows), ptr(actions), C.byref(pack.struct), ptr(d_value),
                                                      ptr(d_log_prob), ptr(d_entropy), stream_ptr()), "hab_policy_backward")

    def set_extra_grads(self, d_rnn_output, d_perception_embed):
        """Gradients wrt `rnn_output` / `perception_embed` ([B][hidden] fp32, frame order of the last evaluate; either may be None) from
        auxiliary losses: added inside the NEXT backward() where those tensors sit in its chain.  The caller keeps the tensors alive
        until that backward has been enqueued."""
        for t in (d_rnn_output, d_perception_embed):
            assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())
        check(self.L.hab_policy_set_extra_grads(self.h, ptr(d_rnn_output), ptr(d_perception_embed)), "hab_policy_set_extra_grads")

    def tap(self, which: int) -> torch.Tensor:
        p, n = C.c_void_p(), C.c_int64(0)
        check(self.L.hab_policy_tap(self.h, which, C.byref(p), C.byref(n)), "hab_policy_tap")
        off = (p.value - self.work.data_ptr()) // 4
        return self.work[off:off + n.value]

    def probe_enable(self, tag: int):
        check(self.L.hab_policy_probe_enable(self.h, tag))

    def probe_enable_mask(self, tags):
        mask = 0
        for t in tags:
            mask |= 1 << int(t)
        check(self.L.hab_policy_probe_enable_mask(self.h, mask))

    def probe_read_tag(self, tag: int):
        ms, cnt = C.c_double(0), C.c_int(0)
        check(self.L.hab_policy_probe_read_tag(self.h, int(tag), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def probe_work(self, tag: int):
        """(FLOPs, algorithmic bytes) launched by the call sites of `tag` since the probe was enabled (tags that report it)."""
        fl, by = C.c_double(0.0), C.c_double(0.0)
        check(self.L.hab_policy_probe_work(self.h, int(tag), C.byref(fl), C.byref(by)))
        return fl.value, by.value

    def probe_read(self):
        ms, cnt = C.c_double(0), C.c_int(0)
        check(self.L.hab_policy_probe_read(self.h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


class NativeComm:
    """RCCL communicator owned by libhabitat_amd (csrc/comm.hip).  The 128-byte unique id is created by rank 0 (`unique_id()`) and
    handed to every rank as `ident`; for compatibility `exchange(list_of_one_object)` (torch.distributed.broadcast_object_list for a
    process group, identity for one rank) may carry it instead."""

    @staticmethod
    def unique_id() -> bytes:
        L = _lib.lib()
        if not L.hab_comm_available():
            raise _lib.HabError("librccl was not found: no device-side exchange")
        buf = (C.c_uint8 * 128)()
        check(L.hab_comm_unique_id(buf), "hab_comm_unique_id")
        return bytes(buf)

    def __init__(self, world: int, rank: int, exchange=None, ident: bytes = None):
        L = _lib.lib()
        if not L.hab_comm_available():
            raise _lib.HabError("librccl was not found: no device-side exchange")
        if ident is None:
            box = [None]
            if rank == 0:
                box[0] = NativeComm.unique_id()
            if exchange is not None:
                exchange(box)
            ident = box[0]
        raw = (C.c_uint8 * 128).from_buffer_copy(ident)
        h = C.c_void_p()
        check(L.hab_comm_create(raw, int(world), int(rank), C.byref(h)), "hab_comm_create")
        self.L, self.h, self.world, self.rank = L, h, int(world), int(rank)

    def world_size(self) -> int:
        """Ranks of the communicator as the library sees them (hab_comm_world_size)."""
        return int(self.L.hab_comm_world_size(self.h))

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        check(self.L.hab_comm_allreduce_sum(self.h, ptr(t), t.numel(), stream_ptr()), "hab_comm_allreduce_sum")
        return t

    def close(self):
        if self.h is not None:
            self.L.hab_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass
