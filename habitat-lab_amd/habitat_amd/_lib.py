"""ctypes binding of libhabitat_amd.so (the C-ABI declared in include/habitat_amd.h).

The library is built in-tree (``make -C habitat-lab_amd/csrc`` or ``__graft_entry__.build()``).
There is NO fallback: if the shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhabitat_amd.so")

vp = c_void_p
ALLREDUCE_FN = C.CFUNCTYPE(None, c_void_p, c_int, c_float, c_void_p)  # hab_allreduce_fn
GRAD_READY_FN = C.CFUNCTYPE(None, c_int64, c_int64, c_void_p)       # hab_grad_ready_fn


class HabError(RuntimeError):
    pass


class PolicyDesc(C.Structure):
    _fields_ = [(n, c_int32) for n in (
        "arch", "backbone", "baseplanes", "normalize_visual_inputs", "rnn_type", "rnn_layers", "hidden", "num_actions",
        "H", "W", "has_rgb", "has_depth", "goal_dim", "max_frames", "max_envs", "visual_order", "has_semantic",
        "num_object_categories", "has_compass", "has_gps", "action_dist", "gauss_flags")] + [("gauss_min_std", c_float),
                                                                                                ("gauss_max_std", c_float),
                                                                                                ("pointgoal_dim", c_int32),
                                                                                                ("proximity_dim", c_int32)]


GAUSS_TANH_MU, GAUSS_USE_LOG_STD, GAUSS_USE_SOFTPLUS, GAUSS_USE_STD_PARAM, GAUSS_CLAMP_STD = 1, 2, 4, 8, 16  # HAB_GAUSS_*


class Obs(C.Structure):
    _fields_ = [("rgb", vp), ("depth", vp), ("goal", vp), ("prev_actions", vp), ("semantic", vp), ("objectgoal", vp), ("compass", vp),
                ("gps", vp), ("visual_features", vp), ("pointgoal", vp), ("proximity", vp)]


class EmbedSlot(C.Structure):
    _fields_ = [("kind", c_int32), ("input", vp), ("weight", vp), ("bias", vp), ("d_weight", vp), ("d_bias", vp), ("num_tokens", c_int32)]


class PackInfo(C.Structure):
    _fields_ = [("select_inds", vp), ("frag_env", vp), ("frag_start", vp), ("step_offsets_host", vp),
                ("num_seqs_at_step_host", vp), ("P", c_int32), ("F", c_int32), ("max_len", c_int32), ("env_first_frame", vp)]


# name -> (restype, argtypes).  Every symbol of include/habitat_amd.h appears here; tests/test_capi.py
# checks that the header, this table and the shared object agree.
SIGNATURES = {
    "hab_abi_version": (c_int, []),
    "hab_error_string": (c_char_p, [c_int]),
    "hab_synth_step": (c_int, [vp, vp, vp, vp, vp, vp, vp, c_uint32, c_uint32, c_int, c_int, c_int, c_int, vp]),
    "hab_synth_objectnav_sensors": (c_int, [vp, vp, vp, vp, vp, c_uint32, c_uint32, c_int, c_int, c_int, vp]),
    "hab_obs_resize_crop": (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    "hab_rollout_step_stats": (c_int, [vp, vp, vp, vp, vp, vp, vp, c_int, c_int, vp]),
    "hab_compute_returns": (c_int, [vp, vp, vp, vp, vp, c_int, c_int, c_float, c_float, c_int, c_int, vp]),
    "hab_advantages": (c_int, [vp, vp, vp, c_int, c_int, vp, vp, vp]),
    "hab_ppo_loss": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, c_int, c_float, c_float, c_float, c_int, vp, vp, vp, vp, vp]),
    "hab_ppo_loss_ver": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, c_int, c_float, c_float, c_float, c_int, vp, vp, vp, c_int64, vp, c_float,
                                 vp, vp, vp, vp, vp]),
    "hab_lagrange_adam_step": (c_int, [vp, vp, vp, vp, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_float, vp, vp]),
    "hab_ver_compute_returns": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, c_int, c_double, c_double, vp]),
    "hab_ver_is_coeffs": (c_int, [vp, c_int, c_int, c_int, vp, vp, vp]),
    "hab_clip_adam_step": (c_int, [vp, vp, vp, vp, c_size_t, vp, c_int, c_float, c_float, c_float, c_float, c_float,
                                   c_float, c_int, vp, vp]),
    "hab_sample_actions": (c_int, [vp, vp, vp, c_int, c_int, c_int, vp]),
    "hab_set_matrix_path": (c_int, [c_int]),
    "hab_conv2d_fwd": (c_int, [vp, vp, vp, vp] + [c_int] * 10 + [vp, c_size_t, vp]),
    "hab_obs_conv2d_fwd": (c_int, [vp, vp, vp, vp, vp, vp] + [c_int] * 9 + [vp, c_size_t, vp]),
    "hab_conv2d_dgrad": (c_int, [vp, vp, vp, vp, vp] + [c_int] * 9 + [vp, c_size_t, vp]),
    "hab_conv2d_wgrad": (c_int, [vp, vp, vp, vp] + [c_int] * 9 + [vp, c_size_t, vp]),
    "hab_obs_conv2d_wgrad": (c_int, [vp, vp, vp, vp, vp, vp] + [c_int] * 8 + [vp, c_size_t, vp]),
    "hab_linear_fwd": (c_int, [vp, c_int, vp, c_int, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, c_size_t, vp]),
    "hab_linear_dgrad": (c_int, [vp, c_int, vp, c_int, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, vp, c_size_t, vp]),
    "hab_linear_wgrad": (c_int, [vp, c_int, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp, c_size_t, vp]),
    "hab_colsum": (c_int, [vp, c_int, c_int, c_int, vp, c_int, vp, c_size_t, vp]),
    "hab_repack_conv_weight": (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp]),
    "hab_repack_flatten_weight": (c_int, [vp, vp, c_int, c_int, c_int, vp]),
    "hab_transpose2d": (c_int, [vp, vp, c_int, c_int, vp]),
    "hab_obs_ingest_pool": (c_int, [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    "hab_channel_moments": (c_int, [vp, c_int64, c_int, c_int, vp, vp, vp, c_int, vp]),
    "hab_running_mean_var_update": (c_int, [vp, vp, vp, vp, vp, c_float, c_int, vp]),
    "hab_running_mean_var_normalize": (c_int, [vp, c_int64, c_int, c_int, vp, vp, vp]),
    "hab_groupnorm_fwd": (c_int, [vp, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_float, vp, c_int64, vp]),
    "hab_groupnorm_bwd": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp, c_int64, vp]),
    "hab_split_weight_planes": (c_int, [vp, c_int, c_int, vp, vp]),
    "hab_stem_split_weights": (c_int, [vp, vp, vp]),
    "hab_stem_conv_wgrad": (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, vp, c_size_t, vp, vp]),
    "hab_stem_conv_fwd": (c_int, [vp, vp, vp, c_int, c_int, c_int, vp, vp, c_int, vp]),
    "hab_conv_gn_fwd": (c_int, [vp] * 9 + [c_int] * 11 + [c_float, vp]),
    "hab_maxpool3x3s2_fwd": (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, vp]),
    "hab_maxpool3x3s2_bwd": (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, vp]),
    "hab_nav_embed_fwd": (c_int, [POINTER(EmbedSlot), c_int, vp, vp, vp, c_int, c_int, c_int, vp, vp]),
    "hab_nav_embed_bwd": (c_int, [POINTER(EmbedSlot), c_int, vp, vp, c_int, c_int, c_int, vp, c_size_t, vp]),
    "hab_build_pack_info": (c_int, [vp, c_int, c_int] + [vp] * 12),
    "hab_build_pack_info_from_ids": (c_int, [vp, vp, vp, c_int] + [vp] * 11),
    "hab_policy_create": (c_int, [POINTER(PolicyDesc), POINTER(vp)]),
    "hab_policy_destroy": (None, [vp]),
    "hab_policy_num_params": (c_int, [vp]),
    "hab_policy_param_info": (c_int, [vp, c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int), POINTER(c_int64)]),
    "hab_policy_param_is_buffer": (c_int, [vp, c_int]),
    "hab_policy_set_training": (c_int, [vp, c_int]),
    "hab_comm_available": (c_int, []),
    "hab_comm_unique_id": (c_int, [vp]),
    "hab_comm_create": (c_int, [vp, c_int, c_int, vp]),
    "hab_comm_destroy": (None, [vp]),
    "hab_comm_world_size": (c_int, [vp]),
    "hab_comm_allreduce_sum": (c_int, [vp, vp, c_int64, vp]),
    "hab_policy_set_comm": (c_int, [vp, vp]),
    "hab_policy_grad_sync": (c_int, [vp, vp]),
    "hab_policy_set_grad_ready": (c_int, [vp, GRAD_READY_FN, vp]),
    "hab_policy_set_allreduce": (c_int, [vp, ALLREDUCE_FN, vp, c_int]),
    "hab_policy_param_floats": (c_int64, [vp]),
    "hab_policy_packed_floats": (c_int64, [vp]),
    "hab_policy_work_floats": (c_int64, [vp]),
    "hab_policy_bind": (c_int, [vp, vp, vp, vp, vp, c_int64]),
    "hab_policy_repack": (c_int, [vp, vp]),
    "hab_policy_encode": (c_int, [vp, POINTER(Obs), c_int, vp, vp]),
    "hab_policy_visual_feature_shape": (c_int, [vp, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "hab_policy_act": (c_int, [vp, POINTER(Obs), vp, vp, vp, c_int, c_int, vp, vp, vp, vp, vp, vp]),
    "hab_policy_evaluate": (c_int, [vp, POINTER(Obs), vp, vp, c_int, vp, vp, POINTER(PackInfo), c_int, c_int, vp, vp, vp, vp]),
    "hab_policy_final_hidden": (c_int, [vp, vp, vp]),
    "hab_policy_set_extra_grads": (c_int, [vp, vp, vp]),
    "hab_policy_backward": (c_int, [vp, POINTER(Obs), vp, vp, POINTER(PackInfo), vp, vp, vp, vp]),
    "hab_policy_probe_enable": (c_int, [vp, c_int]),
    "hab_policy_probe_read": (c_int, [vp, POINTER(c_double), POINTER(c_int)]),
    "hab_policy_probe_enable_mask": (c_int, [vp, C.c_uint64]),
    "hab_policy_probe_read_tag": (c_int, [vp, c_int, POINTER(c_double), POINTER(c_int)]),
    "hab_policy_probe_work": (c_int, [vp, c_int, POINTER(c_double), POINTER(c_double)]),
    "hab_policy_tap": (c_int, [vp, c_int, POINTER(vp), POINTER(c_int64)]),
}

_lib = None


def lib():
    """Loads the shared object (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HabError(
                f"{LIB_PATH} is missing: build it with `make -C habitat-lab_amd/csrc` (or __graft_entry__.build()). "
                "habitat_amd has no CPU / PyTorch fallback for its kernels.")
        # PyTorch-ROCm wheels bundle their own HIP / HSA runtime.  It must be in the process BEFORE this library is mapped, so that
        # the library's libamdhip64 dependency resolves to the runtime torch uses: two runtimes in one process cannot share
        # streams or allocations (the second one fails with hipErrorNoDevice on the first launch).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int, what: str = ""):
    if code != 0:
        msg = lib().hab_error_string(code)
        raise HabError(f"{what or 'habitat_amd call'} failed: [{code}] {msg.decode() if msg else '?'}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
