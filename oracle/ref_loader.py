"""TEST INFRASTRUCTURE ONLY -- loads the *unmodified* reference hot-path files.

`/root/reference` is a Python code base whose package `habitat_baselines` cannot
be imported here (hydra / gym / habitat-sim are not installed).  The hot-path
files themselves are pure torch/numpy, so this module installs a handful of
stub modules (gym.spaces, habitat.*, torchvision.transforms, cv2, ...) and then
loads the reference *source files where they lie* with
``importlib.util.spec_from_file_location``.  Nothing is copied into this repo.

Used by: ``tests/golden/make_golden.py`` (fixture generation) and the
``test_oracle_vs_reference_live`` tests (skipped when ``/root/reference`` is
absent, e.g. on the GPU box).  The product package never imports this file.
"""
from __future__ import annotations

import collections
import importlib.util
import logging
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("HABITAT_REFERENCE_ROOT", "/root/reference")
HB = os.path.join(REFERENCE_ROOT, "habitat-baselines", "habitat_baselines")

_LOADED = {}


def reference_available() -> bool:
    return os.path.isdir(HB)


# ----------------------------------------------------------------------------
# gym.spaces stub: only what policy.py / rollout_storage.py / resnet_policy.py
# touch (shape, dtype, high/low, n, .spaces, iteration helpers).
# ----------------------------------------------------------------------------
class _Space:
    shape = None
    dtype = None


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)


class Discrete(_Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)


class MultiDiscrete(_Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)


class Dict(_Space):
    def __init__(self, spaces=None):
        self.spaces = collections.OrderedDict(spaces or {})

    def items(self):
        return self.spaces.items()

    def keys(self):
        return self.spaces.keys()

    def values(self):
        return self.spaces.values()

    def __getitem__(self, k):
        return self.spaces[k]

    def __contains__(self, k):
        return k in self.spaces

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    if "gym" not in sys.modules:
        spaces = _mod(
            "gym.spaces",
            Box=Box,
            Discrete=Discrete,
            MultiDiscrete=MultiDiscrete,
            Dict=Dict,
            Space=_Space,
        )
        _mod("gym", spaces=spaces)

    class _Logger(logging.Logger):
        def add_filehandler(self, *_a, **_k):
            pass

    logger = _Logger("habitat-ref-stub", level=logging.WARNING)

    class Singleton(type):
        _instances = {}

        def __call__(cls, *a, **k):
            if cls not in cls._instances:
                cls._instances[cls] = super().__call__(*a, **k)
            return cls._instances[cls]

    class Registry(metaclass=Singleton):
        # Semantics of habitat-lab/habitat/core/registry.py:43-69 (name -> class map).
        mapping = collections.defaultdict(dict)

        @classmethod
        def _register_impl(cls, _type, to_register, name, assert_type=None):
            def wrap(obj):
                if assert_type is not None:
                    assert issubclass(obj, assert_type)
                cls.mapping[_type][obj.__name__ if name is None else name] = obj
                return obj

            return wrap if to_register is None else wrap(to_register)

        @classmethod
        def _get_impl(cls, _type, name):
            return cls.mapping[_type].get(name, None)

    class EmptySpace(_Space):
        pass

    def _uuid_cls(uuid):
        return type("Sensor_" + uuid, (), {"cls_uuid": uuid})

    prof = _mod(
        "habitat.utils.profiling_wrapper",
        range_push=lambda *a, **k: None,
        range_pop=lambda *a, **k: None,
        on_start_step=lambda *a, **k: None,
        configure=lambda *a, **k: None,
    )

    class RangeContext:
        def __init__(self, *_a, **_k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def __call__(self, f):
            return f

    prof.RangeContext = RangeContext
    hutils = _mod("habitat.utils", profiling_wrapper=prof)
    _mod("habitat.utils.visualizations")
    _mod("habitat.utils.visualizations.utils", images_to_video=None)
    _mod("habitat.core")
    _mod("habitat.core.registry", Registry=Registry)
    _mod("habitat.core.dataset", Episode=object)
    _mod("habitat.core.spaces", EmptySpace=EmptySpace)
    _mod("habitat.core.utils", Singleton=Singleton)
    _mod("habitat.tasks")
    _mod("habitat.tasks.nav")
    _mod(
        "habitat.tasks.nav.nav",
        PointGoalSensor=_uuid_cls("pointgoal"),
        ImageGoalSensor=_uuid_cls("imagegoal"),
        IntegratedPointGoalGPSAndCompassSensor=_uuid_cls("pointgoal_with_gps_compass"),
        HeadingSensor=_uuid_cls("heading"),
        EpisodicCompassSensor=_uuid_cls("compass"),
        EpisodicGPSSensor=_uuid_cls("gps"),
        ProximitySensor=_uuid_cls("proximity"),
    )
    _mod("habitat.tasks.nav.object_nav_task", ObjectGoalSensor=_uuid_cls("objectgoal"))
    _mod(
        "habitat.tasks.nav.instance_image_nav_task",
        InstanceImageGoalSensor=_uuid_cls("instance_imagegoal"),
    )
    _mod("habitat", logger=logger, utils=hutils)
    if "cv2" not in sys.modules:
        _mod("cv2")
    if "torchvision" not in sys.modules:
        tf = _mod("torchvision.transforms.functional")
        tr = _mod("torchvision.transforms", functional=tf)
        _mod("torchvision", transforms=tr)


def _load(modname, relpath):
    if modname in sys.modules and getattr(sys.modules[modname], "__file__", None):
        return sys.modules[modname]
    path = os.path.join(HB, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Returns a namespace of the reference hot-path modules, loaded in place."""
    if _LOADED:
        return _LOADED["ns"]
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REFERENCE_ROOT}")
    _install_stubs()
    # parent packages as empty shells so that the `from habitat_baselines.x import y` lines resolve
    for pkg in [
        "habitat_baselines",
        "habitat_baselines.common",
        "habitat_baselines.utils",
        "habitat_baselines.rl",
        "habitat_baselines.rl.models",
        "habitat_baselines.rl.ppo",
        "habitat_baselines.rl.ddppo",
        "habitat_baselines.rl.ddppo.policy",
        "habitat_baselines.rl.ddppo.algo",
        "habitat_baselines.rl.ver",
    ]:
        if pkg not in sys.modules:
            p = _mod(pkg)
            p.__path__ = []  # mark as package
    _mod("habitat_baselines.rl.ver.ver_rollout_storage", VERRolloutStorage=type("VERRolloutStorage", (), {}))
    _mod("habitat_baselines.common.tensorboard_utils", TensorboardWriter=object)
    ns = types.SimpleNamespace()
    ns.windowed_running_mean = _load("habitat_baselines.common.windowed_running_mean", "common/windowed_running_mean.py")
    ns.timing = _load("habitat_baselines.utils.timing", "utils/timing.py")
    ns.tensor_dict = _load("habitat_baselines.common.tensor_dict", "common/tensor_dict.py")
    ns.baseline_registry = _load("habitat_baselines.common.baseline_registry", "common/baseline_registry.py")
    ns.storage = _load("habitat_baselines.common.storage", "common/storage.py")
    ns.common = _load("habitat_baselines.utils.common", "utils/common.py")
    ns.rnn_state_encoder = _load("habitat_baselines.rl.models.rnn_state_encoder", "rl/models/rnn_state_encoder.py")
    ns.simple_cnn = _load("habitat_baselines.rl.models.simple_cnn", "rl/models/simple_cnn.py")
    ns.rollout_storage = _load("habitat_baselines.common.rollout_storage", "common/rollout_storage.py")
    ns.updater = _load("habitat_baselines.rl.ppo.updater", "rl/ppo/updater.py")
    ns.policy = _load("habitat_baselines.rl.ppo.policy", "rl/ppo/policy.py")
    ns.ppo = _load("habitat_baselines.rl.ppo.ppo", "rl/ppo/ppo.py")
    # `from habitat_baselines.rl.ppo import Net, NetPolicy, PPO` (resnet_policy.py:38, ddppo.py:14)
    pkg = sys.modules["habitat_baselines.rl.ppo"]
    pkg.Net, pkg.NetPolicy, pkg.PPO = ns.policy.Net, ns.policy.NetPolicy, ns.ppo.PPO
    ns.running_mean_and_var = _load(
        "habitat_baselines.rl.ddppo.policy.running_mean_and_var", "rl/ddppo/policy/running_mean_and_var.py"
    )
    ns.resnet = _load("habitat_baselines.rl.ddppo.policy.resnet", "rl/ddppo/policy/resnet.py")
    sys.modules["habitat_baselines.rl.ddppo.policy"].resnet = ns.resnet
    ns.resnet_policy = _load("habitat_baselines.rl.ddppo.policy.resnet_policy", "rl/ddppo/policy/resnet_policy.py")
    ns.ddppo = _load("habitat_baselines.rl.ddppo.algo.ddppo", "rl/ddppo/algo/ddppo.py")
    ns.spaces = sys.modules["gym.spaces"]
    _LOADED["ns"] = ns
    return ns


def load_reference_ver():
    """Adds the reference's VER pieces to the namespace: the real VERRolloutStorage (replacing the placeholder
    class rl/ppo/ppo.py was given to import) and the real InferenceWorkerProcess, so that tests/golden/make_golden.py can drive the
    reference's own `step()` / storage writes.  Stubbed: the batched queue (the reference's fallback class subclasses
    torch.multiprocessing.Queue, which is a bound method in this torch) and the obs-transformer helpers (identity: none configured)."""
    ns = load_reference()
    if getattr(ns, "inference_worker", None) is not None:
        return ns
    for name in ("habitat_baselines.rl.ver.ver_rollout_storage",):
        sys.modules.pop(name, None)
    ns.ver_rollout_storage = _load("habitat_baselines.rl.ver.ver_rollout_storage", "rl/ver/ver_rollout_storage.py")
    ns.ppo.VERRolloutStorage = ns.ver_rollout_storage.VERRolloutStorage  # the isinstance check of ppo.py:292

    class BatchedQueue:
        def __init__(self, *_a, **_k):
            self.items = []

        def put(self, x, *_a, **_k):
            self.items.append(x)

        def put_many(self, xs, *_a, **_k):
            self.items.extend(xs)

        def get_many(self, *_a, **_k):
            out, self.items = self.items, []
            if not out:
                import queue
                raise queue.Empty
            return out

        def empty(self):
            return not self.items

    _mod("habitat_baselines.rl.ver.queue", BatchedQueue=BatchedQueue)
    _mod("habitat_baselines.common.obs_transformers", apply_obs_transforms_batch=lambda obs, _t: obs, get_active_obs_transforms=lambda _c: [])
    ns.task_enums = _load("habitat_baselines.rl.ver.task_enums", "rl/ver/task_enums.py")
    ns.worker_common = _load("habitat_baselines.rl.ver.worker_common", "rl/ver/worker_common.py")
    ns.inference_worker = _load("habitat_baselines.rl.ver.inference_worker", "rl/ver/inference_worker.py")
    ns.BatchedQueue = BatchedQueue
    # the straggler-preemption schedule (its arithmetic is plain numpy; the process / rendezvous machinery around it is not used)
    if "habitat_baselines.rl.ddppo.ddp_utils" not in sys.modules:
        _mod("habitat_baselines.rl.ddppo.ddp_utils", init_distrib_slurm=lambda *a, **k: None, rank0_only=lambda: True)
    ns.preemption_decider = _load("habitat_baselines.rl.ver.preemption_decider", "rl/ver/preemption_decider.py")
    return ns


def load_reference_aux():
    """Adds the reference's auxiliary loss (rl/ppo/cpc_aux_loss.py) and the action embeddings it is built on
    (rl/models/action_embedding.py) to the namespace.  habitat.core.spaces.ActionSpace (habitat-lab/habitat/core/spaces.py:33-57: a Dict of
    per-action argument spaces, sorted by name, `n` = number of actions) is stubbed with those semantics."""
    ns = load_reference()
    if getattr(ns, "cpc_aux_loss", None) is not None:
        return ns
    hs = sys.modules["habitat.core.spaces"]

    class ActionSpace(Dict):
        def __init__(self, spaces):
            super().__init__(sorted(spaces.items()) if isinstance(spaces, dict) else spaces)

        @property
        def n(self):
            return len(self.spaces)

    hs.ActionSpace = ActionSpace
    ns.ActionSpace, ns.EmptySpace = ActionSpace, hs.EmptySpace
    ns.action_embedding = _load("habitat_baselines.rl.models.action_embedding", "rl/models/action_embedding.py")
    ns.cpc_aux_loss = _load("habitat_baselines.rl.ppo.cpc_aux_loss", "rl/ppo/cpc_aux_loss.py")
    return ns


def make_config(**ppo_overrides):
    """A plain attribute-tree carrying the config keys the hot path reads
    (default_structured_configs.py:288-316,343-363)."""
    ppo = dict(
        clip_param=0.2, ppo_epoch=4, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.01,
        lr=2.5e-4, eps=1e-5, max_grad_norm=0.5, num_steps=5, use_gae=True, use_linear_lr_decay=False,
        use_linear_clip_decay=False, gamma=0.99, tau=0.95, reward_window_size=50,
        use_normalized_advantage=False, hidden_size=512, entropy_target_factor=0.0,
        use_adaptive_entropy_pen=False, use_clipped_value_loss=True, use_double_buffered_sampler=False,
    )
    ppo.update(ppo_overrides)
    return types.SimpleNamespace(**ppo)
