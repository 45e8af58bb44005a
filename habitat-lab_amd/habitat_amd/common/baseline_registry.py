"""Plugin registry with the interface of habitat_baselines/common/baseline_registry.py:28-193
(on habitat-lab/habitat/core/registry.py:43-69): decorators register_{trainer,policy,updater,storage,
agent_access_mgr,obs_transformer,auxiliary_loss,env} and the matching get_* look-ups by name."""
from __future__ import annotations

import collections
from typing import Any, Callable, Optional


class _Registry:
    def __init__(self):
        self.mapping = collections.defaultdict(dict)

    def _register_impl(self, _type: str, to_register: Optional[Any], name: Optional[str], assert_type=None) -> Callable:
        def wrap(obj):
            if assert_type is not None:
                assert issubclass(obj, assert_type), f"{obj} must be a subclass of {assert_type}"
            self.mapping[_type][obj.__name__ if name is None else name] = obj
            return obj

        return wrap if to_register is None else wrap(to_register)

    def _get_impl(self, _type: str, name: str):
        return self.mapping[_type].get(name, None)

    # ---- decorators -------------------------------------------------------------------------
    def register_trainer(self, to_register=None, *, name: Optional[str] = None):
        from habitat_amd.common.base_trainer import BaseTrainer

        return self._register_impl("trainer", to_register, name, assert_type=BaseTrainer)

    def register_policy(self, to_register=None, *, name: Optional[str] = None):
        from habitat_amd.rl.ppo.policy import Policy

        return self._register_impl("policy", to_register, name, assert_type=Policy)

    def register_updater(self, to_register=None, *, name: Optional[str] = None):
        return self._register_impl("updater", to_register, name)

    def register_storage(self, to_register=None, *, name: Optional[str] = None):
        return self._register_impl("storage", to_register, name)

    def register_agent_access_mgr(self, to_register=None, *, name: Optional[str] = None):
        return self._register_impl("agent", to_register, name)

    def register_obs_transformer(self, to_register=None, *, name: Optional[str] = None):
        return self._register_impl("obs_transformer", to_register, name)

    def register_auxiliary_loss(self, to_register=None, *, name: Optional[str] = None):
        return self._register_impl("aux_loss", to_register, name)

    def register_env(self, to_register=None, *, name: Optional[str] = None):
        return self._register_impl("env", to_register, name)

    # ---- look-ups ---------------------------------------------------------------------------
    def get_trainer(self, name): return self._get_impl("trainer", name)
    def get_policy(self, name): return self._get_impl("policy", name)
    def get_updater(self, name): return self._get_impl("updater", name)
    def get_storage(self, name): return self._get_impl("storage", name)
    def get_agent_access_mgr(self, name): return self._get_impl("agent", name)
    def get_obs_transformer(self, name): return self._get_impl("obs_transformer", name)
    def get_auxiliary_loss(self, name): return self._get_impl("aux_loss", name)
    def get_env(self, name): return self._get_impl("env", name)


baseline_registry = _Registry()
