from habitat_amd.rl.ver.ver_rollout_storage import VERRolloutStorage  # noqa: F401
