"""habitat_baselines.rl.ddppo.policy.resnet_policy.PointNavResNetPolicy, engine-backed (see habitat_amd/rl/ppo/policy.py)."""
from habitat_amd.rl.ppo.policy import PointNavResNetPolicy  # noqa: F401
