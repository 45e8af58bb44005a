"""Same import path as habitat_baselines.rl.ddppo.policy (resnet_policy.py:50)."""
from habitat_amd.rl.ppo.policy import PointNavResNetPolicy  # noqa: F401
