from habitat_amd.rl.ppo.policy import NetPolicy, PointNavBaselinePolicy, Policy, PolicyActionData  # noqa: F401
from habitat_amd.rl.ppo.ppo import PPO  # noqa: F401
