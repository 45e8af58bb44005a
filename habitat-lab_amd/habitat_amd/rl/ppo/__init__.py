from habitat_amd.rl.ppo.policy import (NetPolicy, PointNavBaselinePolicy, PointNavResNetPolicy, Policy,  # noqa: F401
                                       PolicyActionData)
from habitat_amd.rl.ppo.ppo import PPO  # noqa: F401
