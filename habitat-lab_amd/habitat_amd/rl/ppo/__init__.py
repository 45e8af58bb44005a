from habitat_amd.rl.ppo.policy import (NetPolicy, PointNavBaselinePolicy, PointNavResNetPolicy, Policy,  # noqa: F401
                                       PolicyActionData)
from habitat_amd.rl.ppo.ppo import PPO  # noqa: F401
from habitat_amd.rl.ppo.cpc_aux_loss import CPCA  # noqa: F401  (registers `cpca`, as the reference's rl/ppo/__init__.py:7 does)
