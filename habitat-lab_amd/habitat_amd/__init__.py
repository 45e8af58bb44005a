"""habitat_amd -- MI355X-native drop-in for the habitat_baselines PPO / DD-PPO training path.

Registry names, class names, method signatures and config keys follow habitat_baselines
(trainer "ppo"/"ddppo", policies "PointNavBaselinePolicy"/"PointNavResNetPolicy", updaters
"PPO"/"DDPPO", storage "RolloutStorage"); the arithmetic runs in libhabitat_amd.so (hand-written
HIP for gfx950, include/habitat_amd.h).  There is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
