from habitat_amd.rl.ddppo.ddppo import DDPPO  # noqa: F401
