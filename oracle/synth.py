"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the synthetic PointNav environment source.

The reference obtains observations from habitat-sim through ``VectorEnv``
(habitat-lab/habitat/core/vector_env.py:402-410); BASELINE.json's metric is quoted on
*synthetic* observations of the same dtypes/layouts (habitat_simulator.py:107-227: rgb uint8
(H,W,3), depth float32 (H,W,1) in [0,1), pointgoal_with_gps_compass float32 (2,)).
This file defines that generator as a counter-based integer hash so that the HIP generator
(`habitat-lab_amd/csrc/synth.hip`) and this numpy version are bit-identical:

  word(seed, sensor, env, t, i) = mix(mix(mix(mix(seed + GOLD*(sensor+1)) ^ env) ^ t) ^ i)
  mix = the 'lowbias32' 32-bit finaliser.

  rgb   : bytes of word i (little endian), i over H*W*3/4 words
  depth : (word >> 8) * 2^-24
  goal  : rho = u0*10, phi = (u1*2-1)*pi          (fp32, one rounding each)
  reward: ((u0+u1)+(u2+u3) - 2) * sqrt(3)         (Irwin-Hall(4), unit variance; fp32)
  done  : word0 < 2^32/25  or  steps_since_reset+1 >= 500
"""
from __future__ import annotations

import numpy as np

GOLD = np.uint32(0x9E3779B9)
SENSOR_RGB, SENSOR_DEPTH, SENSOR_GOAL, SENSOR_REWARD, SENSOR_DONE = 0, 1, 2, 3, 4
SENSOR_SEMANTIC, SENSOR_OBJECTGOAL, SENSOR_COMPASS, SENSOR_GPS = 5, 6, 7, 8
NUM_SEMANTIC_IDS, NUM_OBJECT_CATEGORIES = 40, 21  # semantic ids in [0,40), objectgoal in [0,21)
DONE_THRESHOLD = np.uint32((1 << 32) // 25)
MAX_EPISODE_STEPS = 500


def mix(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def stream_key(seed, sensor, env, t):
    with np.errstate(over="ignore"):
        h = mix(np.uint32(seed) + GOLD * np.uint32(sensor + 1))
        h = mix(h ^ np.asarray(env, dtype=np.uint32))
        h = mix(h ^ np.asarray(t, dtype=np.uint32))
    return h


def words(seed, sensor, env, t, n):
    key = stream_key(seed, sensor, env, t)
    return mix(key ^ np.arange(n, dtype=np.uint32))


def u01(w):
    return (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def rgb(seed, env, t, H, W):
    w = words(seed, SENSOR_RGB, env, t, (H * W * 3 + 3) // 4)
    return w.view(np.uint8)[: H * W * 3].reshape(H, W, 3).copy()


def depth(seed, env, t, H, W):
    return u01(words(seed, SENSOR_DEPTH, env, t, H * W)).reshape(H, W, 1)


def goal(seed, env, t):
    u = u01(words(seed, SENSOR_GOAL, env, t, 2))
    rho = u[0] * np.float32(10.0)
    phi = (u[1] * np.float32(2.0) - np.float32(1.0)) * np.float32(np.pi)
    return np.array([rho, phi], dtype=np.float32)


def semantic(seed, env, t, H, W):
    """int32 (H,W,1) instance/category ids, uniform in [0, 40)."""
    return (words(seed, SENSOR_SEMANTIC, env, t, H * W) % np.uint32(NUM_SEMANTIC_IDS)).astype(np.int32).reshape(H, W, 1)


def objectgoal(seed, env, t):
    """int64 (1,) goal category, constant over an env's life (drawn from the env id only)."""
    return np.array([int(words(seed, SENSOR_OBJECTGOAL, env, 0, 1)[0] % np.uint32(NUM_OBJECT_CATEGORIES))], dtype=np.int64)


def compass(seed, env, t):
    u = u01(words(seed, SENSOR_COMPASS, env, t, 1))
    return ((u * np.float32(2.0) - np.float32(1.0)) * np.float32(np.pi)).astype(np.float32)


def gps(seed, env, t):
    u = u01(words(seed, SENSOR_GPS, env, t, 8))
    a = (u[0] + u[1]) + (u[2] + u[3])
    b = (u[4] + u[5]) + (u[6] + u[7])
    return np.array([(a - np.float32(2.0)) * np.float32(np.sqrt(3.0)), (b - np.float32(2.0)) * np.float32(np.sqrt(3.0))], dtype=np.float32)


def reward(seed, env, t):
    u = u01(words(seed, SENSOR_REWARD, env, t, 4))
    s = (u[0] + u[1]) + (u[2] + u[3])
    return np.float32((s - np.float32(2.0)) * np.float32(np.sqrt(3.0)))


def done_draw(seed, env, t):
    return bool(words(seed, SENSOR_DONE, env, t, 1)[0] < DONE_THRESHOLD)


class SyntheticEnvs:
    """N independent synthetic envs; `env_offset` makes ids globally unique across ranks
    (mirrors seed += rank * num_environments, rl/ppo/ppo_trainer.py:208-211)."""

    def __init__(self, num_envs, H, W, seed=100, env_offset=0, use_rgb=True, use_depth=True, task="pointnav"):
        self.N, self.H, self.W, self.seed, self.off = num_envs, H, W, seed, env_offset
        self.use_rgb, self.use_depth, self.task = use_rgb, use_depth, task
        self.t = np.zeros(num_envs, dtype=np.int64)
        self.since = np.zeros(num_envs, dtype=np.int64)

    def _obs(self):
        o = {}
        if self.use_rgb:
            o["rgb"] = np.stack([rgb(self.seed, self.off + n, self.t[n], self.H, self.W) for n in range(self.N)])
        if self.use_depth:
            o["depth"] = np.stack([depth(self.seed, self.off + n, self.t[n], self.H, self.W) for n in range(self.N)])
        if self.task == "objectnav":  # rgb, depth, semantic + objectgoal, compass, gps (ddppo_objectnav.yaml sensor set)
            o["semantic"] = np.stack([semantic(self.seed, self.off + n, self.t[n], self.H, self.W) for n in range(self.N)])
            o["objectgoal"] = np.stack([objectgoal(self.seed, self.off + n, self.t[n]) for n in range(self.N)])
            o["compass"] = np.stack([compass(self.seed, self.off + n, self.t[n]) for n in range(self.N)])
            o["gps"] = np.stack([gps(self.seed, self.off + n, self.t[n]) for n in range(self.N)])
            return o
        o["pointgoal_with_gps_compass"] = np.stack([goal(self.seed, self.off + n, self.t[n]) for n in range(self.N)])
        return o

    def reset(self):
        self.t[:] = 0
        self.since[:] = 0
        return self._obs()

    def step(self, actions=None):
        self.t += 1
        rew = np.array([reward(self.seed, self.off + n, self.t[n]) for n in range(self.N)], dtype=np.float32)
        dn = np.zeros(self.N, dtype=bool)
        for n in range(self.N):
            self.since[n] += 1
            d = done_draw(self.seed, self.off + n, self.t[n]) or self.since[n] >= MAX_EPISODE_STEPS
            if d:
                self.since[n] = 0
            dn[n] = d
        return self._obs(), rew, dn
