"""Pendulum-v1 as a built-in environment (the observation / action shapes of BASELINE.json configs[4]).

The reference reaches it through gymnasium (not installed here); this module realises the id with the same interface:
observation Box(3) float32 (cos th, sin th, th_dot), one continuous action (torque in [-2, 2]), reward
-(th^2 + 0.1 th_dot^2 + 0.001 u^2) with th normalised to [-pi, pi], no termination, `truncated` after 200 steps.
Dynamics: th_dot += (3 g / (2 l) sin th + 3 / (m l^2) u) dt with g = 10, m = l = 1, dt = 0.05, th_dot clipped to [-8, 8],
th += th_dot dt; start th ~ U[-pi, pi], th_dot ~ U[-1, 1].  The device twin is `srlx_pendulum_step` (csrc/srlx_ppo.hip)."""
import math
import random
from dataclasses import dataclass
from typing import Any, Optional, Tuple

import numpy as np

from simple_distributed_rl_amd.base.env import registration
from simple_distributed_rl_amd.base.env.base import EnvBase
from simple_distributed_rl_amd.base.spaces.box import BoxSpace

registration.register("Pendulum-v1", __name__ + ":Pendulum", {}, check_duplicate=False)

MAX_SPEED, MAX_TORQUE, DT, G, M, L = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0


@dataclass
class Pendulum(EnvBase):
    max_steps: int = 200

    def __post_init__(self):
        super().__init__()
        self.th, self.thdot, self.steps = 0.0, 0.0, 0

    @property
    def action_space(self) -> BoxSpace:
        return BoxSpace((1,), -MAX_TORQUE, MAX_TORQUE, np.float32)

    @property
    def observation_space(self) -> BoxSpace:
        high = np.array([1.0, 1.0, MAX_SPEED], np.float32)
        return BoxSpace((3,), -high, high, np.float32)

    @property
    def player_num(self) -> int:
        return 1

    @property
    def max_episode_steps(self) -> int:
        return self.max_steps

    @property
    def reward_range(self) -> Tuple[float, float]:
        return -(math.pi ** 2 + 0.1 * MAX_SPEED ** 2 + 0.001 * MAX_TORQUE ** 2) * self.max_steps, 0.0

    def _obs(self) -> np.ndarray:
        return np.array([math.cos(self.th), math.sin(self.th), self.thdot], np.float32)

    def reset(self, *, seed: Optional[int] = None, **kwargs) -> Any:
        if seed is not None:
            random.seed(seed)
        self.th, self.thdot, self.steps = random.uniform(-math.pi, math.pi), random.uniform(-1.0, 1.0), 0
        return self._obs()

    def step(self, action) -> Tuple[Any, float, bool, bool]:
        u = float(np.clip(np.asarray(action, np.float64).reshape(-1)[0], -MAX_TORQUE, MAX_TORQUE))
        ang = ((self.th + math.pi) % (2 * math.pi)) - math.pi
        reward = -(ang * ang + 0.1 * self.thdot * self.thdot + 0.001 * u * u)
        self.thdot = float(np.clip(self.thdot + (3 * G / (2 * L) * math.sin(self.th) + 3.0 / (M * L * L) * u) * DT, -MAX_SPEED, MAX_SPEED))
        self.th = self.th + self.thdot * DT
        self.steps += 1
        return self._obs(), reward, False, self.steps >= self.max_steps

    def backup(self, **kwargs) -> Any:
        return (self.th, self.thdot, self.steps)

    def restore(self, data: Any, **kwargs) -> None:
        self.th, self.thdot, self.steps = data
