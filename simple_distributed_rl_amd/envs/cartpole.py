"""CartPole-v1 as a built-in environment (BASELINE.json configs[1]: "DQN on CartPole-v1, uniform replay 1e5").

The reference reaches CartPole through gymnasium (srl/base/env/gym_user_wrapper.py, `srl.Runner("CartPole-v1", ...)`),
which is not installed here, so the id is realised by this module with the same interface: observation Box(4) float32
[x, x_dot, theta, theta_dot], two discrete actions (push left / right), reward 1 per step, `terminated` when the pole
leaves +-12 degrees or the cart +-2.4, `truncated` after 500 steps.  Dynamics: the classic cart-pole equations of Barto,
Sutton & Anderson (1983) integrated with explicit Euler at 0.02 s, start state uniform in [-0.05, 0.05]^4."""
import math
import random
from dataclasses import dataclass
from typing import Any, Optional, Tuple

import numpy as np

from simple_distributed_rl_amd.base.env import registration
from simple_distributed_rl_amd.base.env.base import EnvBase
from simple_distributed_rl_amd.base.spaces.box import BoxSpace
from simple_distributed_rl_amd.base.spaces.discrete import DiscreteSpace

registration.register("CartPole-v1", __name__ + ":CartPole", {}, check_duplicate=False)

GRAVITY, MASS_CART, MASS_POLE, HALF_LENGTH, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
THETA_LIMIT, X_LIMIT = 12 * 2 * math.pi / 360, 2.4


@dataclass
class CartPole(EnvBase):
    max_steps: int = 500

    def __post_init__(self):
        super().__init__()
        self.state = np.zeros(4, np.float64)
        self.steps = 0

    @property
    def action_space(self) -> DiscreteSpace:
        return DiscreteSpace(2)

    @property
    def observation_space(self) -> BoxSpace:
        high = np.array([X_LIMIT * 2, np.finfo(np.float32).max, THETA_LIMIT * 2, np.finfo(np.float32).max], np.float32)
        return BoxSpace((4,), -high, high, np.float32)

    @property
    def player_num(self) -> int:
        return 1

    @property
    def max_episode_steps(self) -> int:
        return self.max_steps

    @property
    def reward_range(self) -> Tuple[float, float]:
        return 0.0, float(self.max_steps)

    @property
    def reward_baseline(self) -> dict:
        return {"episode": 10, "baseline": 100.0}

    def reset(self, *, seed: Optional[int] = None, **kwargs) -> Any:
        if seed is not None:
            random.seed(seed)
        self.state = np.array([random.uniform(-0.05, 0.05) for _ in range(4)], np.float64)
        self.steps = 0
        return self.state.astype(np.float32)

    def step(self, action) -> Tuple[Any, float, bool, bool]:
        x, x_dot, theta, theta_dot = self.state
        force = FORCE if int(action) == 1 else -FORCE
        cos_t, sin_t = math.cos(theta), math.sin(theta)
        total_mass, pole_ml = MASS_CART + MASS_POLE, MASS_POLE * HALF_LENGTH
        temp = (force + pole_ml * theta_dot * theta_dot * sin_t) / total_mass
        theta_acc = (GRAVITY * sin_t - cos_t * temp) / (HALF_LENGTH * (4.0 / 3.0 - MASS_POLE * cos_t * cos_t / total_mass))
        x_acc = temp - pole_ml * theta_acc * cos_t / total_mass
        x, x_dot = x + TAU * x_dot, x_dot + TAU * x_acc
        theta, theta_dot = theta + TAU * theta_dot, theta_dot + TAU * theta_acc
        self.state = np.array([x, x_dot, theta, theta_dot], np.float64)
        self.steps += 1
        terminated = bool(x < -X_LIMIT or x > X_LIMIT or theta < -THETA_LIMIT or theta > THETA_LIMIT)
        truncated = (not terminated) and self.steps >= self.max_steps
        return self.state.astype(np.float32), 1.0, terminated, truncated

    def backup(self, **kwargs) -> Any:
        return (self.state.copy(), self.steps)

    def restore(self, data: Any, **kwargs) -> None:
        self.state, self.steps = data[0].copy(), data[1]
