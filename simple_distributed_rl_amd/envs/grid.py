"""Grid world (srl/envs/grid.py:20-375): the 4x3 stochastic grid of the reference's README example,
registered under the same ids ("Grid", "EasyGrid").  field codes: 0 road, 1 goal, 2 start, -1 hole, 9 wall.
An action succeeds with `move_prob`, otherwise slips to one of the two perpendicular directions;
every move costs `move_reward`; goal / hole end the episode with +1 / -1 (`terminated`)."""
import enum
import random
from dataclasses import dataclass, field
from typing import Any, List, Optional, Tuple

import numpy as np

from simple_distributed_rl_amd.base.env import registration
from simple_distributed_rl_amd.base.env.base import EnvBase
from simple_distributed_rl_amd.base.spaces.array_discrete import ArrayDiscreteSpace
from simple_distributed_rl_amd.base.spaces.discrete import DiscreteSpace

registration.register("Grid", __name__ + ":Grid", {"move_reward": -0.04, "move_prob": 0.8, "reward_baseline_": {"episode": 100, "baseline": 0.65}}, check_duplicate=False)
registration.register("EasyGrid", __name__ + ":Grid", {"move_reward": 0.0, "move_prob": 1.0, "reward_baseline_": {"episode": 100, "baseline": 0.9}}, check_duplicate=False)


class Action(enum.Enum):
    LEFT = 0
    DOWN = 1
    RIGHT = 2
    UP = 3


_SLIPS = {Action.UP: (Action.RIGHT, Action.LEFT), Action.DOWN: (Action.RIGHT, Action.LEFT),
          Action.RIGHT: (Action.UP, Action.DOWN), Action.LEFT: (Action.UP, Action.DOWN)}
_DELTA = {Action.LEFT: (-1, 0), Action.DOWN: (0, 1), Action.RIGHT: (1, 0), Action.UP: (0, -1)}


@dataclass
class Grid(EnvBase):
    move_prob: float = 0.8
    move_reward: float = -0.04
    reward_baseline_: dict = field(default_factory=lambda: {"episode": 10, "baseline": 0})
    goal_reward: float = 1.0
    hole_reward: float = -1.0
    field: List[List[int]] = field(
        default_factory=lambda: [
            [9, 9, 9, 9, 9, 9],
            [9, 0, 0, 0, 1, 9],
            [9, 0, 9, 0, -1, 9],
            [9, 2, 0, 0, 0, 9],
            [9, 9, 9, 9, 9, 9],
        ]
    )

    def __post_init__(self):
        super().__init__()
        self.start_pos_list = [(x, y) for y in range(self.H) for x in range(self.W) if self.field[y][x] == 2]
        assert len(self.start_pos_list) > 0, "There is no initial position. Enter '2' locations in the field."
        self.player_pos = self.start_pos_list[0]

    @property
    def W(self) -> int:
        return len(self.field[0])

    @property
    def H(self) -> int:
        return len(self.field)

    @property
    def action_space(self) -> DiscreteSpace:
        return DiscreteSpace(len(Action))

    @property
    def observation_space(self) -> ArrayDiscreteSpace:
        return ArrayDiscreteSpace(2, 0, [self.W - 1, self.H - 1])

    @property
    def player_num(self) -> int:
        return 1

    @property
    def max_episode_steps(self) -> int:
        return 50

    @property
    def reward_range(self) -> Tuple[float, float]:
        return (self.max_episode_steps - 1) * self.move_reward - 1, 5 * self.move_reward + 1

    @property
    def reward_baseline(self) -> dict:
        return self.reward_baseline_

    def reset(self, *, seed: Optional[int] = None, **kwargs) -> Any:
        self.player_pos = random.choice(self.start_pos_list)
        return [self.player_pos[0], self.player_pos[1]]

    def step(self, action) -> Tuple[Any, float, bool, bool]:
        action = Action(action)
        side = (1 - self.move_prob) / 2
        choices = [action, _SLIPS[action][0], _SLIPS[action][1]]
        actual = choices[np.random.choice(3, p=[self.move_prob, side, side])]
        dx, dy = _DELTA[actual]
        nx, ny = self.player_pos[0] + dx, self.player_pos[1] + dy
        if 0 <= nx < self.W and 0 <= ny < self.H and self.field[ny][nx] != 9:
            self.player_pos = (nx, ny)
        cell = self.field[self.player_pos[1]][self.player_pos[0]]
        if cell == 1:
            reward, done = self.goal_reward, True
        elif cell == -1:
            reward, done = self.hole_reward, True
        else:
            reward, done = self.move_reward, False
        return [self.player_pos[0], self.player_pos[1]], reward, done, False

    def backup(self, **kwargs) -> Any:
        return tuple(self.player_pos)

    def restore(self, data: Any, **kwargs) -> None:
        self.player_pos = tuple(data)

    def action_to_str(self, action) -> str:
        return {0: "←", 1: "↓", 2: "→", 3: "↑"}.get(action, str(action))
