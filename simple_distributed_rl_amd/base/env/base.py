"""EnvBase: the environment plugin contract (srl/base/env/base.py:18-206), single- or multi-player,
gymnasium-style step() -> (state, reward(s), terminated, truncated)."""
import math
from abc import ABC, abstractmethod
from typing import Any, List, Optional, Tuple, Union


class EnvBase(ABC):
    def __init__(self) -> None:
        self.init_base()

    def __post_init__(self) -> None:  # dataclass envs
        self.init_base()

    def init_base(self):
        if not hasattr(self, "next_player"):
            self.next_player: int = 0
        if not hasattr(self, "done_reason"):
            self.done_reason: str = ""
        if not hasattr(self, "info"):
            self.info: dict = {}
        self.env_run = None
        self.training = False
        return self

    @property
    @abstractmethod
    def action_space(self):
        raise NotImplementedError()

    @property
    @abstractmethod
    def observation_space(self):
        raise NotImplementedError()

    @property
    @abstractmethod
    def max_episode_steps(self) -> int:
        raise NotImplementedError()

    @property
    @abstractmethod
    def player_num(self) -> int:
        raise NotImplementedError()

    @property
    def reward_range(self) -> Tuple[float, float]:
        return (-math.inf, math.inf)

    @property
    def reward_baseline(self):
        return None

    def setup(self, **kwargs) -> None:
        self.training = bool(kwargs.get("training", False))

    def teardown(self, **kwargs) -> None:
        pass

    @abstractmethod
    def reset(self, *, seed: Optional[int] = None, **kwargs) -> Any:
        raise NotImplementedError()

    @abstractmethod
    def step(self, action) -> Tuple[Any, Union[float, List[float]], bool, bool]:
        raise NotImplementedError()

    def get_invalid_actions(self, player_index: int = -1) -> list:
        return []

    def backup(self, **kwargs) -> Any:
        raise NotImplementedError()

    def restore(self, data: Any, **kwargs) -> None:
        raise NotImplementedError()

    def close(self) -> None:
        pass

    def action_to_str(self, action) -> str:
        return str(action)
