"""DiscreteSpace (srl/base/spaces/discrete.py): n actions from `start`; sample :43-46, get_onehot :116-121."""
import random
from typing import Any, List

import numpy as np

from simple_distributed_rl_amd.base.define import SpaceTypes

from .space import SpaceBase


class DiscreteSpace(SpaceBase):
    def __init__(self, n: int, start: int = 0) -> None:
        assert n > 0
        self._n, self._start = int(n), int(start)

    @property
    def n(self) -> int:
        return self._n

    @property
    def start(self) -> int:
        return self._start

    @property
    def stype(self):
        return SpaceTypes.DISCRETE

    def sample(self, mask: List[int] = []) -> int:
        acts = [a for a in range(self._start, self._start + self._n) if a not in mask]
        assert len(acts) > 0, f"No valid actions. {mask}"
        return random.choice(acts)

    def get_valid_actions(self, mask: List[int] = []) -> List[int]:
        return [a for a in range(self._start, self._start + self._n) if a not in mask]

    def get_default(self) -> int:
        return self._start

    def get_onehot(self, x: int) -> List[float]:
        onehot = [0.0] * self._n
        onehot[x - self._start] = 1.0
        return onehot

    def check_val(self, val: Any) -> bool:
        return isinstance(val, (int, np.integer)) and self._start <= val < self._start + self._n

    def to_str(self, val: int) -> str:
        return str(int(val))

    def copy(self) -> "DiscreteSpace":
        return DiscreteSpace(self._n, self._start)

    def create_stack_space(self, length: int):
        from .array_discrete import ArrayDiscreteSpace

        return ArrayDiscreteSpace(length, self._start, self._start + self._n - 1)

    def encode_stack(self, val: List[int]):
        return [int(v) for v in val]

    def __eq__(self, o) -> bool:
        return isinstance(o, DiscreteSpace) and (self._n, self._start) == (o._n, o._start)

    def __str__(self) -> str:
        return f"Discrete({self._n})"
