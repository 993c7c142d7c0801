"""ArrayDiscreteSpace (srl/base/spaces/array_discrete.py): fixed-length list of bounded ints (Grid's (x, y))."""
import random
from typing import Any, List, Union

import numpy as np

from simple_distributed_rl_amd.base.define import SpaceTypes

from .space import SpaceBase


class ArrayDiscreteSpace(SpaceBase):
    def __init__(self, size: int, low: Union[int, List[int]], high: Union[int, List[int]]) -> None:
        self._size = int(size)
        self._low = [int(low)] * size if isinstance(low, (int, np.integer)) else [int(v) for v in low]
        self._high = [int(high)] * size if isinstance(high, (int, np.integer)) else [int(v) for v in high]
        assert len(self._low) == size and len(self._high) == size

    @property
    def size(self) -> int:
        return self._size

    @property
    def low(self) -> List[int]:
        return self._low

    @property
    def high(self) -> List[int]:
        return self._high

    @property
    def stype(self):
        return SpaceTypes.DISCRETE

    def sample(self, mask: List[Any] = []) -> List[int]:
        return [random.randint(self._low[i], self._high[i]) for i in range(self._size)]

    def get_default(self) -> List[int]:
        return [0 if self._low[i] <= 0 <= self._high[i] else self._low[i] for i in range(self._size)]

    def check_val(self, val: Any) -> bool:
        return isinstance(val, list) and len(val) == self._size and all(self._low[i] <= v <= self._high[i] for i, v in enumerate(val))

    def to_str(self, val) -> str:
        return ",".join(str(int(v)) for v in val)

    def copy(self) -> "ArrayDiscreteSpace":
        return ArrayDiscreteSpace(self._size, self._low[:], self._high[:])

    def create_stack_space(self, length: int):
        return ArrayDiscreteSpace(length * self._size, self._low * length, self._high * length)

    def encode_stack(self, val: List[List[int]]):
        return [e for sub in val for e in sub]

    def __eq__(self, o) -> bool:
        return isinstance(o, ArrayDiscreteSpace) and (self._size, self._low, self._high) == (o._size, o._low, o._high)

    def __str__(self) -> str:
        return f"ArrayDiscrete({self._size}, range[{min(self._low)}, {max(self._high)}])"
