"""NpArraySpace: a flat float vector with per-component bounds -- the RL-side action space of continuous-control
algorithms (reference contract: srl/base/spaces/np_array.py; used by base_ppo.py:17-23 and ppo.py:332-339).

`rescale_from(x, src_low, src_high)` maps a policy output living in [src_low, src_high] (the reference's policies use
[-1, 1]) affinely onto [low, high]; `sanitize` makes whatever the policy produced a legal action (shape, dtype, clipped)."""
from typing import Any, List, Union

import numpy as np

from simple_distributed_rl_amd.base.define import SpaceTypes

from .space import SpaceBase


class NpArraySpace(SpaceBase):
    def __init__(self, size: int, low: Union[float, np.ndarray] = -np.inf, high: Union[float, np.ndarray] = np.inf, dtype=np.float32) -> None:
        self._size = int(size)
        self._dtype = dtype
        self._low = np.full((self._size,), low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype).reshape(-1)
        self._high = np.full((self._size,), high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype).reshape(-1)
        assert self._low.shape == self._high.shape == (self._size,)

    @property
    def size(self) -> int:
        return self._size

    @property
    def shape(self):
        return (self._size,)

    @property
    def low(self) -> np.ndarray:
        return self._low

    @property
    def high(self) -> np.ndarray:
        return self._high

    @property
    def dtype(self):
        return self._dtype

    @property
    def stype(self):
        return SpaceTypes.CONTINUOUS

    def _finite_bounds(self):
        return np.where(np.isfinite(self._low), self._low, -1.0), np.where(np.isfinite(self._high), self._high, 1.0)

    def sample(self, mask: List[Any] = []) -> np.ndarray:
        lo, hi = self._finite_bounds()
        return np.random.uniform(lo, hi).astype(self._dtype)

    def sanitize(self, val: Any) -> np.ndarray:
        x = np.asarray(val, dtype=self._dtype).reshape(-1)
        if x.size != self._size:
            x = np.resize(x, self._size)
        return np.clip(x, self._low, self._high).astype(self._dtype)

    def rescale_from(self, x: Any, src_low: float = -1.0, src_high: float = 1.0) -> np.ndarray:
        lo, hi = self._finite_bounds()
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        return (((x - src_low) / (src_high - src_low)) * (hi - lo) + lo).astype(self._dtype)

    def get_default(self) -> np.ndarray:
        return np.zeros((self._size,), self._dtype)

    def check_val(self, val: Any) -> bool:
        return isinstance(val, np.ndarray) and val.shape == (self._size,) and bool(np.all(val >= self._low) and np.all(val <= self._high))

    def copy_value(self, v):
        return np.array(v, copy=True)

    def to_str(self, val) -> str:
        return ",".join(str(v) for v in np.asarray(val).reshape(-1).tolist())

    def copy(self) -> "NpArraySpace":
        return NpArraySpace(self._size, self._low, self._high, self._dtype)

    def __eq__(self, o) -> bool:
        return isinstance(o, NpArraySpace) and self._size == o._size and np.array_equal(self._low, o._low) and np.array_equal(self._high, o._high)

    def __str__(self) -> str:
        return f"NpArray({self._size}), range[{np.min(self._low)}, {np.max(self._high)}]"
