"""BoxSpace (srl/base/spaces/box.py): bounded ndarray with an image/value type; frame-stack space and
encode_stack follow :285-312 (GRAY frames are stacked on a trailing channel axis, everything else on a
leading axis)."""
from typing import Any, List, Optional, Tuple, Union

import numpy as np

from simple_distributed_rl_amd.base.define import SpaceTypes

from .space import SpaceBase


class BoxSpace(SpaceBase):
    def __init__(self, shape: Tuple[int, ...], low: Union[float, np.ndarray] = -np.inf, high: Union[float, np.ndarray] = np.inf,
                 dtype=np.float32, stype: SpaceTypes = SpaceTypes.UNKNOWN, is_stack_ch: Optional[bool] = None) -> None:
        self._shape = tuple(int(s) for s in shape)
        self._dtype = dtype
        self._low = np.full(self._shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self._high = np.full(self._shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        if stype == SpaceTypes.UNKNOWN:
            stype = SpaceTypes.DISCRETE if "int" in str(np.dtype(dtype)) else SpaceTypes.CONTINUOUS
        self._stype = stype
        # gray frames gain a channel axis when stacked (box.py:24-30)
        self._is_stack_ch = (stype in (SpaceTypes.GRAY_HW, SpaceTypes.GRAY_HW1)) if is_stack_ch is None else is_stack_ch

    @property
    def shape(self):
        return self._shape

    @property
    def low(self):
        return self._low

    @property
    def high(self):
        return self._high

    @property
    def dtype(self):
        return self._dtype

    @property
    def stype(self):
        return self._stype

    def is_image_like(self) -> bool:
        return SpaceTypes.is_image(self._stype)

    def sample(self, mask: List[Any] = []) -> np.ndarray:
        lo = np.where(np.isfinite(self._low), self._low, -1.0)
        hi = np.where(np.isfinite(self._high), self._high, 1.0)
        return (np.random.uniform(lo, hi, self._shape)).astype(self._dtype)

    def get_default(self) -> np.ndarray:
        return np.zeros(self._shape, self._dtype)

    def check_val(self, val: Any) -> bool:
        return isinstance(val, np.ndarray) and val.shape == self._shape

    def copy_value(self, v):
        return np.array(v, copy=True)

    def to_str(self, val) -> str:
        return ",".join(str(v) for v in np.asarray(val).reshape(-1).tolist())

    def copy(self, **kw) -> "BoxSpace":
        return BoxSpace(self._shape, self._low, self._high, self._dtype, kw.get("stype", self._stype), kw.get("is_stack_ch", self._is_stack_ch))

    def create_stack_space(self, length: int) -> "BoxSpace":
        if self._is_stack_ch:
            return BoxSpace((self._shape[0], self._shape[1], length), np.min(self._low), np.max(self._high), self._dtype, SpaceTypes.IMAGE_MAP)
        return BoxSpace((length,) + self._shape, np.min(self._low), np.max(self._high), self._dtype, self._stype)

    def encode_stack(self, val: List[np.ndarray]) -> np.ndarray:
        state = np.asarray(val, self._dtype)
        if self._is_stack_ch:
            if self._stype == SpaceTypes.GRAY_HW:
                state = np.transpose(state, (1, 2, 0))
            elif self._stype == SpaceTypes.GRAY_HW1:
                state = np.transpose(np.squeeze(state, axis=-1), (1, 2, 0))
            else:
                raise ValueError(self._stype)
        return state

    def __eq__(self, o) -> bool:
        return isinstance(o, BoxSpace) and self._shape == o._shape and self._stype == o._stype and np.dtype(self._dtype) == np.dtype(o._dtype)

    def __str__(self) -> str:
        return f"Box{self._shape}, range[{np.min(self._low)}, {np.max(self._high)}], {np.dtype(self._dtype)}, {self._stype.name}"
