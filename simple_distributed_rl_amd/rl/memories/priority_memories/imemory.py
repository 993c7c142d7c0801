"""The priority-memory seam: what `PriorityReplayBuffer` expects of a memory implementation.

This is the drop-in boundary b1 of SURVEY.md section 8: the reference defines the same seven operations in
srl/rl/memories/priority_memories/imemory.py:7-34 and selects an implementation by name (or by entry point through
`set_custom`, priority_replay_buffer.py:111-117).  Every memory of this package -- the HIP sum-tree, the device-sorted
rank-based memory, the host-side uniform ring and sorted list -- subclasses it, and so can a user's own.

Contract of the operations (what the callers rely on):

* `add(batch, priority)`      -- `batch` is opaque; `priority=None` means "as important as anything seen so far".
* `sample(batch_size, step)`  -- exactly `batch_size` items; returns `(batches, weights, update_args)` where `weights` are the
                                 importance-sampling weights already divided by their maximum (list or ndarray) and
                                 `update_args` is whatever `update` needs to find the same items again.  `step` drives the
                                 beta schedule.
* `update(update_args, priorities)` -- new |TD| priorities for the items of the matching `sample` call (any order of calls).
* `backup()` / `restore(data)`      -- a picklable snapshot and its inverse; `restore` tolerates a different capacity.
* `clear()`, `length()`"""
import abc
from typing import Any, List, Optional, Sequence, Tuple, Union

import numpy as np

Weights = Union[Sequence[float], np.ndarray]


class IPriorityMemory(abc.ABC):
    @abc.abstractmethod
    def add(self, batch: Any, priority: Optional[float] = None) -> None:
        ...

    @abc.abstractmethod
    def sample(self, batch_size: int, step: int) -> Tuple[List[Any], Weights, List[Any]]:
        ...

    @abc.abstractmethod
    def update(self, update_args: List[Any], priorities: np.ndarray) -> None:
        ...

    @abc.abstractmethod
    def length(self) -> int:
        ...

    @abc.abstractmethod
    def clear(self) -> None:
        ...

    @abc.abstractmethod
    def backup(self) -> Any:
        ...

    @abc.abstractmethod
    def restore(self, data: Any) -> None:
        ...
