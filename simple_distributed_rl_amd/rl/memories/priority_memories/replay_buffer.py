"""The "no priorities" member of the priority-memory family: a uniform ring behind the IPriorityMemory seam
(reference contract: srl/rl/memories/priority_memories/replay_buffer.py:10-55; BASELINE.json configs[1] uses it for DQN).

`sample` is `random.sample` over the stored items (a seeded run draws what the reference draws), every importance
weight is 1.0 and there is nothing to update.  Host-only by nature: the items are opaque Python objects."""
import random
from typing import Any, List, Optional

import numpy as np

from .imemory import IPriorityMemory


class ReplayBuffer(IPriorityMemory):
    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        self.memory: List[Any] = []
        self.idx = 0  # slot the next add overwrites once the ring is full

    # ---- IPriorityMemory ------------------------------------------------------------------------
    def add(self, batch: Any, priority: Optional[float] = None) -> None:
        if self.length() == self.capacity:
            self.memory[self.idx] = batch
        else:
            self.memory.append(batch)
        self.idx = (self.idx + 1) % self.capacity

    def sample(self, batch_size: int, step: int):
        return random.sample(self.memory, batch_size), [1.0] * batch_size, []

    def update(self, update_args: List[Any], priorities: np.ndarray) -> None:
        return None  # uniform replay has no priorities

    def length(self) -> int:
        return len(self.memory)

    def clear(self) -> None:
        self.memory, self.idx = [], 0

    def backup(self):
        return [list(self.memory), self.idx]

    def restore(self, data) -> None:
        items, idx = list(data[0]), data[1]
        surplus = len(items) - self.capacity
        if surplus > 0:  # a backup of a larger ring: the newest `capacity` items survive
            items, idx = items[surplus:], max(0, idx - surplus)
        self.memory, self.idx = items, idx % self.capacity
