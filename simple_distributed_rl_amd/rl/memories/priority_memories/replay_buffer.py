"""Uniform ring behind IPriorityMemory (srl/rl/memories/priority_memories/replay_buffer.py:10-55):
`random.sample`, weights all 1.0, empty update_args.  Host-only by nature (opaque Python items)."""
import random
from typing import Any, List, Optional

import numpy as np

from .imemory import IPriorityMemory


class ReplayBuffer(IPriorityMemory):
    def __init__(self, capacity: int):
        self.capacity = capacity
        self.clear()

    def clear(self):
        self.memory = []
        self.idx = 0

    def length(self) -> int:
        return len(self.memory)

    def add(self, batch: Any, priority: Optional[float] = None):
        if len(self.memory) < self.capacity:
            self.memory.append(batch)
        else:
            self.memory[self.idx] = batch
        self.idx += 1
        if self.idx >= self.capacity:
            self.idx = 0

    def sample(self, batch_size: int, step: int):
        return random.sample(self.memory, batch_size), [1.0 for _ in range(batch_size)], []

    def update(self, update_args: List[Any], priorities: np.ndarray) -> None:
        pass

    def backup(self):
        return [self.memory[:], self.idx]

    def restore(self, data):
        self.memory = data[0][:]
        self.idx = data[1]
        if len(self.memory) > self.capacity:
            self.idx = max(0, self.idx - (len(self.memory) - self.capacity))
            self.memory = self.memory[-self.capacity :]
        if self.idx >= self.capacity:
            self.idx = 0
