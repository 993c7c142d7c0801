"""RankBasedMemoryLinear (srl/rl/memories/priority_memories/rankbased_memory_linear.py:14-113): items kept SORTED by
priority; the sampling density grows linearly with the rank (weight of rank k is 1 + k*alpha), a draw inverts the
closed-form cumulative sum (:14-23), sampled items LEAVE the memory and come back through update() with their new
priority (:84-93).

Host-side like the reference: the structure is a sorted list of opaque Python items whose every operation moves list
entries -- there is no array arithmetic to hand to the GPU (the O(B) weights are numpy as in :74-79).  Priorities and
items live in two parallel lists; among equal priorities the items' own order decides, like the reference's (priority, item)
tuples (:51) -- except that a tie between unorderable items goes after the equal priorities instead of raising."""
import bisect
import logging
import random
from typing import Any, List, Optional

import numpy as np

from .imemory import IPriorityMemory

logger = logging.getLogger(__name__)


def rank_sum(k, a):
    """sum_{i<k} (1 + i*a)  (:14-15)"""
    return k * (2 + (k - 1) * a) / 2


def rank_sum_inverse(k, a):
    """the rank whose cumulative weight reaches k  (:18-23)"""
    if a == 0:
        return k
    return (a - 2 + np.sqrt((2 - a) ** 2 + 8 * a * k)) / (2 * a)


class RankBasedMemoryLinear(IPriorityMemory):
    def __init__(self, capacity: int = 100_000, alpha: float = 1.0, beta_initial: float = 0.4, beta_steps: int = 1_000_000):
        self.capacity, self.alpha, self.beta_initial, self.beta_steps = capacity, alpha, beta_initial, beta_steps
        self.clear()

    def clear(self):
        self.keys: List[float] = []  # ascending priorities
        self.items: List[Any] = []
        self.max_priority: float = 1.0

    def length(self) -> int:
        return len(self.items)

    def add(self, batch: Any, priority: Optional[float] = None):
        if priority is None:
            priority = self.max_priority
        if self.max_priority < priority:
            self.max_priority = priority
        if len(self.items) >= self.capacity:  # the lowest priority makes room (:48-49)
            del self.keys[0]
            del self.items[0]
        lo, pos = bisect.bisect_left(self.keys, priority), bisect.bisect_right(self.keys, priority)
        if lo < pos:  # equal priorities: the reference's (priority, item) tuples fall back to the items' own order (:51)
            try:
                pos = bisect.bisect_right(self.items, batch, lo, pos)
            except TypeError:  # unorderable items: after the equal priorities (the reference raises here)
                pass
        self.keys.insert(pos, priority)
        self.items.insert(pos, batch)

    def sample(self, batch_size: int, step: int):
        beta = min(1.0, self.beta_initial + (1 - self.beta_initial) * step / self.beta_steps)  # :55-57
        size = len(self.items)
        total = rank_sum(size, self.alpha)
        picked: List[int] = []
        for _ in range(batch_size):  # no duplicates inside a batch (:62-70); 999 tries like the reference
            idx = 0
            for _ in range(999):
                idx = int(rank_sum_inverse(random.random() * total, self.alpha))
                if idx not in picked:
                    break
            picked.append(idx)
        picked.sort(reverse=True)
        idx_arr = np.array(picked)
        prob = (rank_sum(idx_arr + 1, self.alpha) - rank_sum(idx_arr, self.alpha)) / total  # :74-76
        weights = (size * prob) ** (-beta)
        weights = weights / weights.max()
        batches = [self.items[i] for i in picked]
        for i in picked:  # descending, so earlier deletions do not shift the later ones (:81-82)
            del self.keys[i]
            del self.items[i]
        return batches, weights, batches

    def update(self, batches, priorities: np.ndarray) -> None:
        for b, p in zip(batches, priorities):
            self.add(b, p)

    def backup(self):
        return [self.capacity, [[k, it] for k, it in zip(self.keys, self.items)], self.max_priority]  # the reference's layout (:95-100)

    def restore(self, data):
        if self.capacity != data[0]:
            logger.warning("Capacity mismatch: expected %d, but got %d", self.capacity, data[0])
        self.keys = [d[0] for d in data[1]]
        self.items = [d[1] for d in data[1]]
        self.max_priority = data[2]
