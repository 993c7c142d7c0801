"""Rank-based prioritised replay behind the IPriorityMemory surface
(srl/rl/memories/priority_memories/rankbased_memory.py:13-77), SURVEY 8 f3.

The reference argsorts all N priorities on every sample().  Here the priorities live on the GPU: which RANKS are drawn
is decided on the host exactly like the reference does it (`np.random.choice` over the rank probabilities: same
generator, same consumption, same ranks), and `srlx_rank_select` turns ranks into buffer indices with one descending
radix sort in HBM.  Host-side edits (add / update) are mirrored in a float32 array and uploaded lazily before the next
sample.  Ties between equal priorities are ordered by index on the device (numpy's introsort leaves them unspecified)."""
import ctypes
import logging
from typing import Any, List, Optional

import numpy as np

from simple_distributed_rl_amd import _native as N

from .imemory import IPriorityMemory

logger = logging.getLogger(__name__)


class RankBasedMemory(IPriorityMemory):
    def __init__(self, capacity: int = 100_000, alpha: float = 0.6, beta_initial: float = 0.4, beta_steps: int = 1_000_000, device: int = 0):
        import torch

        self.capacity, self.alpha, self.beta_initial, self.beta_steps = int(capacity), alpha, beta_initial, beta_steps
        self._torch = torch
        self._lib = N.lib()  # raises without libsrlx / a GPU: no CPU fallback
        self._dev = torch.device(f"cuda:{device}")
        h = N.c_p()
        N.check(self._lib.srlx_rank_create(ctypes.byref(h), self.capacity, int(device)))
        self._h = h
        p = N.c_p()
        N.check(self._lib.srlx_rank_priorities(self._h, ctypes.byref(p)))
        self._prob_cache = (0, None)
        self.clear()

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.srlx_rank_destroy(self._h)
            self._h = None

    def clear(self):
        self.buffer: List[Any] = []
        self.priorities = np.zeros(self.capacity, dtype=np.float32)  # host mirror of the device array
        self.pos = 0
        self._dirty = True

    def length(self) -> int:
        return len(self.buffer)

    def add(self, batch, priority: Optional[float] = None):
        if len(self.buffer) < self.capacity:
            self.buffer.append(batch)
        else:
            self.buffer[self.pos] = batch
        self.priorities[self.pos] = priority
        self.pos = (self.pos + 1) % self.capacity
        self._dirty = True

    def _rank_probs(self, n: int) -> np.ndarray:
        if self._prob_cache[0] != n:  # depends on N and alpha only (:48-50)
            probs = (1 / np.arange(1, n + 1)) ** self.alpha
            probs /= probs.sum()
            self._prob_cache = (n, probs)
        return self._prob_cache[1]

    def _upload(self, n: int):
        torch = self._torch
        host = torch.from_numpy(self.priorities[:n])
        dev = host.to(self._dev)
        N.check(self._lib.srlx_rank_set(self._h, n, None, N.tptr(dev), 0, N.torch_stream_ptr()))
        self._keep = dev
        self._dirty = False

    def sample(self, batch_size: int, step: int):
        torch = self._torch
        beta = min(1, self.beta_initial + (1 - self.beta_initial) * step / self.beta_steps)
        n = len(self.buffer)
        probs = self._rank_probs(n)
        ranks = np.random.choice(n, size=batch_size, p=probs, replace=False)  # the reference's draw, on rank positions (:54)
        if self._dirty:
            self._upload(n)
        d_ranks = torch.from_numpy(np.ascontiguousarray(ranks, dtype=np.int64)).to(self._dev)
        out = torch.empty(batch_size, dtype=torch.int64, device=self._dev)
        N.check(self._lib.srlx_rank_select(self._h, n, batch_size, N.tptr(d_ranks), N.tptr(out), N.torch_stream_ptr()))
        sampled = out.cpu().numpy()
        weights = (n * probs[ranks]) ** (-beta)  # :55-56
        weights = weights / weights.max()
        return [self.buffer[i] for i in sampled], weights, sampled

    def update(self, indices, priorities: np.ndarray) -> None:
        idx = np.asarray(indices, dtype=np.int64)
        self.priorities[idx] = np.asarray(priorities, dtype=np.float32)  # later duplicates win, like the reference's loop
        self._dirty = True

    def backup(self):
        return [self.capacity, self.buffer[:], self.priorities.copy(), self.pos]

    def restore(self, data):
        if self.capacity != data[0]:
            logger.warning("Capacity mismatch: expected %d, but got %d", self.capacity, data[0])
        self.buffer = data[1][:]
        self.priorities = np.zeros(self.capacity, dtype=np.float32)
        src = np.asarray(data[2], dtype=np.float32)
        self.priorities[: min(len(src), self.capacity)] = src[: self.capacity]
        self.pos = data[3]
        self._dirty = True
