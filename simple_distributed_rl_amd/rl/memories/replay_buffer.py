"""Uniform replay of opaque items (reference contract: srl/rl/memories/replay_buffer.py:11-149).

A fixed-capacity ring with a warm-up gate; `sample` draws a batch with `random.sample` (so a seeded run reproduces the
reference's draws), items may be kept zlib-compressed pickles (`compress`), and an item that already crossed a process
boundary arrives serialized and is stored without a decode/encode round trip.  `RLReplayBuffer` is the RLMemory flavour
that algorithms subclass; PPO's memory wraps the plain class.  Host-side like the reference: the items are Python
objects (PPO's per-step dicts), there is nothing here for a GPU to do."""
import pickle
import random
import zlib
from dataclasses import dataclass
from typing import Any, List, Optional

from simple_distributed_rl_amd.base.rl.memory import RLMemory


class _ItemCodec:
    """How items are held in the ring: as they are, or as zlib-compressed pickles."""

    def __init__(self, compress: bool, level: int):
        self.compress, self.level = compress, level

    def pack(self, item: Any) -> Any:  # in-process add
        return zlib.compress(pickle.dumps(item), level=self.level) if self.compress else item

    def unpack(self, stored: Any) -> Any:
        return pickle.loads(zlib.decompress(stored)) if self.compress else stored

    def wire(self, item: Any) -> bytes:  # what a worker process sends
        raw = pickle.dumps(item)
        return zlib.compress(raw, level=self.level) if self.compress else raw

    def from_wire(self, data: bytes) -> Any:  # a wire item is already in stored form when the ring compresses
        return data if self.compress else pickle.loads(data)


@dataclass
class ReplayBufferConfig:
    capacity: int = 100_000
    warmup_size: int = 1_000
    compress: bool = True
    compress_level: int = -1

    def create_memory(self, batch_size: int = 32):
        return ReplayBuffer(batch_size, self.capacity, self.warmup_size, self.compress, self.compress_level)


class ReplayBuffer:
    def __init__(self, batch_size: int = 32, capacity: int = 100_000, warmup_size: int = 1000, compress: bool = True, compress_level: int = -1):
        if warmup_size > capacity:
            raise ValueError(f"assert {warmup_size} <= {capacity}")
        if not 0 < batch_size <= warmup_size:
            raise ValueError(f"assert 0 < {batch_size} <= {warmup_size}")
        self.batch_size, self.capacity, self.warmup_size = batch_size, capacity, warmup_size
        self.compress, self.compress_level = compress, compress_level
        self._codec = _ItemCodec(compress, compress_level)
        self.clear()

    def clear(self):
        self.buffer: List[Any] = []
        self.idx = 0  # next ring slot to overwrite once the ring is full

    def length(self) -> int:
        return len(self.buffer)

    def is_warmup_needed(self) -> bool:
        return self.length() < self.warmup_size

    def serialize(self, batch: Any) -> Any:
        return self._codec.wire(batch)

    def add(self, batch: Any, serialized: bool = False) -> None:
        stored = self._codec.from_wire(batch) if serialized else self._codec.pack(batch)
        if self.length() < self.capacity:
            self.buffer.append(stored)
        else:
            self.buffer[self.idx] = stored
        self.idx = (self.idx + 1) % self.capacity

    def sample(self, batch_size: int = -1) -> Optional[List[Any]]:
        if self.is_warmup_needed():
            return None
        picked = random.sample(self.buffer, self.batch_size if batch_size < 1 else batch_size)
        return [self._codec.unpack(b) for b in picked]

    def call_backup(self, **kwargs):
        return [list(self.buffer), self.idx, self.compress]

    def call_restore(self, data: Any, **kwargs) -> None:
        items, idx, was_compressed = list(data[0]), data[1], data[2]
        overflow = len(items) - self.capacity
        if overflow > 0:  # restoring into a smaller ring keeps the newest items
            items, idx = items[overflow:], max(0, idx - overflow)
        if was_compressed != self.compress:  # re-encode to this ring's storage form
            old = _ItemCodec(was_compressed, self.compress_level)
            items = [self._codec.pack(old.unpack(b)) for b in items]
        self.buffer, self.idx = items, idx % self.capacity


class RLReplayBuffer(ReplayBuffer, RLMemory):
    def __init__(self, *args):
        RLMemory.__init__(self, *args)
        m = self.config.memory
        ReplayBuffer.__init__(self, self.config.batch_size, m.capacity, m.warmup_size, m.compress, m.compress_level)

    def setup(self, register_add: bool = True, register_sample: bool = True) -> None:
        if register_add:
            self.register_worker_func_custom(self.add, self.serialize)
        if register_sample:
            self.register_trainer_recv_func(self.sample)
