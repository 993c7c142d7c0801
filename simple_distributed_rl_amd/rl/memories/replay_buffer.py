"""ReplayBuffer (srl/rl/memories/replay_buffer.py:11-149): a ring list with a warm-up gate, uniform `random.sample`
batches and optional zlib+pickle item compression; `RLReplayBuffer` is the RLMemory flavour algorithms subclass.
Host-side like the reference: it holds opaque Python items (PPO's per-step dicts), there is nothing for a GPU to do."""
import pickle
import random
import zlib
from dataclasses import dataclass
from typing import Any

from simple_distributed_rl_amd.base.rl.memory import RLMemory


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
        self.batch_size, self.capacity, self.warmup_size = batch_size, capacity, warmup_size
        self.compress, self.compress_level = compress, compress_level
        self.buffer = []
        self.idx = 0
        if not (warmup_size <= capacity):
            raise ValueError(f"assert {warmup_size} <= {capacity}")
        if not (0 < batch_size <= warmup_size):
            raise ValueError(f"assert 0 < {batch_size} <= {warmup_size}")

    def clear(self):
        self.buffer = []
        self.idx = 0

    def length(self) -> int:
        return len(self.buffer)

    def add(self, batch: Any, serialized: bool = False) -> None:
        if serialized:  # :69-71 a serialized item stays compressed if the buffer compresses
            if not self.compress:
                batch = pickle.loads(batch)
        elif self.compress:
            batch = zlib.compress(pickle.dumps(batch), level=self.compress_level)
        if len(self.buffer) < self.capacity:
            self.buffer.append(batch)
        else:
            self.buffer[self.idx] = batch
        self.idx = (self.idx + 1) % self.capacity

    def serialize(self, batch: Any) -> Any:
        batch = pickle.dumps(batch)
        return zlib.compress(batch, level=self.compress_level) if self.compress else batch

    def is_warmup_needed(self) -> bool:
        return len(self.buffer) < self.warmup_size

    def sample(self, batch_size: int = -1):
        if len(self.buffer) < self.warmup_size:
            return None
        batches = random.sample(self.buffer, batch_size if batch_size >= 1 else self.batch_size)
        return [pickle.loads(zlib.decompress(b)) for b in batches] if self.compress else batches

    def call_backup(self, **kwargs):
        return [self.buffer[:], self.idx, self.compress]

    def call_restore(self, data: Any, **kwargs) -> None:
        self.buffer, self.idx, compressed = data[0][:], data[1], data[2]
        if len(self.buffer) > self.capacity:  # :116-121 a smaller buffer keeps the newest items
            self.idx = max(0, self.idx - (len(self.buffer) - self.capacity))
            self.buffer = self.buffer[-self.capacity:]
        if self.idx >= self.capacity:
            self.idx = 0
        if compressed and not self.compress:
            self.buffer = [pickle.loads(zlib.decompress(b)) for b in self.buffer]
        if not compressed and self.compress:
            self.buffer = [zlib.compress(pickle.dumps(b)) for b in self.buffer]


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
