"""PriorityReplayBufferConfig / PriorityReplayBuffer / RLPriorityReplayBuffer
(srl/rl/memories/priority_replay_buffer.py:17-274): warm-up gate, optional zlib+pickle item compression,
weight dtype cast, `step` bookkeeping (sample uses the LAST update()'s step for beta, :232,250), and the
memory selector.  "Proportional" and "Proportional_cpp" both resolve to the HBM-resident sum-tree of
libsrlx (there is no CPU tree in this build); `set_custom` keeps working for user memories.  The demo-memory mix
(:175-187,237-240) is kept: a uniform host ring of demonstration items supplies `max(1, int(batch_size * demo_ratio))` items of
every batch with weight 1 (the reference appends a single weight whatever that count is and would decode an already decoded
item under `compress`; here every demonstration item gets its weight and is decoded once)."""
import pickle
import zlib
from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np

from simple_distributed_rl_amd.base.exception import UndefinedError
from simple_distributed_rl_amd.base.rl.memory import RLMemory

from .priority_memories.imemory import IPriorityMemory


@dataclass
class PriorityReplayBufferConfig:
    capacity: int = 100_000
    warmup_size: int = 1_000
    compress: bool = True
    compress_level: int = -1
    name: str = field(default="ReplayBuffer")
    kwargs: dict = field(default_factory=dict)
    enable_demo_memory: bool = False
    select_memory: str = "main"
    demo_ratio: float = 1.0 / 256.0

    def set_replay_buffer(self):
        self.name, self.kwargs = "ReplayBuffer", {}
        return self

    def set_proportional(self, alpha: float = 0.6, beta_initial: float = 0.4, beta_steps: int = 1_000_000, has_duplicate: bool = True,
                         epsilon: float = 0.0001, device: int = 0):
        self.name = "Proportional"
        self.kwargs = dict(alpha=alpha, beta_initial=beta_initial, beta_steps=beta_steps, has_duplicate=has_duplicate, epsilon=epsilon, device=device)
        return self

    def set_proportional_cpp(self, alpha: float = 0.6, beta_initial: float = 0.4, beta_steps: int = 1_000_000, has_duplicate: bool = True,
                             epsilon: float = 0.0001, force_build: bool = False):
        """Same constructor surface as the pybind11 twin (cpp_module/src/proportional_memory.cpp:250-259); redirected to the HIP tree."""
        self.set_proportional(alpha, beta_initial, beta_steps, has_duplicate, epsilon)
        self.name = "Proportional_cpp"
        return self

    def set_rankbased(self, alpha: float = 0.6, beta_initial: float = 0.4, beta_steps: int = 1_000_000, device: int = 0):
        self.name = "RankBased"
        self.kwargs = dict(alpha=alpha, beta_initial=beta_initial, beta_steps=beta_steps, device=device)
        return self

    def set_rankbased_linear(self, alpha: float = 0.6, beta_initial: float = 0.4, beta_steps: int = 1_000_000):
        self.name = "RankBasedLinear"
        self.kwargs = dict(alpha=alpha, beta_initial=beta_initial, beta_steps=beta_steps)
        return self

    def set_custom(self, entry_point: str, kwargs: dict):
        self.name = "custom"
        self.kwargs = dict(entry_point=entry_point, kwargs=kwargs)
        return self

    def create_memory(self, capacity: int) -> IPriorityMemory:
        if self.name == "ReplayBuffer":
            from .priority_memories.replay_buffer import ReplayBuffer

            return ReplayBuffer(capacity, **self.kwargs)
        if self.name in ("Proportional", "Proportional_cpp"):
            from .priority_memories.proportional_memory import ProportionalMemory

            return ProportionalMemory(capacity, **self.kwargs)
        if self.name == "RankBased":
            from .priority_memories.rankbased_memory import RankBasedMemory

            return RankBasedMemory(capacity, **self.kwargs)
        if self.name == "RankBasedLinear":
            from .priority_memories.rankbased_memory_linear import RankBasedMemoryLinear

            return RankBasedMemoryLinear(capacity, **self.kwargs)
        if self.name == "custom":
            from simple_distributed_rl_amd.utils.common import load_module

            return load_module(self.kwargs["entry_point"])(capacity, **self.kwargs["kwargs"])
        raise UndefinedError(self.name)

    def requires_priority(self) -> bool:
        return self.name in ("Proportional", "Proportional_cpp", "RankBased", "RankBasedLinear")

    def validate_params(self) -> None:
        if not (self.warmup_size <= self.capacity):
            raise ValueError(f"assert {self.warmup_size} <= {self.capacity}")


class PriorityReplayBuffer:
    def __init__(self, config: PriorityReplayBufferConfig, batch_size: int, dtype=np.float32):
        self.cfg = config
        self.dtype = dtype
        self.memory = self.cfg.create_memory(self.cfg.capacity)
        self.step = 0
        if self.cfg.enable_demo_memory:  # :175-187: a second, uniform ring of demonstration items that takes a share of every batch
            from simple_distributed_rl_amd.rl.memories.replay_buffer import ReplayBuffer

            self.demo_batch_size = max(1, int(batch_size * self.cfg.demo_ratio))
            self.demo_memory = ReplayBuffer(self.demo_batch_size, self.cfg.capacity, self.demo_batch_size, self.cfg.compress, self.cfg.compress_level)
            batch_size = batch_size - self.demo_batch_size
        self.batch_size = batch_size
        if not (self.cfg.warmup_size <= self.cfg.capacity):
            raise ValueError(f"assert {self.cfg.warmup_size} <= {self.cfg.capacity}")
        if not (batch_size > 0):
            raise ValueError(f"assert {batch_size} > 0")
        if not (batch_size <= self.cfg.warmup_size):
            raise ValueError(f"assert {batch_size} <= {self.cfg.warmup_size}")

    def length(self) -> int:
        return self.memory.length() + (self.demo_memory.length() if self.cfg.enable_demo_memory else 0)

    def add(self, batch: Any, priority: Optional[float] = None, serialized: bool = False) -> None:
        to_demo = self.cfg.enable_demo_memory and self.cfg.select_memory == "demo"  # :212-215
        if to_demo:
            if serialized:  # the demo ring has its own codec: hand it the plain item
                batch = pickle.loads(zlib.decompress(batch) if self.cfg.compress else batch)
            self.demo_memory.add(batch)
            return
        if serialized:
            if not self.cfg.compress:
                batch = pickle.loads(batch)
        elif self.cfg.compress:
            batch = zlib.compress(pickle.dumps(batch), level=self.cfg.compress_level)
        self.memory.add(batch, priority)

    def serialize(self, batch: Any, priority: Optional[float] = None) -> Any:
        batch = pickle.dumps(batch)
        if self.cfg.compress:
            batch = zlib.compress(batch, level=self.cfg.compress_level)
        return (batch, priority)

    def is_warmup_needed(self) -> bool:
        return self.memory.length() < self.cfg.warmup_size

    def sample(self, step: int = -1, batch_size: int = -1):
        if self.memory.length() < self.cfg.warmup_size:
            return None
        if self.cfg.enable_demo_memory and self.demo_memory.length() < self.demo_batch_size:
            return None  # the demonstration ring cannot fill its share yet (the reference's random.sample raises here): still warming up, never a short batch
        batch_size = batch_size if batch_size > -1 else self.batch_size
        step = step if step > -1 else self.step
        batches, weights, update_args = self.memory.sample(batch_size, step)
        weights = np.asarray(weights, dtype=self.dtype)
        if self.cfg.compress:
            batches = [pickle.loads(zlib.decompress(b)) for b in batches]
        if self.cfg.enable_demo_memory:  # :237-240: demonstration items ride at the tail of the batch with weight 1
            demo = self.demo_memory.sample(self.demo_batch_size)
            if demo is not None:
                batches = list(batches) + demo
                weights = np.append(weights, np.ones(len(demo))).astype(weights.dtype)
        return batches, weights, update_args

    def update(self, update_args: List[Any], priorities: np.ndarray, step: int) -> None:
        if self.cfg.enable_demo_memory:
            priorities = priorities[: self.batch_size]  # :247-248: the demonstration tail has no priority
        self.memory.update(update_args, priorities)
        self.step = step

    def call_backup(self, **kwargs):
        return [self.memory.backup(), self.demo_memory.call_backup() if self.cfg.enable_demo_memory else None]

    def call_restore(self, data: Any, **kwargs) -> None:
        self.memory.restore(data[0])
        if self.cfg.enable_demo_memory and data[1] is not None:
            self.demo_memory.call_restore(data[1])


class RLPriorityReplayBuffer(PriorityReplayBuffer, RLMemory):
    def __init__(self, *args):
        RLMemory.__init__(self, *args)
        assert hasattr(self.config, "memory") and hasattr(self.config, "batch_size")
        assert isinstance(self.config.memory, PriorityReplayBufferConfig)
        PriorityReplayBuffer.__init__(self, self.config.memory, self.config.batch_size, self.config.get_dtype("np"))

    def setup(self, register_add: bool = True, register_sample: bool = True) -> None:
        if register_add:
            self.register_worker_func_custom(self.add, self.serialize)
        if register_sample:
            self.register_trainer_recv_func(self.sample)
            self.register_trainer_send_func(self.update)
