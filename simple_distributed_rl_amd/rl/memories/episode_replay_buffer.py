"""EpisodeReplayBuffer (srl/rl/memories/episode_replay_buffer.py:10-191): whole episodes (lists of opaque step objects)
with window sampling for sequence models -- `sample` draws an episode then a start so that prefix + 1 + suffix steps fit
(:86-117), `sample_steps` hands out one whole episode (:119-126), `sample_sequential` streams batch_size parallel
time-ordered windows advancing by `sequential_stride` (:128-179).  Capacity counts SAMPLEABLE window starts, the oldest
episodes leave first (:67-75).  Host-side like the reference (opaque Python steps); the `random` call order is the
reference's, so a seeded run draws the same windows."""
import logging
import pickle
import random
import zlib
from typing import Any, Callable, List, Optional

logger = logging.getLogger(__name__)


class EpisodeReplayBuffer:
    def __init__(self, batch_size: int = 32, capacity: int = 100_000, warmup_size: int = 1000, compress: bool = True, compress_level: int = -1,
                 prefix_size: int = 0, suffix_size: int = 0, skip_head: int = 0, skip_tail: int = 0, sequential_stride: int = 1):
        self.batch_size, self.capacity, self.warmup_size = batch_size, capacity, warmup_size
        self.compress, self.compress_level = compress, compress_level
        self.prefix_size, self.suffix_size = prefix_size, suffix_size
        self.skip_head, self.skip_tail, self.sequential_stride = skip_head, skip_tail, sequential_stride
        self.buffer = []  # (episode or its compressed bytes, sampleable starts)
        self.total_size = 0
        self._streams = [[] for _ in range(batch_size)]
        if not (warmup_size <= capacity):
            raise ValueError(f"assert {warmup_size} <= {capacity}")
        if not (0 < batch_size <= warmup_size):
            raise ValueError(f"assert 0 < {batch_size} <= {warmup_size}")

    @property
    def batch_length(self) -> int:
        return self.prefix_size + 1 + self.suffix_size

    def length(self) -> int:
        return self.total_size

    def is_warmup(self) -> bool:
        return self.total_size < self.warmup_size

    def _open(self, stored):
        return pickle.loads(zlib.decompress(stored)) if self.compress else stored

    def add(self, steps: List[Any], size: int = 0, serialized: bool = False) -> None:
        if serialized:  # already pickled (+ compressed): keep it packed if this buffer compresses (:58-60)
            if not self.compress:
                steps = pickle.loads(steps)
        else:
            size = len(steps)
            if self.compress:
                steps = zlib.compress(pickle.dumps(steps), level=self.compress_level)
        starts = max(0, size - (self.batch_length + self.skip_head + self.skip_tail) + 1)
        self.total_size += starts
        self.buffer.append((steps, starts))
        while self.total_size > self.capacity:
            _, dropped = self.buffer.pop(0)
            self.total_size -= dropped

    def serialize(self, steps) -> Any:
        size = len(steps)
        data = pickle.dumps(steps)
        return (zlib.compress(data, level=self.compress_level) if self.compress else data), size

    def sample(self, batch_size: int = -1, prefix_size: int = -1, suffix_size: int = -1, skip_head: int = -1, skip_tail: int = -1):
        if self.total_size < self.warmup_size:
            return None
        batch_size = self.batch_size if batch_size == -1 else batch_size
        prefix_size = self.prefix_size if prefix_size == -1 else prefix_size
        suffix_size = self.suffix_size if suffix_size == -1 else suffix_size
        skip_head = self.skip_head if skip_head == -1 else skip_head
        skip_tail = self.skip_tail if skip_tail == -1 else skip_tail
        window = prefix_size + 1 + suffix_size
        out = []
        while len(out) < batch_size:
            steps = self._open(self.buffer[random.randint(0, len(self.buffer) - 1)][0])
            last_start = len(steps) - window - skip_tail
            if len(steps) < last_start + window:  # kept from the reference (:109-111); cannot trigger for skip_tail >= 0
                logger.warning("Episode length must be equal to or greater than batch_length.")
                continue
            j = random.randint(skip_head, last_start)
            out.append(steps[j: j + window])
        return out

    def sample_steps(self, batch_size: int = -1):
        if self.total_size < self.warmup_size:
            return None
        return self._open(self.buffer[random.randint(0, len(self.buffer) - 1)][0])

    def sample_sequential(self, dummy_step: Optional[list] = None, should_drop_batch_func: Optional[Callable[[int, List[list]], bool]] = None,
                          batch_size: int = -1, batch_length: int = -1, skip_head: int = -1, skip_tail: int = -1, sequential_stride: int = -1):
        if self.total_size < self.warmup_size:
            return None
        batch_size = self.batch_size if batch_size == -1 else batch_size
        assert batch_size <= self.batch_size
        batch_length = self.batch_length if batch_length == -1 else batch_length
        skip_head = self.skip_head if skip_head == -1 else skip_head
        skip_tail = self.skip_tail if skip_tail == -1 else skip_tail
        stride = self.sequential_stride if sequential_stride == -1 else sequential_stride
        out: List[list] = []
        for i in range(batch_size):
            for _ in range(99):  # like the reference: give up on a stream whose windows keep being dropped
                while len(self._streams[i]) < batch_length:  # refill stream i with another (trimmed) episode
                    steps = self._open(self.buffer[random.randint(0, len(self.buffer) - 1)][0])
                    if len(steps) <= skip_head + skip_tail:
                        logger.warning("Episode length must be greater than skip_head + skip_tail.")
                        continue
                    steps = steps[skip_head:] if skip_tail <= 0 else steps[skip_head:-skip_tail]
                    if dummy_step is not None:
                        self._streams[i].extend([dummy_step] * i)  # de-phases the parallel streams (:163-164)
                    self._streams[i].extend(steps)
                window = self._streams[i][:batch_length]
                self._streams[i] = self._streams[i][stride:]
                if should_drop_batch_func is not None and should_drop_batch_func(i, window):
                    continue
                out.append(window)
                break
            else:
                logger.error("Failed to add batch.")
                out.append([None])
        return out

    def call_backup(self, **kwargs):
        return [self.total_size, self.buffer[:]]

    def call_restore(self, data: Any, **kwargs) -> None:
        self.total_size = data[0]
        self.buffer = data[1][:]
        self._streams = [[] for _ in range(self.batch_size)]
