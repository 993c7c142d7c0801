"""ProportionalMemory on the MI355X: the sum-tree lives in HBM behind libsrlx.so.

Drop-in for the reference's srl/rl/memories/priority_memories/proportional_memory.py:95-205
(`ProportionalMemory`) and for its pybind11 twin (cpp_module/src/proportional_memory.cpp:250-275):
same constructor arguments, same `clear/length/add/sample/update/backup/restore`, same return
values (`sample` -> (batches, float64 weights, list of TREE indices)).  Selectable from an
unmodified reference config through
    cfg.memory.set_custom("simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory:ProportionalMemory", {...})
(srl/rl/memories/priority_replay_buffer.py:111-117,149-152).

Like the C++ twin, the opaque `batch` objects stay on the host in a list; the device owns the
priorities.  Random numbers: the reference calls `random.random()` once per descent attempt
(:147); this class draws the same stream from Python's `random` and leaves the generator in
exactly the state the reference would (rejected draws included), so a seeded run interleaves
with other `random` users identically.

Round 6 (the seam at the speed of what it replaces; reference loop: tests/quick/rl/memories/speedtest.py:15-58): `add` only queues -- the tree is observed by
`sample / update / backup / restore / max_priority` alone, where the queue is flushed as ONE `srlx_per_add(n)` (consecutive adds of one priority kind; `add` does
not move `max_priority` in the reference, proportional_memory.py:120-129, so a queued `priority=None` add resolves to the same value as an immediate one: the
class mirrors `max_priority` on the host when it transforms priorities itself); `update` copies its arguments into a device-visible pinned slot and returns
without synchronising; `sample` reads its uniforms from and writes its results to such a slot (one synchronisation, no copy commands) -- `on_device = 2` of
include/srlx.h.
"""
import ctypes
import random
import threading
from array import array
from typing import Any, List, Optional

import numpy as np

from simple_distributed_rl_amd import _native as N

from .imemory import IPriorityMemory


class ProportionalMemory(IPriorityMemory):
    def __init__(
        self,
        capacity: int,
        alpha: float = 0.6,
        beta_initial: float = 0.4,
        beta_steps: int = 1_000_000,
        has_duplicate: bool = True,
        epsilon: float = 0.0001,
        device: int = 0,
        host_transform: bool = True,
    ):
        """host_transform=True evaluates (|p|+eps)**alpha with the host's Python/numpy exactly as
        proportional_memory.py:124,172 does (so the tree is bit-identical to the reference run on
        this host); False ships raw priorities and lets the kernel transform them."""
        self.capacity = int(capacity)
        self.alpha = alpha
        self.beta_initial = beta_initial
        self.beta_steps = beta_steps
        self.has_duplicate = has_duplicate
        self.epsilon = epsilon
        self.device = int(device)
        self.host_transform = host_transform
        self._lock = threading.Lock()  # play_mp.py:248-286 calls add() and sample() from two threads
        self._bufs = {}
        self._lib = N.lib()
        h = N.c_p()
        N.check(
            self._lib.srlx_per_create(
                ctypes.byref(h),
                self.capacity,
                float(alpha),
                float(beta_initial),
                float(beta_steps),
                int(bool(has_duplicate)),
                float(epsilon),
                self.device,
            )
        )
        self._h = h
        self.data: List[Any] = [None] * self.capacity
        self._write = 0
        self._queue: List[float] = []  # values of queued adds (one priority kind per run)
        self._queue_kind = None
        self._max_host = 1.0  # mirror of max_priority (exact while host_transform: every value that can raise it passes through `update` here)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.srlx_per_destroy(h)
            self._h = None

    # ---- IPriorityMemory -----------------------------------------------------------------
    def clear(self) -> None:
        with self._lock:
            self._queue, self._queue_kind = [], None
            N.check(self._lib.srlx_per_clear(self._h, None))
            self.data = [None] * self.capacity
            self._write = 0
            self._max_host = 1.0

    def length(self) -> int:
        return min(self.capacity, int(self._lib.srlx_per_length(self._h)) + len(self._queue))

    def _flush(self) -> None:
        """The queued adds as ONE launch (call with the lock held)."""
        q, kind = self._queue, self._queue_kind
        if not q:
            return
        self._queue, self._queue_kind = [], None
        for i in range(0, len(q), self.capacity):  # (srlx_per_add takes at most `capacity` items per call)
            part = q[i : i + self.capacity]
            if kind == N.PRIO_NONE:
                N.check(self._lib.srlx_per_add(self._h, len(part), None, N.PRIO_NONE, 0, None))
            else:
                v = np.asarray(part, np.float64)
                N.check(self._lib.srlx_per_add(self._h, len(part), N.np_ptr(v), kind, 2, None))

    def add(self, batch: Any, priority: Optional[float] = None, _restore_skip: bool = False) -> None:
        with self._lock:
            self.data[self._write] = batch
            self._write = (self._write + 1) % self.capacity
            if priority is None:
                if self.host_transform:  # max_priority is mirrored here: the add is a plain value
                    kind, priority = N.PRIO_RAW, self._max_host
                else:
                    kind, priority = N.PRIO_NONE, 0.0
            else:
                priority = float(priority)  # see oracle/gen_golden.py trace (4): numpy scalars are widened
                if _restore_skip:
                    kind = N.PRIO_RAW
                elif self.host_transform:
                    priority = (abs(priority) + self.epsilon) ** self.alpha  # proportional_memory.py:124
                    kind = N.PRIO_RAW
                else:
                    kind = N.PRIO_F64
            if self._queue and kind != self._queue_kind:
                self._flush()
            self._queue_kind = kind
            self._queue.append(priority)
            if len(self._queue) >= 2048:  # (a pinned slot holds 2048 float64 values)
                self._flush()

    def _buffers(self, batch_size: int):
        """Per batch size: the result arrays and their ctypes pointers (building a pointer object costs a microsecond; a call needs six)."""
        b = self._bufs.get(batch_size)
        if b is None:
            idx, w, slots, used = np.empty(batch_size, np.int64), np.empty(batch_size, np.float64), np.empty(batch_size, np.int64), N.c_i64(0)
            b = self._bufs[batch_size] = (idx, w, slots, used, N.np_ptr(idx), N.np_ptr(w), N.np_ptr(slots), ctypes.byref(used))
        return b

    def sample(self, batch_size: int, step: int):
        batch_size = int(batch_size)
        idx, w, slots, used, p_idx, p_w, p_slots, p_used = self._buffers(batch_size)
        with self._lock:
            # up to 16 queued adds of one kind ride INSIDE the sampling launch (srlx_per_sample_after_adds); a longer queue is flushed by a launch of its own first
            q_kind = self._queue_kind
            if len(self._queue) > 16 or (self._queue and q_kind not in (N.PRIO_RAW, N.PRIO_NONE)):
                self._flush()
            adds, self._queue, self._queue_kind = self._queue, [], None
            add_arr = array("d", adds) if (adds and q_kind == N.PRIO_RAW) else None
            add_ptr = add_arr.buffer_info()[0] if add_arr is not None else None
            # Uniforms: the reference calls random.random() once per descent attempt (:147).  Without rejected draws a batch consumes exactly `batch_size` of them: the
            # first attempt takes the generator's raw output -- 2 x 32 bits per uniform, what `batch_size` calls of random.random() would have consumed, in one
            # getrandbits() -- and libsrlx turns the words into the same doubles (srlx_per_sample_after_adds_mt).  No snapshot of the generator either
            # (random.getstate() copies 625 words: more host time than the kernel runs); only when the kernel reports that rejections ate the uniforms is the state
            # captured -- AFTER the ones consumed so far -- before more are drawn, so that an over-provisioned retry can be rolled back to what the reference consumes.
            st = N.ERR_UNIFORMS_EXHAUSTED
            raw = None
            if batch_size <= 8192:
                raw = random.getrandbits(64 * batch_size).to_bytes(8 * batch_size, "little")
                st = self._lib.srlx_per_sample_after_adds_mt(self._h, len(adds), add_ptr, q_kind if adds else N.PRIO_NONE, batch_size, int(step), raw, batch_size, p_idx, p_w,
                                                             None, p_used, p_slots, None)
                adds = []  # (applied by the first attempt, whatever the draw's outcome)
            if st != N.OK:
                self._sample_slow(batch_size, step, raw, adds, add_arr, q_kind, idx, w, used)
                np.subtract(idx, self.capacity - 1, out=slots)
            data = self.data
            indices = idx.tolist()
            batches = [data[i] for i in slots.tolist()]
        return batches, w.copy(), indices

    def _sample_slow(self, batch_size, step, raw, adds, add_arr, q_kind, idx, w, used):
        """Rejected draws ate the first attempt's uniforms (or the batch is larger than one sampling launch takes): the list-based loop (lock held)."""
        cap = 8192 if not self.has_duplicate else 9999 * batch_size  # without duplicates one call walks at most 8192 uniforms
        if raw is not None:  # the uniforms the first attempt consumed: random.random() of consecutive generator outputs (a >> 5, b >> 6)
            wd = np.frombuffer(raw, dtype="<u4")
            drawn = (((wd[0::2] >> 5).astype(np.float64) * 67108864.0 + (wd[1::2] >> 6).astype(np.float64)) * (1.0 / 9007199254740992.0)).tolist()
            state, base = random.getstate(), len(drawn)
            drawn.extend(random.random() for _ in range(min(len(drawn) + 16, cap - len(drawn))))
        else:
            drawn = [random.random() for _ in range(batch_size)]
            state, base = None, 0  # state: the generator AFTER `base` of the drawn uniforms
        forced = False
        add_np = np.asarray(add_arr, np.float64) if add_arr is not None else None
        while True:
            m = len(drawn)
            u = np.asarray(drawn, np.float64)
            if m <= 8192:
                st = self._lib.srlx_per_sample_after_adds(self._h, len(adds), N.np_ptr(add_np) if (adds and add_np is not None) else None, q_kind if adds else N.PRIO_NONE,
                                                          batch_size, int(step), N.np_ptr(u), m, N.np_ptr(idx), N.np_ptr(w), None, ctypes.byref(used), None)
                adds = []
            else:
                if adds:
                    N.check(self._lib.srlx_per_add(self._h, len(adds), N.np_ptr(add_np) if add_np is not None else None, q_kind, 2, None))
                    adds = []
                st = self._lib.srlx_per_sample(self._h, batch_size, int(step), None, N.np_ptr(u), m, N.np_ptr(idx), N.np_ptr(w), None, ctypes.byref(used), 2, None)
            if st == N.ERR_UNIFORMS_EXHAUSTED and m < cap:  # rejected draws ate the uniforms: again with more (the same prefix: the same walk up to there)
                state, base = random.getstate(), m
                drawn.extend(random.random() for _ in range(min(m + 16, cap - m)))
                continue
            if st == N.ERR_UNIFORMS_EXHAUSTED and not self.has_duplicate and not forced:
                # fewer distinct non-zero leaves than the batch needs: the reference gives up on a draw after 9999 tries and takes
                # it, duplicate or not (:146-158); here the batch is completed with duplicates from the same uniforms
                N.check(self._lib.srlx_per_set_has_duplicate(self._h, 1))
                forced = True
                continue
            if forced:
                N.check(self._lib.srlx_per_set_has_duplicate(self._h, 0))
            N.check(st)
            break
        if used.value != len(drawn):  # the retry drew more than the walk consumed: leave `random` where the reference would
            random.setstate(state)
            for _ in range(used.value - base):
                random.random()

    def update(self, indices: List[Any], priorities: np.ndarray) -> None:
        n = len(indices)
        if n == 0:
            return
        idx = array("q", indices)  # (a C loop over the list: a third of np.ascontiguousarray's time at 64 entries)
        # (a list of Python floats -- the reference's speedtest -- through array('d'): a C loop, a third of np.asarray's time; same float64 values)
        pr = np.frombuffer(array("d", priorities), np.float64) if type(priorities) is list else np.asarray(priorities)
        if self.host_transform:
            pr = np.ascontiguousarray((np.abs(pr) + self.epsilon) ** self.alpha, dtype=np.float64)  # :172
            kind = N.PRIO_RAW
            mx = float(pr[:n].max())  # :176-177 (the device raises its own copy the same way)
            if self._max_host < mx:
                self._max_host = mx
        elif pr.dtype == np.float32:
            pr = np.ascontiguousarray(pr)
            kind = N.PRIO_F32
        else:
            pr = np.ascontiguousarray(pr, dtype=np.float64)
            kind = N.PRIO_F64
        if pr.shape[0] < n:
            raise IndexError("priorities shorter than indices")
        with self._lock:
            self._flush()
            N.check(self._lib.srlx_per_update(self._h, n, idx.buffer_info()[0], pr.ctypes.data, kind, 2, None))  # asynchronous: ordered before every later call

    def backup(self):
        """Same list layout as proportional_memory.py:179-187."""
        with self._lock:
            self._flush()
            mp, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
            tree = np.empty(2 * self.capacity - 1, np.float64)
            N.check(self._lib.srlx_per_backup(self._h, ctypes.byref(mp), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
            return [self.capacity, mp.value, size.value, write.value, tree.tolist(), self.data[:]]

    def restore(self, data) -> None:
        with self._lock:
            self._queue, self._queue_kind = [], None  # (restore replaces the tree: :189-205)
            if self.capacity == data[0]:  # :190-194
                self._max_host = float(data[1])
                tree = np.ascontiguousarray(data[4], dtype=np.float64)
                N.check(self._lib.srlx_per_restore(self._h, float(data[1]), int(data[2]), int(data[3]), N.np_ptr(tree)))
                self.data = list(data[5][:])
                self._write = int(data[3])
            else:  # :195-205 -- clear, then re-add the first `size` leaves with _restore_skip
                old_cap, old_size = int(data[0]), int(data[2])
                tree = np.ascontiguousarray(data[4], dtype=np.float64)
                N.check(self._lib.srlx_per_restore_resized(self._h, old_cap, old_size, N.np_ptr(tree)))
                self._max_host = 1.0  # (clear() of :196; the re-adds use _restore_skip and leave it alone)
                self.data = [None] * self.capacity
                self._write = 0
                for i in range(old_size):
                    self.data[self._write] = data[5][i]
                    self._write = (self._write + 1) % self.capacity

    # ---- extras (not in the reference interface) --------------------------------------------
    @property
    def max_priority(self) -> float:
        with self._lock:
            self._flush()
        mp, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
        N.check(self._lib.srlx_per_backup(self._h, ctypes.byref(mp), ctypes.byref(size), ctypes.byref(write), None))
        return mp.value

    def tree_array(self) -> np.ndarray:
        return np.asarray(self.backup()[4])
