"""The three-role multiprocessing topology: actors -> MEMORY process -> learner (srl/base/run/play_mp_memory.py:253-796), which
`Runner.train_mp(enable_mp_memory=True)` -- the reference's default -- selects.

In `play_mp` the learner process also owns the replay: a helper thread feeds it and `train()` samples it in between, all under
the GIL.  Here the replay lives in a process of its own (`_memory_server`) that does three things in turn, forever:
    1. take one actor item off the ingest queue and run the memory's registered worker function on it;
    2. for every registered trainer-recv function (`sample`): while fewer than `prefetch_depth` (reference: 5) of its batches are
       waiting, draw one and ship it to the learner -- sampling and (de)compression overlap the learner's GPU work;
    3. take one trainer-send call (`update(update_args, priorities, step)`) off the bounded write-back queue (reference: 100
       calls) and apply it.
The learner process (the CALLING process: it owns the trainer's GPU) sees a `_PrefetchedMemory` in place of the memory: its
recv functions pop a batch a pump thread has already unpickled (None while nothing is waiting -> `train()` returns untrained, as
with a cold memory), its send functions enqueue for step 3 and block while the write-back queue is full.
Actors are play_mp's (`_run_actor`): only the queue they write to differs.  With a device-backed memory (the proportional
sum-tree of libsrlx) the memory process opens its own HIP context on the same GPU.
"""
import ctypes
import logging
import multiprocessing as mp
import collections
import pickle
import queue as pyqueue
import threading
import time
import traceback
from dataclasses import dataclass, field
from typing import Any, Dict, List

from simple_distributed_rl_amd.base.context import RunStateTrainer
from simple_distributed_rl_amd.base.run import play_mp
from simple_distributed_rl_amd.base.run.play_mp import MpConfig

logger = logging.getLogger(__name__)


@dataclass
class MemoryLink:
    prefetch_depth: int = 5      # mem_to_train_queue_capacity (play_mp_memory.py:38)
    writeback_backlog: int = 100  # train_to_mem_queue_capacity (:39)
    return_memory: bool = False


class _Counter:
    """A shared int with its lock (queue sizes that both ends can read without touching the queue)."""

    def __init__(self, ctx):
        self.v = ctx.Value(ctypes.c_int, 0)

    def add(self, d: int):
        with self.v.get_lock():
            self.v.value += d

    @property
    def value(self) -> int:
        return self.v.value


# ---------------------------------------------------------------------------------------------
# memory process
# ---------------------------------------------------------------------------------------------
def _memory_server(cfg_blob: bytes, link: MemoryLink, q_ingest, n_ingest, q_batches, n_batches: List, q_writeback, n_writeback, end_signal, memory_dat, q_final):
    try:
        from simple_distributed_rl_amd.base.env.registration import make as make_env

        mp_cfg: MpConfig = pickle.loads(cfg_blob)
        c = mp_cfg.context
        c.rl_config.setup(make_env(c.env_config))
        c.setup_device()
        memory = c.rl_config.make_memory()
        if memory_dat is not None:
            memory.restore(memory_dat)
        ingest = memory.get_worker_funcs()
        draw = memory.get_trainer_recv_funcs()
        apply_ = memory.get_trainer_send_funcs()
        assert len(draw) == len(n_batches)
        info: Dict[str, Any] = {"memory": memory, "act_to_mem": 0, "mem_to_train_list": [0] * len(draw), "train_to_mem": 0}
        listeners = [cb for cb in mp_cfg.callbacks if hasattr(cb, "on_memory")]
        for cb in mp_cfg.callbacks:
            cb.on_memory_start(c, info)
        while not end_signal.value:
            busy = False
            try:  # 1. one actor item
                name, raw, custom = q_ingest.get_nowait()
                n_ingest.add(-1)
                if custom:
                    ingest[name][0](*raw, serialized=True)
                else:
                    args, kwargs = pickle.loads(raw)
                    ingest[name][0](*args, **kwargs)
                info["act_to_mem"] += 1
                busy = True
            except pyqueue.Empty:
                pass
            for i, f in enumerate(draw):  # 2. keep the learner's prefetch queue topped up
                if n_batches[i].value < link.prefetch_depth:
                    batch = f()
                    if batch is not None:
                        q_batches.put(pickle.dumps((i, batch)))
                        n_batches[i].add(1)
                        info["mem_to_train_list"][i] += 1
                        busy = True
            try:  # 3. one write-back call
                name, raw = q_writeback.get_nowait()
                args, kwargs = pickle.loads(raw)
                apply_[name](*args, **kwargs)
                n_writeback.add(-1)
                info["train_to_mem"] += 1
                busy = True
            except pyqueue.Empty:
                pass
            for cb in listeners:
                cb.on_memory(c, info)
            if not busy:
                time.sleep(0.001)
        for cb in mp_cfg.callbacks:
            cb.on_memory_end(c, info)
        if link.return_memory:
            q_final.put(memory.backup(compress=True))
    except Exception:
        traceback.print_exc()
        raise
    finally:
        end_signal.value = True


# ---------------------------------------------------------------------------------------------
# learner side
# ---------------------------------------------------------------------------------------------
class _PrefetchedMemory:
    """What the trainer plugin holds instead of the memory (reference: _TrainerRLMemoryInterceptor, :361-424)."""

    def __init__(self, base_memory, link: MemoryLink, n_batches: List, q_writeback, n_writeback, end_signal):
        self._base, self._link, self._end = base_memory, link, end_signal
        self._n_batches, self._q_wb, self._n_wb = n_batches, q_writeback, n_writeback
        self._waiting: List[collections.deque] = []  # FIFO like the reference's mem_to_train queue: the oldest prefetched batch is trained on first
        self._lock = threading.Lock()
        for i, f in enumerate(base_memory.get_trainer_recv_funcs()):
            self._waiting.append(collections.deque())
            setattr(self, f.__name__, self._make_pop(i))
        for name in base_memory.get_trainer_send_funcs():
            setattr(self, name, self._make_send(name))
        self.received = 0

    def _make_pop(self, i: int):
        def pop(*args, **kwargs):
            with self._lock:
                if not self._waiting[i]:
                    return None
                batch = self._waiting[i].popleft()  # arrival order: update_args and the beta step of a batch must not go stale at the bottom of a stack
            self._n_batches[i].add(-1)
            return batch

        return pop

    def _make_send(self, name: str):
        def send(*args, **kwargs):
            blob = pickle.dumps((args, kwargs))
            t0 = time.time()
            while not self._end.value:
                if self._n_wb.value < self._link.writeback_backlog:
                    self._q_wb.put((name, blob))
                    self._n_wb.add(1)
                    return
                if time.time() - t0 > 9:  # :407-412: give up on this call rather than stall the learner for ever
                    logger.info("write-back queue full (%d): dropping one %s call", self._n_wb.value, name)
                    return
                time.sleep(0.01)

        return send

    def deliver(self, i: int, batch):
        with self._lock:
            self._waiting[i].append(batch)
        self.received += 1

    def length(self) -> int:
        return sum(len(w) for w in self._waiting)

    def __getattr__(self, item):  # config, batch_size, ...
        return getattr(self._base, item)


def _batch_pump(client: _PrefetchedMemory, q_batches, end_signal, share: dict):
    try:
        while not end_signal.value:
            try:
                blob = q_batches.get(timeout=0.1)
            except pyqueue.Empty:
                continue
            i, batch = pickle.loads(blob)
            client.deliver(i, batch)
            share["recv"] += 1
    except Exception:
        share["error"] = traceback.format_exc()
        end_signal.value = True


def train(mp_cfg: MpConfig, parameter, memory, link: MemoryLink = None):
    from simple_distributed_rl_amd.base.run.sequence import play_trainer_only

    link = link or MemoryLink()
    context = mp_cfg.context
    context.check_context_parameter()
    ctx = mp.get_context("spawn")
    manager = ctx.Manager()
    q_ingest, q_batches, q_writeback, q_final = manager.Queue(), ctx.Queue(), ctx.Queue(), ctx.Queue()
    n_ingest_raw = ctx.Value(ctypes.c_int, 0)
    end_signal = ctx.Value(ctypes.c_bool, False)
    n_batches = [_Counter(ctx) for _ in memory.get_trainer_recv_funcs()]
    n_writeback = _Counter(ctx)

    board = manager.dict()
    board["params"] = pickle.dumps((0, parameter.backup(serialized=True)))
    actor_ctx = context.copy()
    actor_ctx.callbacks = []
    blob = pickle.dumps(MpConfig(actor_ctx, list(mp_cfg.callbacks), mp_cfg.queue_capacity, mp_cfg.trainer_parameter_send_interval,
                                 mp_cfg.actor_parameter_sync_interval, mp_cfg.polling_interval))
    memory_dat = memory.backup(compress=True) if memory.length() > 0 else None
    server = ctx.Process(target=_memory_server, args=(blob, link, q_ingest, _SharedCounter(n_ingest_raw), q_batches, n_batches, q_writeback, n_writeback, end_signal,
                                                      memory_dat, q_final), daemon=True)
    actors = [ctx.Process(target=play_mp._run_actor, args=(blob, q_ingest, n_ingest_raw, board, i, end_signal), daemon=True) for i in range(context.actor_num)]
    server.start()
    [p.start() for p in actors]

    client = _PrefetchedMemory(memory, link, n_batches, q_writeback, n_writeback, end_signal)
    share = {"train_count": 0, "sync": 0, "recv": 0, "error": ""}
    t_pump = threading.Thread(target=_batch_pump, args=(client, q_batches, end_signal, share), daemon=True)
    t_par = threading.Thread(target=play_mp._parameter_communicate, args=(parameter, board, end_signal, share, mp_cfg.trainer_parameter_send_interval), daemon=True)
    t_pump.start()
    t_par.start()
    tc = context.copy()
    tc.run_name = "trainer"
    tc.callbacks = list(context.callbacks) + [play_mp._TrainerInterrupt(end_signal, share)]
    trainer = context.rl_config.make_trainer(parameter, client)
    state = None
    try:
        state = play_trainer_only(tc, trainer, RunStateTrainer())
    finally:
        end_signal.value = True
        t_pump.join(timeout=5)
        t_par.join(timeout=5)
        if link.return_memory:
            try:
                memory.restore(q_final.get(timeout=60))
            except pyqueue.Empty:
                logger.warning("the memory process did not hand its contents back")
        for p in actors + [server]:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        manager.shutdown()
    if share["error"]:
        raise RuntimeError("learner helper thread failed:\n" + share["error"])
    bad = [p.exitcode for p in actors + [server] if p.exitcode not in (0, None, -15)]
    if bad:
        raise RuntimeError(f"actor / memory process exited with {bad}")
    return state


class _SharedCounter:
    """`_Counter` interface over an existing shared Value (picklable across the spawn boundary)."""

    def __init__(self, value):
        self.v = value

    def add(self, d: int):
        with self.v.get_lock():
            self.v.value += d

    @property
    def value(self) -> int:
        return self.v.value
