"""Multiprocessing actor/learner topology of the plugin path (srl/base/run/play_mp.py:54-642).

N actor processes run the sequential loop with the trainer disabled; their `memory.add` is intercepted,
serialised with the memory's own serialiser and put on a bounded queue (back-pressure: the actor sleeps
while the learner is `queue_capacity` items behind, play_mp.py:76-118).  The learner runs in the CALLING
process (it owns the GPU): one thread drains the queue into the memory (`add(..., serialized=True)`,
:248-286), one thread publishes `(train_count, parameter.backup(serialized=True))` on a manager board every
`trainer_parameter_send_interval` seconds (:289-318), the main thread trains back to back
(core_train_only).  Actors pull the board every `actor_parameter_sync_interval` seconds (:121-165).
Liveness: actors stop when the end signal is set or the parent dies; the learner raises if an actor exits
with a non-zero code (:572-635).

This is the reference-compatible transport for existing single-environment plugins; the MI355X-native
multi-GPU transport (RCCL gather/broadcast of device tensors) is device/dist.py.
"""
import ctypes
import logging
import multiprocessing as mp
import pickle
import queue as pyqueue
import threading
import time
import traceback
from dataclasses import dataclass, field
from typing import Any, List

from simple_distributed_rl_amd.base.context import RunContext, RunStateTrainer
from simple_distributed_rl_amd.base.run.callback import RunCallback

logger = logging.getLogger(__name__)


@dataclass
class MpConfig:
    context: RunContext
    callbacks: List[RunCallback] = field(default_factory=list)
    queue_capacity: int = 1000
    trainer_parameter_send_interval: float = 1
    actor_parameter_sync_interval: float = 1
    polling_interval: float = 0.2


# ---------------------------------------------------------------------------------------------
# actor side
# ---------------------------------------------------------------------------------------------
class _ActorMemoryInterceptor:
    """Stands in for the memory inside an actor: every registered worker function becomes
    "serialise + enqueue" (play_mp.py:54-118)."""

    def __init__(self, memory, remote_queue, remote_qsize, end_signal, queue_capacity: int, actor_id: int):
        self._memory = memory
        self._q, self._qsize, self._end = remote_queue, remote_qsize, end_signal
        self._cap, self._actor_id = queue_capacity, actor_id
        self.sent = 0
        for name, (func, serialize_func) in memory.get_worker_funcs().items():
            setattr(self, name, self._make(name, serialize_func))

    def _make(self, name, serialize_func):
        def _send(*args, **kwargs):
            while self._qsize.value >= self._cap and not self._end.value:  # back-pressure
                time.sleep(0.05)
            raw = pickle.dumps((args, kwargs)) if serialize_func is None else serialize_func(*args, **kwargs)
            if serialize_func is not None and not isinstance(raw, tuple):
                raw = (raw,)
            self._q.put((name, raw, serialize_func is not None))
            with self._qsize.get_lock():
                self._qsize.value += 1
            self.sent += 1

        return _send

    def length(self) -> int:
        return self._qsize.value

    def __getattr__(self, item):  # config etc.
        return getattr(self._memory, item)


class _ActorInterrupt(RunCallback):
    def __init__(self, board, parameter, end_signal, sync_interval: float):
        self.board, self.parameter, self.end_signal, self.interval = board, parameter, end_signal, sync_interval
        self.t0 = time.time()
        self.last_count = -1

    def on_step_end(self, context, state, **kwargs) -> bool:
        if self.end_signal.value:
            return True
        parent = mp.parent_process()
        if parent is not None and not parent.is_alive():
            return True
        if time.time() - self.t0 < self.interval:
            return False
        self.t0 = time.time()
        dat = self.board.get("params")
        if dat is not None:
            count, params = pickle.loads(dat)
            if count != self.last_count and params is not None:
                self.parameter.restore(params, from_serialized=True)
                self.last_count = count
                state.sync_actor += 1
        return False


def _run_actor(cfg_blob: bytes, remote_queue, remote_qsize, board, actor_id: int, end_signal):
    try:
        from simple_distributed_rl_amd.base.env.registration import make as make_env
        from simple_distributed_rl_amd.base.run.sequence import play

        mp_cfg: MpConfig = pickle.loads(cfg_blob)
        c = mp_cfg.context
        c.run_name = "actor"
        c.actor_id = actor_id
        c.device = "CPU" if isinstance(c.actor_devices, str) and c.actor_devices.upper() in ("CPU", "AUTO") else (
            c.actor_devices if isinstance(c.actor_devices, str) else c.actor_devices[actor_id % len(c.actor_devices)])
        c.disable_trainer = True
        env = make_env(c.env_config)
        c.rl_config.setup(env)
        c.setup_device()
        parameter = c.rl_config.make_parameter()
        dat = board.get("params")
        if dat is not None:
            params = pickle.loads(dat)[1]
            if params is not None:
                parameter.restore(params, from_serialized=True)
        memory = _ActorMemoryInterceptor(c.rl_config.make_memory(), remote_queue, remote_qsize, end_signal, mp_cfg.queue_capacity, actor_id)
        c.callbacks = list(mp_cfg.callbacks) + [_ActorInterrupt(board, parameter, end_signal, mp_cfg.actor_parameter_sync_interval)]
        c.timeout = 0
        c.max_train_count = 0
        c.max_steps = 0
        c.max_episodes = 0
        c.training = True
        c.distributed = True
        c.check_context_parameter = lambda *a, **k: None  # actors stop on the end signal
        worker = c.rl_config.make_worker(env, parameter, memory)
        play(c, env, worker, trainer=None)
    except Exception:
        traceback.print_exc()
        raise
    finally:
        end_signal.value = True


# ---------------------------------------------------------------------------------------------
# learner side
# ---------------------------------------------------------------------------------------------
def _memory_communicate(memory, remote_queue, remote_qsize, end_signal, share: dict):
    funcs = memory.get_worker_funcs()
    try:
        while not end_signal.value:
            try:
                name, raw, custom = remote_queue.get(timeout=0.1)
            except pyqueue.Empty:
                continue
            with remote_qsize.get_lock():
                remote_qsize.value -= 1
            if custom:
                funcs[name][0](*raw, serialized=True)
            else:
                args, kwargs = pickle.loads(raw)
                funcs[name][0](*args, **kwargs)
            share["recv"] += 1
    except Exception:
        share["error"] = traceback.format_exc()
        end_signal.value = True


def _parameter_communicate(parameter, board, end_signal, share: dict, interval: float):
    try:
        while not end_signal.value:
            time.sleep(interval)
            board["params"] = pickle.dumps((share["train_count"], parameter.backup(serialized=True)))
            share["sync"] += 1
    except Exception:
        share["error"] = traceback.format_exc()
        end_signal.value = True


class _TrainerInterrupt(RunCallback):
    def __init__(self, end_signal, share):
        self.end_signal, self.share = end_signal, share

    def on_train_after(self, context, state, **kwargs) -> bool:
        self.share["train_count"] = state.train_count
        state.sync_trainer = self.share["sync"]
        state.trainer_recv_q = self.share["recv"]
        if not state.is_step_trained:
            time.sleep(0.005)  # warm-up: let the queue thread run
        return bool(self.end_signal.value)


def train(mp_cfg: MpConfig, parameter, memory):
    from simple_distributed_rl_amd.base.run.sequence import play_trainer_only

    context = mp_cfg.context
    context.check_context_parameter()
    ctx = mp.get_context("spawn")  # play_mp.py:508-515
    manager = ctx.Manager()
    remote_queue = manager.Queue()
    remote_qsize = ctx.Value(ctypes.c_int, 0)
    end_signal = ctx.Value(ctypes.c_bool, False)
    board = manager.dict()
    board["params"] = pickle.dumps((0, parameter.backup(serialized=True)))

    actor_ctx = context.copy()
    actor_ctx.callbacks = []
    blob = pickle.dumps(MpConfig(actor_ctx, list(mp_cfg.callbacks), mp_cfg.queue_capacity, mp_cfg.trainer_parameter_send_interval,
                                 mp_cfg.actor_parameter_sync_interval, mp_cfg.polling_interval))
    actors = [ctx.Process(target=_run_actor, args=(blob, remote_queue, remote_qsize, board, i, end_signal), daemon=True) for i in range(context.actor_num)]
    [p.start() for p in actors]

    share = {"train_count": 0, "sync": 0, "recv": 0, "error": ""}
    t_mem = threading.Thread(target=_memory_communicate, args=(memory, remote_queue, remote_qsize, end_signal, share), daemon=True)
    t_par = threading.Thread(target=_parameter_communicate, args=(parameter, board, end_signal, share, mp_cfg.trainer_parameter_send_interval), daemon=True)
    t_mem.start()
    t_par.start()

    tc = context.copy()
    tc.run_name = "trainer"
    tc.callbacks = list(context.callbacks) + [_TrainerInterrupt(end_signal, share)]
    trainer = context.rl_config.make_trainer(parameter, memory)
    state = None
    try:
        state = play_trainer_only(tc, trainer, RunStateTrainer())
    finally:
        end_signal.value = True
        t_mem.join(timeout=5)
        t_par.join(timeout=5)
        for p in actors:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        manager.shutdown()
    if share["error"]:
        raise RuntimeError("learner helper thread failed:\n" + share["error"])
    bad = [p.exitcode for p in actors if p.exitcode not in (0, None, -15)]
    if bad:
        raise RuntimeError(f"actor process exited with {bad}")  # play_mp.py:623-635
    return state
