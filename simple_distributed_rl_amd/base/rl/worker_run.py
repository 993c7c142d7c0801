"""WorkerRun: what a plugin `RLWorker` sees of one seat of one host environment.

Interface of srl/base/rl/worker_run.py:24-672 (the attribute and method names below are what algorithm plugins
call: `worker.state`, `worker.next_state`, `worker.reward`, `worker.add_tracking(...)`, ...).  The mechanics are
this package's own, split into three small parts:

    _Lagged        a value with one step of history -- observation, invalid actions and action each are one
    _FrameWindow   the `window_length` most recent one-step observations (bounded deque; the reference shifts a list)
    _TrackingRing  the bounded per-episode record ring behind add_tracking / get_trackings

Timing contract kept from the reference (worker_run.py:310-401): the plugin's `on_step` for transition t is
delivered lazily, at the start of the NEXT `policy()` call -- or at once when the environment reports the end
of the episode -- and while it runs the views shift by one step (`state` is the observation the action was
chosen in, `next_state` the one it led to).  The device engine does not use this class: its E environments
keep their frame history in the HBM frame ring (device/replay.py) and never build a stacked observation.
"""
from collections import deque
from typing import Any, Dict, List, Optional

from simple_distributed_rl_amd.base.context import RunContext, RunState
from simple_distributed_rl_amd.base.define import DoneTypes
from simple_distributed_rl_amd.base.exception import SRLError


class _Lagged:
    """`new` is the latest value, `old` the one before it."""

    __slots__ = ("old", "new")

    def __init__(self, initial_old, initial_new):
        self.old, self.new = initial_old, initial_new

    def push(self, value):
        self.old, self.new = self.new, value

    def view(self, delivering: bool):
        """(previous, current, next) as the plugin sees them: shifted one step back while on_step runs."""
        return (None, self.old, self.new) if delivering else (self.old, self.new, None)


class _FrameWindow:
    def __init__(self, one_step_space, length: int):
        self._space, self._length = one_step_space, length
        self.frames: deque = deque(maxlen=max(1, length))
        self.clear()

    def clear(self):
        self.frames.clear()
        self.frames.extend(self._space.get_default() for _ in range(self._length))  # zero history (worker_run.py:277)

    def absorb(self, one_step_state):
        """Newest frame in, oldest out; returns what the algorithm sees."""
        if self._length <= 1:
            return one_step_state
        self.frames.append(one_step_state)
        return self._space.encode_stack(list(self.frames))


class _TrackingRing:
    def __init__(self):
        self.limit = -1
        self.clear()

    def clear(self):
        self.rows: deque = deque(maxlen=self.limit if self.limit > 0 else None)
        self.keys: List[str] = []

    def set_limit(self, limit: int):
        self.limit = limit
        self.rows = deque(self.rows, maxlen=limit if limit > 0 else None)

    def add(self, row: Dict[str, Any]):
        self.keys.extend(k for k in row if k not in self.keys)
        self.rows.append(row)

    def column(self, key: str, size: Optional[int], dummy: Any) -> list:
        col = [r.get(key, dummy) for r in self.rows]
        if size is None:
            return col
        if size <= 0:
            return []
        return [dummy] * (size - len(col)) + col if len(col) < size else col[-size:]

    def table(self, keys: Optional[List[str]], size: int, padding: dict, pad_at: str) -> list:
        keys = self.keys if keys is None else keys
        body = [[r.get(k) for k in keys] for r in self.rows]
        if size <= 0:
            return body
        missing = size - len(body)
        if missing <= 0:
            return body[-size:]
        filler = [[padding.get(k) for k in keys] for _ in range(missing)]
        if pad_at == "head":
            return filler + body
        if pad_at == "tail":
            return body + filler
        raise ValueError(pad_at)


class WorkerRun:
    def __init__(self, worker, env):
        worker.config.setup(env, enable_log=False)
        worker._set_worker_run(self)
        self._worker, self._config, self._env = worker, worker.config, env
        self._is_setup = False
        self._context, self._run_state = RunContext(), RunState()
        self._step_in_training = 0
        self._tracking = _TrackingRing()
        self._window = _FrameWindow(self._config.observation_space_one_step, self._config.window_length)
        self._begin_episode(0, None)

    # ---- who / where ----------------------------------------------------------------------------
    worker = property(lambda self: self._worker)
    config = property(lambda self: self._config)
    env = property(lambda self: self._env)
    context = property(lambda self: self._context)
    run_state = property(lambda self: self._run_state)
    info = property(lambda self: self._worker.info)
    player_index = property(lambda self: self._seat)
    episode_seed = property(lambda self: self._episode_seed)
    distributed = property(lambda self: self._context.distributed)
    training = property(lambda self: self._context.training)
    train_only = property(lambda self: self._context.train_only)
    rollout = property(lambda self: self._context.rollout)
    actor_id = property(lambda self: self._context.actor_id)
    train_count = property(lambda self: self._run_state.train_count)
    step_in_training = property(lambda self: self._step_in_training)
    step_in_episode = property(lambda self: self._step_in_episode)

    # ---- the transition as the plugin sees it ---------------------------------------------------
    prev_state = property(lambda self: self._obs.view(self._delivering)[0])
    state = property(lambda self: self._obs.view(self._delivering)[1])
    next_state = property(lambda self: self._obs.view(self._delivering)[2])
    prev_invalid_actions = property(lambda self: self._invalid.view(self._delivering)[0])
    invalid_actions = property(lambda self: self._invalid.view(self._delivering)[1])
    next_invalid_actions = property(lambda self: self._invalid.view(self._delivering)[2])
    prev_action = property(lambda self: self._act.old)
    action = property(lambda self: self._act.new)
    reward = property(lambda self: self._reward)
    done_type = property(lambda self: self._env._done)
    done = property(lambda self: self._env._done != DoneTypes.NONE)
    terminated = property(lambda self: self._env._done == DoneTypes.TERMINATED)
    done_reason = property(lambda self: self._env.env.done_reason)

    def get_state_one_step(self, idx: int = -1):
        return self._window.frames[idx] if self._config.window_length > 1 else self._obs.new

    def get_onehot_prev_action(self):
        return self._config.action_space.get_onehot(self._act.old)

    def get_onehot_action(self, action=None):
        return self._config.action_space.get_onehot(self._act.new if action is None else action)

    def get_valid_actions(self) -> list:
        return self._config.action_space.get_valid_actions(self.invalid_actions)

    def add_invalid_actions(self, invalid_actions: list, encode: bool = False) -> None:
        extra = [self._config.action_encode(a) for a in invalid_actions] if encode else list(invalid_actions)
        self._invalid.new = list(set(self._invalid.new) | set(extra))

    def sample_action(self):
        return self._config.action_space.sample(self._invalid.new)

    def override_action(self, env_action, encode: bool = True):
        self._act.new = self._config.action_encode(env_action) if encode else env_action
        return self._act.new

    def abort_episode(self):
        self._env.abort_episode()

    # ---- lifecycle ------------------------------------------------------------------------------
    def setup(self, context: Optional[RunContext] = None, render_mode: str = "", run_state: Optional[RunState] = None):
        self._context = RunContext(self._env.config, self._config) if context is None else context
        self._run_state = RunState() if run_state is None else run_state
        self._step_in_training = 0
        self._tracking.set_limit(-1)
        self._worker.on_setup(self, self._context)
        self._is_setup = True

    def teardown(self):
        self._worker.on_teardown(self)
        self._is_setup = False

    def reset(self, player_index: int, seed: Optional[int] = None) -> None:
        if not self._is_setup:
            raise SRLError("Cannot call worker.on_reset() before calling worker.setup()")
        self._begin_episode(player_index, seed)

    def _begin_episode(self, seat: int, seed: Optional[int]):
        cfg = self._config
        self._seat, self._episode_seed = seat, seed
        self._announced = False  # on_reset not delivered yet
        self._delivering = False  # inside the plugin's on_step
        self._step_in_episode = 0
        self._obs = _Lagged(cfg.observation_space.get_default(), cfg.observation_space.get_default())
        self._invalid = _Lagged([], [])
        self._act = _Lagged(cfg.action_space.get_default(), cfg.action_space.get_default())
        self._pending_reward = 0.0  # env reward accumulated since the last delivery
        self._reward = 0.0
        self._window.clear()
        self._tracking.clear()

    def _absorb(self):
        """Take the environment's current observation in, then tell the plugin: `on_reset` for the first
        observation of an episode, `on_step` (with the shifted views) for every later one."""
        cfg, env = self._config, self._env
        self._obs.push(self._window.absorb(cfg.state_encode_one_step(env.state, env)))
        self._invalid.push([cfg.action_encode(a) for a in env.get_invalid_actions(self._seat)])
        if not self._announced:
            self._announced = True
            self._worker.on_reset(self)
            return
        self._reward = (self._pending_reward + cfg.reward_shift) * cfg.reward_scale  # worker_run.py:346
        self._pending_reward = 0.0
        self._step_in_episode += 1
        self._step_in_training += 1
        self._delivering = True
        try:
            self._worker.on_step(self)
        finally:
            self._delivering = False

    def policy(self):
        self._absorb()
        self._act.push(None)
        self._act.new = self._worker.policy(self)
        return self._config.action_decode(self._act.new)

    def on_step(self) -> None:
        """Called by the loop after every env.step (also for seats that did not act)."""
        if not self._announced:
            return
        self._pending_reward += self._env.rewards[self._seat]
        if self._env._done != DoneTypes.NONE:
            self._absorb()  # there will be no further policy() in this episode: deliver the terminal transition now

    # ---- tracking ring (worker_run.py:548-610) --------------------------------------------------
    def set_tracking_max_size(self, max_size: int = -1):
        self._tracking.set_limit(max_size)

    def get_tracking_length(self) -> int:
        return len(self._tracking.rows)

    def add_tracking(self, data: Dict[str, Any]):
        self._tracking.add(data)

    def get_tracking_data(self) -> List[Dict[str, Any]]:
        return list(self._tracking.rows)

    def get_tracking(self, key: str, size: Optional[int] = None, dummy: Any = None) -> list:
        return self._tracking.column(key, size, dummy)

    def get_trackings(self, keys: Optional[List[str]] = None, size: int = 0, padding_data: dict = {}, padding_direct: str = "head") -> list:
        return self._tracking.table(keys, size, padding_data, padding_direct)

    # ---- backup / restore (worker_run.py:612-672; search-type algorithms roll a worker back) -------
    _SNAPSHOT = ("_is_setup", "_step_in_training", "_seat", "_episode_seed", "_announced", "_step_in_episode", "_pending_reward", "_reward")

    def backup(self) -> Any:
        obs, one, act = self._config.observation_space, self._config.observation_space_one_step, self._config.action_space
        return dict(
            scalars={k: getattr(self, k) for k in self._SNAPSHOT},
            obs=(obs.copy_value(self._obs.old), obs.copy_value(self._obs.new)),
            frames=[one.copy_value(f) for f in self._window.frames],
            act=(act.copy_value(self._act.old), act.copy_value(self._act.new)),
            invalid=(list(self._invalid.old), list(self._invalid.new)),
            tracking=(self._tracking.limit, list(self._tracking.keys), [dict(r) for r in self._tracking.rows]),
            env=self._env.backup(),
        )

    def restore(self, snap: Any):
        for k, v in snap["scalars"].items():
            setattr(self, k, v)
        self._obs = _Lagged(*snap["obs"])
        self._window.frames.clear()
        self._window.frames.extend(snap["frames"])
        self._act = _Lagged(*snap["act"])
        self._invalid = _Lagged(list(snap["invalid"][0]), list(snap["invalid"][1]))
        limit, keys, rows = snap["tracking"]
        self._tracking.set_limit(limit)
        self._tracking.clear()
        self._tracking.keys = list(keys)
        self._tracking.rows.extend(dict(r) for r in rows)
        self._env.restore(snap["env"])

    def print_discrete_action_info(self, maxa: int, func) -> None:
        for a in range(min(15, self._config.action_space.n)):
            flag = "x" if a in self.invalid_actions else ("*" if a == maxa else " ")
            print(f"{flag}{self._env.action_to_str(a):3s}: {func(a)}")
