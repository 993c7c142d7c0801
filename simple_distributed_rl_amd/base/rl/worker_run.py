"""WorkerRun: drives one RLWorker against one EnvRun (srl/base/rl/worker_run.py:24-610).

Kept semantics: state shifting prev_state/state/next_state with the on_step view (:104-130), frame stacking
over `window_length` one-step states that start as the space default (:277,316-322), invalid-action
tracking, reward shift/scale (:346), lazy `on_step` -- the plugin's on_step for step t runs inside the
NEXT policy() call, or immediately when the episode ended (:310-358,375-401) -- and the bounded tracking
ring (:548-610).  Rendering is out of scope."""
from typing import Any, Dict, List, Optional

from simple_distributed_rl_amd.base.context import RunContext, RunState
from simple_distributed_rl_amd.base.define import DoneTypes
from simple_distributed_rl_amd.base.exception import SRLError


class WorkerRun:
    def __init__(self, worker, env):
        worker.config.setup(env, enable_log=False)
        worker._set_worker_run(self)
        self._worker = worker
        self._config = worker.config
        self._env = env
        self._is_setup = False
        self._setup_val(RunContext(), RunState())
        self._reset_val(0)

    # ---- properties ---------------------------------------------------------------------------
    @property
    def worker(self):
        return self._worker

    @property
    def config(self):
        return self._config

    @property
    def env(self):
        return self._env

    @property
    def context(self) -> RunContext:
        return self._context

    @property
    def distributed(self) -> bool:
        return self._context.distributed

    @property
    def training(self) -> bool:
        return self._context.training

    @property
    def train_only(self) -> bool:
        return self._context.train_only

    @property
    def rollout(self) -> bool:
        return self._context.rollout

    @property
    def actor_id(self) -> int:
        return self._context.actor_id

    @property
    def player_index(self) -> int:
        return self._player_index

    @property
    def info(self) -> dict:
        return self._worker.info

    @property
    def run_state(self) -> RunState:
        return self._run_state

    @property
    def train_count(self) -> int:
        return self._run_state.train_count

    # inside on_step the views shift by one step (:104-130)
    @property
    def prev_state(self):
        return None if self._on_step_in_progress else self._prev_state

    @property
    def state(self):
        return self._prev_state if self._on_step_in_progress else self._state

    @property
    def next_state(self):
        return self._state if self._on_step_in_progress else None

    def get_state_one_step(self, idx: int = -1):
        return self._one_states[idx] if self._use_stacked_state else self._state

    @property
    def prev_action(self):
        return self._prev_action

    @property
    def action(self):
        return self._action

    def get_onehot_prev_action(self):
        return self._config.action_space.get_onehot(self._prev_action)

    def get_onehot_action(self, action=None):
        return self._config.action_space.get_onehot(self._action if action is None else action)

    @property
    def reward(self) -> float:
        return self._reward

    @property
    def done(self) -> bool:
        return self._env._done != DoneTypes.NONE

    @property
    def terminated(self) -> bool:
        return self._env._done == DoneTypes.TERMINATED

    @property
    def done_type(self) -> DoneTypes:
        return self._env._done

    @property
    def done_reason(self) -> str:
        return self._env.env.done_reason

    @property
    def prev_invalid_actions(self) -> list:
        return None if self._on_step_in_progress else self._prev_invalid_actions

    @property
    def invalid_actions(self) -> list:
        return self._prev_invalid_actions if self._on_step_in_progress else self._invalid_actions

    @property
    def next_invalid_actions(self) -> list:
        return self._invalid_actions if self._on_step_in_progress else None

    @property
    def step_in_training(self) -> int:
        return self._step_in_training

    @property
    def step_in_episode(self) -> int:
        return self._step_in_episode

    @property
    def episode_seed(self) -> Optional[int]:
        return self._episode_seed

    # ---- lifecycle ----------------------------------------------------------------------------
    def setup(self, context: Optional[RunContext] = None, render_mode: str = "", run_state: Optional[RunState] = None):
        if context is None:
            context = RunContext(self._env.config, self._config)
        if run_state is None:
            run_state = RunState()
        self._setup_val(context, run_state)
        self._worker.on_setup(self, context)
        self._is_setup = True

    def _setup_val(self, context: RunContext, run_state: RunState):
        self._context = context
        self._run_state = run_state
        self._step_in_training = 0
        self._use_stacked_state = self._config.window_length > 1
        self._tracking_size = -1

    def teardown(self):
        self._worker.on_teardown(self)
        self._is_setup = False

    def reset(self, player_index: int, seed: Optional[int] = None) -> None:
        if not self._is_setup:
            raise SRLError("Cannot call worker.on_reset() before calling worker.setup()")
        self._reset_val(player_index, seed)

    def _reset_val(self, player_index: int, seed: Optional[int] = None):
        self._player_index = player_index
        self._episode_seed = seed
        self._is_reset = False
        self._step_in_episode = 0
        self._on_step_in_progress = False
        obs, one = self._config.observation_space, self._config.observation_space_one_step
        self._prev_state = obs.get_default()
        self._state = obs.get_default()
        self._one_states = [one.get_default() for _ in range(self._config.window_length)]
        self._prev_action = self._config.action_space.get_default()
        self._action = self._config.action_space.get_default()
        self._step_reward = 0.0
        self._reward = 0.0
        self._prev_invalid_actions: list = []
        self._invalid_actions: list = []
        self._tracking_data: List[Dict[str, Any]] = []
        self._tracking_keys: List[str] = []

    def _ready_policy(self):
        """First call of an episode -> on_reset, afterwards -> on_step (worker_run.py:310-358)."""
        self._prev_state = self._state
        state = self._config.state_encode_one_step(self._env.state, self._env)
        if self._use_stacked_state:
            del self._one_states[0]
            self._one_states.append(state)
            state = self._config.observation_space_one_step.encode_stack(self._one_states)
        self._state = state

        self._prev_invalid_actions = self._invalid_actions
        self._invalid_actions = [self._config.action_encode(a) for a in self._env.get_invalid_actions(self._player_index)]

        if not self._is_reset:
            self._is_reset = True
            self._worker.on_reset(self)
        else:
            self._reward = (self._step_reward + self._config.reward_shift) * self._config.reward_scale
            self._step_reward = 0.0
            self._step_in_episode += 1
            self._step_in_training += 1
            self._on_step_in_progress = True
            self._worker.on_step(self)
            self._on_step_in_progress = False

    def policy(self):
        self._ready_policy()
        self._prev_action = self._action
        self._action = None
        self._action = self._worker.policy(self)
        return self._config.action_decode(self._action)

    def on_step(self) -> None:
        if not self._is_reset:
            return
        self._step_reward += self._env.rewards[self._player_index]
        if self._env._done != DoneTypes.NONE:
            self._ready_policy()  # deliver the terminal state to the plugin now

    # ---- invalid actions ------------------------------------------------------------------------
    def get_valid_actions(self) -> list:
        return self._config.action_space.get_valid_actions(self.invalid_actions)

    def add_invalid_actions(self, invalid_actions: list, encode: bool = False) -> None:
        if encode:
            invalid_actions = [self._config.action_encode(a) for a in invalid_actions]
        self._invalid_actions = list(set(self._invalid_actions + invalid_actions))

    # ---- tracking ring (:548-610) -------------------------------------------------------------------
    def set_tracking_max_size(self, max_size: int = -1):
        self._tracking_size = max_size

    def get_tracking_length(self) -> int:
        return len(self._tracking_data)

    def add_tracking(self, data: Dict[str, Any]):
        if self._tracking_size > 0 and len(self._tracking_data) == self._tracking_size:
            del self._tracking_data[0]
        for k in data:
            if k not in self._tracking_keys:
                self._tracking_keys.append(k)
        self._tracking_data.append(data)

    def get_tracking_data(self) -> List[Dict[str, Any]]:
        return self._tracking_data

    def get_tracking(self, key: str, size: Optional[int] = None, dummy: Any = None) -> list:
        vals = [d.get(key, dummy) for d in self._tracking_data]
        if size is None:
            return vals
        if size <= 0:
            return []
        if len(vals) < size:
            return [dummy] * (size - len(vals)) + vals
        return vals[-size:]

    def get_trackings(self, keys: Optional[List[str]] = None, size: int = 0, padding_data: dict = {}, padding_direct: str = "head") -> list:
        if keys is None:
            keys = self._tracking_keys
        rows = [[d.get(k) for k in keys] for d in self._tracking_data]
        if size <= 0:
            return rows
        if len(rows) >= size:
            return rows[-size:]
        pad = [[padding_data.get(k) for k in keys] for _ in range(size - len(rows))]
        if padding_direct == "head":
            return pad + rows
        if padding_direct == "tail":
            return rows + pad
        raise ValueError(padding_direct)

    # ---- backup / restore (:612-672) ----------------------------------------------------------------
    def backup(self) -> Any:
        obs, one, act = self._config.observation_space, self._config.observation_space_one_step, self._config.action_space
        return [
            self._is_setup, self._step_in_training, self._tracking_size, self._player_index, self._episode_seed, self._is_reset,
            self._step_in_episode, obs.copy_value(self._prev_state), obs.copy_value(self._state), [one.copy_value(s) for s in self._one_states],
            act.copy_value(self._prev_action), act.copy_value(self._action), self._step_reward, self._reward,
            self._prev_invalid_actions[:], self._invalid_actions[:], self._env.backup(), [d.copy() for d in self._tracking_data],
        ]

    def restore(self, d: Any):
        (self._is_setup, self._step_in_training, self._tracking_size, self._player_index, self._episode_seed, self._is_reset,
         self._step_in_episode, self._prev_state, self._state, self._one_states, self._prev_action, self._action, self._step_reward,
         self._reward, self._prev_invalid_actions, self._invalid_actions, env_dat, tracking) = d
        self._env.restore(env_dat)
        self._tracking_data = [x.copy() for x in tracking]

    # ---- utils ------------------------------------------------------------------------------
    def sample_action(self):
        return self._config.action_space.sample(self._invalid_actions)

    def override_action(self, env_action, encode: bool = True):
        self._action = self._config.action_encode(env_action) if encode else env_action
        return self._action

    def abort_episode(self):
        self._env.abort_episode()

    def print_discrete_action_info(self, maxa: int, func) -> None:
        for action in range(min(15, self._config.action_space.n)):
            mark = "x" if action in self.invalid_actions else ("*" if action == maxa else " ")
            print(f"{mark}{self._env.action_to_str(action):3s}: {func(action)}")
