"""EnvRun: episode bookkeeping around an EnvBase (srl/base/env/env_run.py:187-370): done types, step
counter with max_episode_steps truncation, frameskip, per-player rewards and invalid actions."""
import time
from typing import Any, List, Optional

from simple_distributed_rl_amd.base.define import DoneTypes
from simple_distributed_rl_amd.base.exception import SRLError

from .registration import EnvConfig, make_base


class EnvRun:
    def __init__(self, config: EnvConfig) -> None:
        self.config = config
        self.env = make_base(config)
        self.env.env_run = self
        self._is_setup = False
        self._done = DoneTypes.RESET
        self._reset_vals()

    # ---- properties ---------------------------------------------------------------------------
    @property
    def name(self) -> str:
        return self.config.name

    @property
    def unwrapped(self):
        return self.env

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def player_num(self) -> int:
        return self.env.player_num

    @property
    def max_episode_steps(self) -> int:
        return self.config.max_episode_steps if self.config.max_episode_steps > 0 else self.env.max_episode_steps

    @property
    def state(self):
        return self._state

    @property
    def next_player(self) -> int:
        return self.env.next_player

    @property
    def step_num(self) -> int:
        return self._step_num

    @property
    def done(self) -> bool:
        return self._done != DoneTypes.NONE

    @property
    def done_type(self) -> DoneTypes:
        return self._done

    @property
    def rewards(self) -> List[float]:
        return self._step_rewards

    @property
    def reward(self) -> float:
        return self._step_rewards[0]

    @property
    def episode_rewards(self) -> List[float]:
        return self._episode_rewards

    @property
    def elapsed_time(self) -> float:
        return time.time() - self._t0

    @property
    def reward_baseline(self):
        return self.env.reward_baseline

    def get_invalid_actions(self, player_index: int = -1) -> list:
        if player_index == -1:
            player_index = self.env.next_player
        return self._invalid_actions_list[player_index]

    def action_to_str(self, action) -> str:
        return self.env.action_to_str(action)

    # ---- lifecycle ----------------------------------------------------------------------------
    def _reset_vals(self):
        n = self.env.player_num
        self._step_num = 0
        self._state = None
        self._step_rewards = [0.0] * n
        self._episode_rewards = [0.0] * n
        self._invalid_actions_list = [[] for _ in range(n)]
        self._t0 = time.time()

    def setup(self, context=None, render_mode: str = "") -> None:
        self._done = DoneTypes.RESET  # reset() must come before step()
        self.env.done_reason = ""
        kwargs = {} if context is None else dict(training=context.training, distributed=context.distributed, seed=context.seed)
        self.env.setup(**kwargs)
        self._is_setup = True

    def teardown(self, **kwargs) -> None:
        self.env.teardown(**kwargs)
        self._is_setup = False

    def reset(self, *, seed: Optional[int] = None, **kwargs) -> None:
        if not self._is_setup:
            raise SRLError("Cannot call env.reset() before calling env.setup()")
        self._reset_vals()
        self._state = self.env.reset(seed=seed, **kwargs)
        self._done = DoneTypes.NONE
        self.env.done_reason = ""
        self._invalid_actions_list = [self.env.get_invalid_actions(i) for i in range(self.env.player_num)]

    def step(self, action, frameskip: int = 0, frameskip_function=None) -> None:
        if self._done != DoneTypes.NONE:
            raise SRLError(f"It is in the done state. Please execute reset(). ({self._done})")
        state, rewards, done = self._step1(action)
        total = rewards
        for _ in range(self.config.frameskip + frameskip):
            if done != DoneTypes.NONE:
                break
            state, rewards, done = self._step1(action)
            total = [total[i] + rewards[i] for i in range(self.env.player_num)]
            if frameskip_function is not None:
                frameskip_function()
        self._state = state
        self._step_rewards = total
        self._done = done
        self._invalid_actions_list[self.env.next_player] = self.env.get_invalid_actions(self.env.next_player)
        self._step_num += 1
        self._episode_rewards = [self._episode_rewards[i] + total[i] for i in range(self.env.player_num)]
        if self._done == DoneTypes.NONE:
            if self._step_num > self.max_episode_steps:  # env_run.py:354-356
                self._done = DoneTypes.TRUNCATED
                self.env.done_reason = "episode step over"
            elif self.config.episode_timeout > 0 and time.time() - self._t0 > self.config.episode_timeout:
                self._done = DoneTypes.TRUNCATED
                self.env.done_reason = "timeout"

    def _step1(self, action):
        state, rewards, terminated, truncated = self.env.step(action)
        rewards = list(rewards) if isinstance(rewards, (list, tuple)) else [float(rewards)]
        if truncated:
            done = DoneTypes.TRUNCATED
        elif terminated:
            done = DoneTypes.TERMINATED
        else:
            done = DoneTypes.NONE
        return state, rewards, done

    def abort_episode(self):
        self._done = DoneTypes.TRUNCATED
        self.env.done_reason = "abort"

    def backup(self) -> Any:
        return [self._step_num, self._state, self._step_rewards[:], self._episode_rewards[:], self._done, [a[:] for a in self._invalid_actions_list], self.env.backup()]

    def restore(self, dat: Any) -> None:
        self._step_num, self._state, self._step_rewards, self._episode_rewards, self._done = dat[0], dat[1], dat[2][:], dat[3][:], dat[4]
        self._invalid_actions_list = [a[:] for a in dat[5]]
        self.env.restore(dat[6])

    def close(self):
        self.env.close()
