"""Tabular Q-learning, registered as "QL" -- BASELINE.json configs[0]: the CPU plumbing check of the
Runner / Config / Worker / Trainer / Memory surface (there is no device work in it).

Behaviour follows srl/algorithms/ql.py:29-189: a table `Q[state string] -> [value per action]`, created lazily
per `q_init`, invalid actions pinned to -inf when they are first seen as a successor's mask, one TD(0) update
and one `train_count` per stored transition, epsilon-greedy with random tie-breaking, parameter backup as the
JSON text of the table (the reference's wire format, :90-95).  Written around a `QTable` value object so that
the update rule reads as arithmetic on rows rather than dictionary plumbing."""
import json
import math
import random
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List

import numpy as np

from simple_distributed_rl_amd.base.rl.algorithms.base_ql import RLConfig, RLWorker
from simple_distributed_rl_amd.base.rl.parameter import RLParameter
from simple_distributed_rl_amd.base.rl.registration import register
from simple_distributed_rl_amd.base.rl.trainer import RLTrainer
from simple_distributed_rl_amd.rl.functions import get_random_max_index
from simple_distributed_rl_amd.rl.memories.single_use_buffer import RLSingleUseBuffer
from simple_distributed_rl_amd.rl.schedulers.scheduler import SchedulerConfig


@dataclass
class Config(RLConfig):
    test_epsilon: float = 0
    epsilon: float = 0.1
    epsilon_scheduler: SchedulerConfig = field(default_factory=lambda: SchedulerConfig())
    lr: float = 0.1
    lr_scheduler: SchedulerConfig = field(default_factory=lambda: SchedulerConfig())
    discount: float = 0.9
    q_init: str = ""  # "" (zeros) | "random" (U[0,1) from `random`) | "normal" (N(0,1) from numpy)

    def get_name(self) -> str:
        return "QL"


register(Config(), __name__ + ":Memory", __name__ + ":Parameter", __name__ + ":Trainer", __name__ + ":Worker", check_duplicate=False)


class Memory(RLSingleUseBuffer):
    pass


class QTable(dict):
    """state string -> list of action values; rows appear on first touch."""

    def __init__(self, n_actions: int, init: str = ""):
        super().__init__()
        self.n_actions, self.init = n_actions, init

    def _fresh(self) -> List[float]:
        if self.init == "random":
            return [random.random() for _ in range(self.n_actions)]
        if self.init == "normal":
            return [float(np.random.normal()) for _ in range(self.n_actions)]
        return [0.0] * self.n_actions

    def row(self, key: str, forbid: Iterable[int] = ()) -> List[float]:
        r = self.get(key)
        if r is None:
            r = self[key] = self._fresh()
        for a in forbid:
            r[a] = -math.inf
        return r


class Parameter(RLParameter):
    def setup(self):
        self.Q = QTable(self.config.action_space.n, self.config.q_init)

    def call_backup(self, **kwargs) -> str:
        return json.dumps(self.Q)

    def call_restore(self, data: Any, **kwargs) -> None:
        table = QTable(self.config.action_space.n, self.config.q_init)
        table.update(json.loads(data))
        self.Q = table

    def get_action_values(self, state: str, update_invalid_actions: list = []) -> List[float]:
        return self.Q.row(state, update_invalid_actions)


class Trainer(RLTrainer):
    def on_setup(self) -> None:
        self.lr_sch = self.config.lr_scheduler.create(self.config.lr)

    def train(self) -> None:
        transitions = self.memory.sample()
        if transitions is None:
            return
        table, gamma = self.parameter.Q, self.config.discount
        lr = self.lr_sch.update(self.train_count).to_float()
        delta = 0.0
        for s, s_next, a, r, is_terminal, masked_next in transitions:
            bootstrap = 0.0 if is_terminal else gamma * max(table.row(s_next, masked_next))
            values = table.row(s)
            delta = (r + bootstrap) - values[a]
            values[a] += lr * delta
            self.train_count += 1
        self.info.update(size=len(table), td_error=delta, lr=lr)


class Worker(RLWorker):
    def on_setup(self, worker, context) -> None:
        self.epsilon_sch = self.config.epsilon_scheduler.create(self.config.epsilon)

    def _key(self, observation) -> str:
        return self.config.observation_space.to_str(observation)

    def policy(self, worker) -> int:
        self.state = self._key(worker.state)
        masked = worker.invalid_actions
        if self.training:
            eps = self.epsilon_sch.update(self.step_in_training).to_float()
        else:
            eps = self.config.test_epsilon
        self.info["epsilon"] = eps
        if random.random() < eps:
            return random.choice([a for a in range(self.config.action_space.n) if a not in masked])
        return get_random_max_index(self.parameter.Q.row(self.state), masked)

    def on_step(self, worker):
        if self.training:
            self.memory.add([self.state, self._key(worker.next_state), worker.action, worker.reward, worker.terminated, worker.next_invalid_actions])
