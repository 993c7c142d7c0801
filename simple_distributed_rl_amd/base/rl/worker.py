"""RLWorker: the policy plugin base (srl/base/rl/worker.py:25-146): on_setup / on_reset / policy / on_step
plus shortcut properties into the WorkerRun that drives it."""
from abc import ABC, abstractmethod

from simple_distributed_rl_amd.base.define import DoneTypes
from simple_distributed_rl_amd.base.rl.memory import DummyRLMemory
from simple_distributed_rl_amd.base.rl.parameter import DummyRLParameter


class RLWorker(ABC):
    def __init__(self, config, parameter=None, memory=None) -> None:
        self.config = config
        self.parameter = DummyRLParameter(config) if parameter is None else parameter
        self.memory = DummyRLMemory(config) if memory is None else memory
        self.info: dict = {}

    def _set_worker_run(self, worker):
        self.__worker_run = worker

    # ---- implement ----------------------------------------------------------------------------
    def on_setup(self, worker, context) -> None:
        pass

    def on_teardown(self, worker) -> None:
        pass

    def on_reset(self, worker) -> None:
        pass

    @abstractmethod
    def policy(self, worker):
        raise NotImplementedError()

    def on_step(self, worker) -> None:
        pass

    def render_terminal(self, worker, **kwargs) -> None:
        pass

    # ---- shortcuts ----------------------------------------------------------------------------
    @property
    def worker(self):
        return self.__worker_run

    @property
    def env(self):
        return self.__worker_run._env

    def terminated(self) -> None:
        self.__worker_run._env._done = DoneTypes.TRUNCATED
        self.__worker_run._env.env.done_reason = "rl"

    @property
    def context(self):
        return self.__worker_run._context

    @property
    def distributed(self) -> bool:
        return self.__worker_run._context.distributed

    @property
    def training(self) -> bool:
        return self.__worker_run._context.training

    @property
    def train_only(self) -> bool:
        return self.__worker_run._context.train_only

    @property
    def rollout(self) -> bool:
        return self.__worker_run._context.rollout

    @property
    def rendering(self) -> bool:
        return self.__worker_run._context.rl_render_mode != ""

    @property
    def player_index(self) -> int:
        return self.__worker_run.player_index

    @property
    def step_in_training(self) -> int:
        return self.__worker_run.step_in_training

    @property
    def step_in_episode(self) -> int:
        return self.__worker_run.step_in_episode

    @property
    def run_state(self):
        return self.__worker_run._run_state

    @property
    def train_count(self) -> int:
        return self.__worker_run._run_state.train_count

    @property
    def max_episode_steps(self) -> int:
        return self.__worker_run._env.max_episode_steps

    @property
    def player_num(self) -> int:
        return self.__worker_run._env.player_num

    @property
    def step(self) -> int:
        return self.__worker_run._env.step_num

    def sample_action(self):
        return self.__worker_run.sample_action()


class DummyRLWorker(RLWorker):
    def policy(self, worker):
        return worker.sample_action()
