"""RLTrainer (srl/base/rl/trainer.py:14-57): `train()` returning without incrementing `train_count`
means "not warmed up" (core_play.py:188-194)."""
from abc import ABC

from simple_distributed_rl_amd.base.context import RunContext


class RLTrainer(ABC):
    def __init__(self, config, parameter, memory):
        self.config = config
        self.parameter = parameter
        self.memory = memory
        self.__context = RunContext()
        self.train_count: int = 0
        self.info: dict = {}

    def get_train_count(self) -> int:
        return self.train_count

    @property
    def context(self) -> RunContext:
        return self.__context

    @property
    def distributed(self) -> bool:
        return self.__context.distributed

    @property
    def train_only(self) -> bool:
        return self.__context.train_only

    def setup(self, context: RunContext) -> None:
        self.__context = context
        self.on_setup()

    def teardown(self) -> None:
        self.on_teardown()

    def on_setup(self) -> None:
        pass

    def on_teardown(self) -> None:
        pass

    def train(self) -> None:
        raise NotImplementedError()


class DummyRLTrainer(RLTrainer):
    def train(self) -> None:
        self.train_count += 1
