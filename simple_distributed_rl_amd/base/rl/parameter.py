"""RLParameter (srl/base/rl/parameter.py:15-62)."""
from abc import ABC, abstractmethod
from typing import Any

from simple_distributed_rl_amd.utils.common import load_file, save_file


class RLParameter(ABC):
    def __init__(self, config=None):
        if config is None:
            from simple_distributed_rl_amd.base.rl.config import DummyRLConfig

            config = DummyRLConfig()
        self.config = config
        self.setup()

    def setup(self) -> None:
        pass

    @abstractmethod
    def call_restore(self, data: Any, from_serialized: bool = False, from_worker: bool = False, **kwargs) -> None:
        raise NotImplementedError()

    @abstractmethod
    def call_backup(self, serialized: bool = False, to_worker: bool = False, **kwargs) -> Any:
        raise NotImplementedError()

    def restore(self, data: Any, from_serialized: bool = False, from_worker: bool = False, **kwargs) -> None:
        self.call_restore(data, from_serialized=from_serialized, from_worker=from_worker, **kwargs)

    def backup(self, serialized: bool = False, to_worker: bool = False, **kwargs) -> Any:
        dat = self.call_backup(serialized=serialized, to_worker=to_worker, **kwargs)
        return None if (isinstance(dat, list) and dat == []) else dat

    def save(self, path: str, compress: bool = True, **kwargs) -> None:
        save_file(path, self.backup(**kwargs), compress)

    def load(self, path: str, **kwargs) -> None:
        self.restore(load_file(path), **kwargs)

    def summary(self, **kwargs):
        pass

    def update_from_worker_parameter(self, worker_parameter: "RLParameter") -> None:
        pass


class DummyRLParameter(RLParameter):
    def call_restore(self, data: Any, **kwargs) -> None:
        pass

    def call_backup(self, **kwargs) -> Any:
        return None
