"""RLParameter: what an algorithm's learnable state looks like to the runner (reference contract:
srl/base/rl/parameter.py:15-62).

Algorithms implement `call_backup` / `call_restore`; the runner, the mp topology and checkpoints go through `backup` /
`restore` / `save` / `load`.  Flags the algorithms may honour: `serialized` (the snapshot must survive a process
boundary: CPU tensors), `to_worker` / `from_worker` (a learner -> actor hand-over may omit optimizer-only state).  An empty
list from `call_backup` means "nothing to send" and becomes None (the mp parameter board skips it)."""
import abc
from typing import Any

from simple_distributed_rl_amd.utils.common import load_file, save_file


class RLParameter(abc.ABC):
    def __init__(self, config=None):
        if config is None:
            from simple_distributed_rl_amd.base.rl.config import DummyRLConfig

            config = DummyRLConfig()
        self.config = config
        self.setup()

    # ---- implemented by the algorithm --------------------------------------------------------
    def setup(self) -> None:
        """Build the networks / tables (called once from the constructor)."""

    @abc.abstractmethod
    def call_backup(self, serialized: bool = False, to_worker: bool = False, **kwargs) -> Any:
        ...

    @abc.abstractmethod
    def call_restore(self, data: Any, from_serialized: bool = False, from_worker: bool = False, **kwargs) -> None:
        ...

    def summary(self, **kwargs) -> None:
        """Optional human-readable description of the model."""

    def update_from_worker_parameter(self, worker_parameter: "RLParameter") -> None:
        """Optional: merge state a worker accumulated locally (unused by the built-in algorithms)."""

    # ---- used by the runner --------------------------------------------------------------------
    def backup(self, serialized: bool = False, to_worker: bool = False, **kwargs) -> Any:
        snapshot = self.call_backup(serialized=serialized, to_worker=to_worker, **kwargs)
        if isinstance(snapshot, list) and len(snapshot) == 0:
            return None
        return snapshot

    def restore(self, data: Any, from_serialized: bool = False, from_worker: bool = False, **kwargs) -> None:
        self.call_restore(data, from_serialized=from_serialized, from_worker=from_worker, **kwargs)

    def save(self, path: str, compress: bool = True, **kwargs) -> None:
        save_file(path, self.backup(**kwargs), compress)

    def load(self, path: str, **kwargs) -> None:
        self.restore(load_file(path), **kwargs)


class DummyRLParameter(RLParameter):
    """For algorithms without learnable state and for workers created without a trainer."""

    def call_backup(self, **kwargs) -> Any:
        return None

    def call_restore(self, data: Any, **kwargs) -> None:
        return None
