"""SpaceBase: the minimal contract WorkerRun/RLConfig need (srl/base/spaces/space.py)."""
from abc import ABC, abstractmethod
from typing import Any, List


class SpaceBase(ABC):
    @property
    @abstractmethod
    def stype(self):
        raise NotImplementedError()

    @abstractmethod
    def sample(self, mask: List[Any] = []) -> Any:
        raise NotImplementedError()

    @abstractmethod
    def get_default(self) -> Any:
        raise NotImplementedError()

    @abstractmethod
    def copy(self) -> "SpaceBase":
        raise NotImplementedError()

    def copy_value(self, v: Any) -> Any:
        import copy

        return copy.deepcopy(v)

    def to_str(self, val: Any) -> str:
        return str(val)

    def is_image_like(self) -> bool:
        return False

    def is_value(self) -> bool:
        return not self.is_image_like()

    def create_stack_space(self, length: int) -> "SpaceBase":
        raise NotImplementedError()

    def encode_stack(self, val: List[Any]) -> Any:
        raise NotImplementedError()
