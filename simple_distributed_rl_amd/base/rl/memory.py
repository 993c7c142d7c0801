"""RLMemory: the memory plugin base with the worker/trainer function registry the multiprocess loops
consume (srl/base/rl/memory.py:48-150)."""
import logging
import lzma
import pickle
from abc import ABC
from typing import Any, Callable, Dict, List, Optional, Tuple

from simple_distributed_rl_amd.utils.common import load_file, save_file

logger = logging.getLogger(__name__)


class RLMemory(ABC):
    def __init__(self, config=None):
        if config is None:
            from simple_distributed_rl_amd.base.rl.config import DummyRLConfig

            config = DummyRLConfig()
        self.config = config
        self.__worker_funcs: Dict[str, Tuple[Callable, Optional[Callable]]] = {}
        self.__trainer_recv_funcs: List[Callable] = []
        self.__trainer_send_funcs: Dict[str, Callable] = {}
        self.setup()

    def setup(self) -> None:
        pass

    # Worker -> Memory (pickle serialisation) (:59-69)
    def register_worker_func(self, func: Callable):
        if func.__name__ in self.__worker_funcs:
            logger.warning(f"'{func.__name__}' is already registered. It has been overwritten.")
        self.__worker_funcs[func.__name__] = (func, None)

    # Worker -> Memory with a hand-written serialiser; `func` takes a trailing `serialized` flag (:71-85)
    def register_worker_func_custom(self, func: Callable, serialize_func: Callable):
        if func.__name__ in self.__worker_funcs:
            logger.warning(f"'{func.__name__}' is already registered. It has been overwritten.")
        self.__worker_funcs[func.__name__] = (func, serialize_func)

    def get_worker_funcs(self):
        return self.__worker_funcs

    # Memory -> Trainer (:90-98)
    def register_trainer_recv_func(self, func: Callable):
        self.__trainer_recv_funcs.append(func)

    def get_trainer_recv_funcs(self):
        return self.__trainer_recv_funcs

    # Trainer -> Memory (:100-108)
    def register_trainer_send_func(self, func: Callable):
        if func.__name__ in self.__trainer_send_funcs:
            logger.warning(f"'{func.__name__}' is already registered. It has been overwritten.")
        self.__trainer_send_funcs[func.__name__] = func

    def get_trainer_send_funcs(self):
        return self.__trainer_send_funcs

    def length(self) -> int:
        return -1

    def call_backup(self, **kwargs) -> Any:
        raise NotImplementedError()

    def call_restore(self, data: Any, **kwargs) -> None:
        raise NotImplementedError()

    def backup(self, compress: bool = False, **kwargs) -> Any:
        dat = self.call_backup(**kwargs)
        if compress:
            dat = (lzma.compress(pickle.dumps(dat)), True)
        return dat

    def restore(self, dat: Any, **kwargs) -> None:
        if isinstance(dat, tuple):
            dat = pickle.loads(lzma.decompress(dat[0]))
        self.call_restore(dat, **kwargs)

    def save(self, path: str, compress: bool = True, **kwargs) -> None:
        save_file(path, self.call_backup(**kwargs), compress)

    def load(self, path: str, **kwargs) -> None:
        self.call_restore(load_file(path), **kwargs)


class DummyRLMemory(RLMemory):
    def call_backup(self, **kwargs) -> Any:
        return None

    def call_restore(self, data: Any, **kwargs) -> None:
        pass
