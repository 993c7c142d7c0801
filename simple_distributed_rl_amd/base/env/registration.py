"""Environment registry (srl/base/env/registration.py:39-136): id -> "module:Class" entry point."""
import logging
from dataclasses import dataclass, field
from typing import Dict, Union

from simple_distributed_rl_amd.utils.common import load_module

logger = logging.getLogger(__name__)
_registry: Dict[str, dict] = {}


@dataclass
class EnvConfig:
    """srl/base/env/config.py: name + kwargs + a few step options."""

    name: str = ""
    kwargs: dict = field(default_factory=dict)
    max_episode_steps: int = -1
    episode_timeout: float = -1
    frameskip: int = 0
    random_noop_max: int = 0
    enable_assertion: bool = False
    enable_sanitize: bool = True

    def make(self):
        return make(self)

    def copy(self) -> "EnvConfig":
        import copy

        return copy.deepcopy(self)


def register(id: str, entry_point: str, kwargs: Dict = {}, check_duplicate: bool = True) -> None:
    if check_duplicate:
        assert id not in _registry, f"{id} was already registered. entry_point={entry_point}"
    elif id in _registry:
        logger.debug(f"{id} was already registered, but I overwrote it. entry_point={entry_point}")
    _registry[id] = {"entry_point": entry_point, "kwargs": kwargs}


def make_base(config: Union[str, EnvConfig]):
    if isinstance(config, str):
        config = EnvConfig(config)
    if config.name not in _registry:
        import simple_distributed_rl_amd.envs  # noqa: F401  (registers the built-in envs)
    if config.name not in _registry:
        raise KeyError(f"'{config.name}' is not registered (registered: {sorted(_registry)})")
    item = _registry[config.name]
    kw = dict(item["kwargs"])
    kw.update(config.kwargs)
    env = load_module(item["entry_point"])(**kw)
    env.init_base()
    return env


def make(config: Union[str, EnvConfig]):
    from .env_run import EnvRun

    if isinstance(config, str):
        config = EnvConfig(config)
    return EnvRun(config)
