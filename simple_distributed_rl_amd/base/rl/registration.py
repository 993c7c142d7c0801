"""Algorithm registry (srl/base/rl/registration.py:30-130,228-251): config name -> entry points of the
four plugin classes, keyed `name` or `name:framework`; `make_*` factories."""
import logging
from typing import Dict, Optional, Tuple

from simple_distributed_rl_amd.utils.common import load_module

logger = logging.getLogger(__name__)
_registry: Dict[str, Tuple[str, str, str, str]] = {}
_registry_worker: Dict[str, str] = {}


def _key(config) -> str:
    name, fw = config.get_name(), config.get_framework()
    return name if fw == "" else f"{name}:{fw}"


def register(config, memory_entry_point: str, parameter_entry_point: str, trainer_entry_point: str, worker_entry_point: str,
             check_duplicate: bool = True) -> None:
    key = _key(config)
    if check_duplicate:
        assert key not in _registry, f"{key} was already registered."
    _registry[key] = (memory_entry_point, parameter_entry_point, trainer_entry_point, worker_entry_point)


def register_rulebase(name: str, entry_point: str, check_duplicate: bool = True) -> None:
    if check_duplicate:
        assert name not in _registry_worker, f"{name} was already registered."
    _registry_worker[name] = entry_point


def _entry(config) -> Tuple[str, str, str, str]:
    key = _key(config)
    if key not in _registry:
        import simple_distributed_rl_amd.algorithms  # noqa: F401  (registers the built-in algorithms)
    if key not in _registry:
        raise KeyError(f"'{key}' is not registered (registered: {sorted(_registry)})")
    return _registry[key]


def _setup(config, env):
    if env is None:
        assert config.is_setup(), "rl_config.setup(env) must be called first (or pass env)"
    else:
        config.setup(env)


def make_memory(config, env=None):
    _setup(config, env)
    return load_module(_entry(config)[0])(config)


def make_parameter(config, env=None):
    _setup(config, env)
    return load_module(_entry(config)[1])(config)


def make_trainer(config, parameter, memory, env=None):
    _setup(config, env)
    return load_module(_entry(config)[2])(config, parameter, memory)


def make_worker(config, env, parameter=None, memory=None):
    from simple_distributed_rl_amd.base.rl.worker_run import WorkerRun

    config.setup(env)
    worker = load_module(_entry(config)[3])(config, parameter, memory)
    return WorkerRun(worker, env)


def make_workers(config, players, env, parameter=None, memory=None, main_worker=None):
    """Single main worker for every seat not claimed by `players` (multi-player seats beyond the first get a
    random worker: rule-base opponents are out of scope)."""
    from simple_distributed_rl_amd.base.rl.worker import DummyRLWorker
    from simple_distributed_rl_amd.base.rl.worker_run import WorkerRun

    workers = []
    main_idx = 0
    for i in range(env.player_num):
        if i == 0:
            workers.append(main_worker if main_worker is not None else make_worker(config, env, parameter, memory))
        else:
            workers.append(WorkerRun(DummyRLWorker(config.copy(reset_env_config=True)), env))
    return workers, main_idx
