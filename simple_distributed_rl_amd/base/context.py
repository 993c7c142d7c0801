"""RunContext / RunState: the run-time flags and counters of a play loop (srl/base/context.py:27-108,297-349).
Field names are the reference's, so callbacks written against it keep working."""
from dataclasses import dataclass, field
from typing import Any, List, Optional, Union


@dataclass
class RunContext:
    env_config: Any = None
    rl_config: Any = None
    callbacks: list = field(default_factory=list)
    run_name: str = "main"
    play_mode: str = ""
    # stop config
    max_episodes: int = 0
    timeout: float = 0
    max_steps: int = 0
    max_train_count: int = 0
    max_memory: int = 0
    # play config
    players: list = field(default_factory=list)
    shuffle_player: bool = True
    disable_trainer: bool = False
    # train option
    train_interval: int = 1
    train_repeat: int = 1
    # play info
    distributed: bool = False
    training: bool = False
    train_only: bool = False
    rollout: bool = False
    env_render_mode: str = ""
    rl_render_mode: str = ""
    # mp
    actor_num: int = 1
    actor_devices: Union[str, List[str]] = "CPU"
    memory_limit: Optional[int] = -1
    enable_stats: bool = True
    seed: Optional[int] = None
    seed_enable_gpu: bool = False
    device: str = "AUTO"

    def __post_init__(self):
        self.actor_id: int = 0
        self.framework: str = ""
        self.used_device_torch: str = "cpu"

    def check_context_parameter(self, check_stop_config: bool = True):
        assert self.rl_config is not None
        assert self.callbacks is not None
        if not check_stop_config:
            return
        if self.train_only or (self.distributed and self.run_name == "trainer"):
            assert self.max_train_count > 0 or self.timeout > 0, "Specify one of the following: 'max_train_count', 'timeout'"
        elif self.training:
            assert (
                self.max_steps > 0 or self.max_episodes > 0 or self.timeout > 0 or self.max_train_count > 0 or self.max_memory > 0
            ), "Specify one of the following: 'max_episodes', 'timeout', 'max_steps', 'max_train_count', 'max_memory'"

    def setup_device(self):
        """srl/base/system/device.py:159-201 reduced to torch: "AUTO"/"GPU" -> cuda:0 when a HIP device is
        visible (ROCm exposes it as cuda), "CPU" -> cpu, explicit "cuda:N"/"gpu:N" honoured."""
        dev = (self.device or "AUTO").upper()
        used = "cpu"
        try:
            import torch

            has = torch.cuda.is_available()
        except ImportError:
            has = False
        if dev.startswith(("CUDA", "GPU")) or dev == "AUTO":
            if has:
                idx = dev.split(":")[1] if ":" in dev else "0"
                used = f"cuda:{idx}"
        self.used_device_torch = used
        if self.rl_config is not None and hasattr(self.rl_config, "_set_device"):
            self.rl_config._set_device(used)
        return used

    def copy(self) -> "RunContext":
        import copy

        c = copy.copy(self)
        c.callbacks = self.callbacks[:]
        c.players = self.players[:]
        c.actor_id = self.actor_id
        c.framework = self.framework
        c.used_device_torch = self.used_device_torch
        return c


@dataclass
class RunState:
    elapsed_t0: float = 0
    worker_indices: List[int] = field(default_factory=list)
    episode_rewards_list: List[List[float]] = field(default_factory=list)
    episode_count: int = -1
    total_step: int = 0
    end_reason: str = ""
    worker_idx: int = 0
    episode_seed: Optional[int] = None
    action: Any = None
    train_count: int = 0
    is_step_trained: bool = False
    sync_actor: int = 0
    actor_send_q: int = 0
    sync_trainer: int = 0
    trainer_recv_q: int = 0
    last_episode_step: float = 0
    last_episode_time: float = 0
    last_episode_rewards: List[float] = field(default_factory=list)
    shared_vars: dict = field(default_factory=dict)


@dataclass
class RunStateActor(RunState):
    env: Any = None
    worker: Any = None
    workers: Any = None
    parameter: Any = None
    memory: Any = None
    trainer: Any = None


@dataclass
class RunStateTrainer(RunState):
    trainer: Any = None
    memory: Any = None
    parameter: Any = None
