"""Runner facade (srl/runner/runner.py:25-402,724-790 and runner_base.py:170-256): lazily builds env /
parameter / memory / trainer / worker from (env_config, rl_config) and drives the play loops.

    runner = Runner("Grid", ql.Config())
    runner.train(timeout=3)
    rewards = runner.evaluate(max_episodes=10)

Kept: train / rollout / train_only / train_mp / evaluate with the reference's keyword names, set_seed,
set_device, save/load of parameter and memory, the RunContext handed to callbacks.  Render / play_window /
history viewers / mlflow are out of scope.

Where the work runs: `train()` and `train_mp()` first ask `device/vector_runner.py` whether the (environment,
algorithm, device) triple is served by the hand-written engine -- the Rainbow family on a GPU with image frames.
If so the same loop (`base/run/sequence.py`) is driven by the device drivers with `vector_envs` environments per
iteration (`set_vector_envs`; "AUTO" = 1024 device-resident lanes, 64 for host-stepped environments) and the trained
weights are written back into `runner.parameter` at the end, so `evaluate()` / `save_parameter()` see them.
Everything else runs the registered plugin classes on one host environment, as the reference does;
`runner.vector_reason` says why."""
from typing import List, Optional, Union

from simple_distributed_rl_amd.base.context import RunContext
from simple_distributed_rl_amd.base.env.registration import EnvConfig
from simple_distributed_rl_amd.base.env.registration import make as make_env_run
from simple_distributed_rl_amd.base.run.sequence import play, play_trainer_only, run_sequence


class Runner:
    def __init__(self, name_or_env_config: Union[str, EnvConfig], rl_config=None, context: Optional[RunContext] = None):
        self.env_config = EnvConfig(name_or_env_config) if isinstance(name_or_env_config, str) else name_or_env_config
        if rl_config is None:
            from simple_distributed_rl_amd.base.rl.config import DummyRLConfig

            rl_config = DummyRLConfig()
        self.rl_config = rl_config
        self.context = context if context is not None else RunContext()
        self.context.env_config = self.env_config
        self.context.rl_config = self.rl_config
        self._env = None
        self._parameter = None
        self._memory = None
        self._trainer = None
        self._worker = None
        self.state = None
        self._vector_actor = None
        self._vector_envs: Union[str, int] = "AUTO"
        self.vector_reason = ""  # why the last train() ran on the plugin path ("" = it ran on the device engine)

    # ---- lazy factories (runner_base.py:170-256) ----------------------------------------------------
    def make_env(self):
        if self._env is None:
            self._env = make_env_run(self.env_config)
        return self._env

    @property
    def env(self):
        return self.make_env()

    def setup_rl_config(self):
        if not self.rl_config.is_setup():
            self.rl_config.setup(self.make_env())

    def make_parameter(self, is_load: bool = True):
        if self._parameter is None:
            self.setup_rl_config()
            self.context.setup_device()
            self._parameter = self.rl_config.make_parameter()
        return self._parameter

    @property
    def parameter(self):
        return self.make_parameter()

    def make_memory(self, is_load: bool = True):
        if self._memory is None:
            self.setup_rl_config()
            self._memory = self.rl_config.make_memory()
        return self._memory

    @property
    def memory(self):
        return self.make_memory()

    def make_trainer(self, parameter=None, memory=None):
        if self._trainer is None or parameter is not None or memory is not None:
            self._trainer = self.rl_config.make_trainer(parameter or self.make_parameter(), memory or self.make_memory())
        return self._trainer

    @property
    def trainer(self):
        return self.make_trainer()

    def make_worker(self, parameter=None, memory=None):
        if self._worker is None or parameter is not None or memory is not None:
            self._worker = self.rl_config.make_worker(self.make_env(), parameter or self.make_parameter(), memory or self.make_memory())
        return self._worker

    # ---- setters ------------------------------------------------------------------------------
    def set_seed(self, seed: Optional[int] = None, seed_enable_gpu: bool = False):
        self.context.seed = seed
        self.context.seed_enable_gpu = seed_enable_gpu

    def set_device(self, device: str = "AUTO", **kwargs):
        self.context.device = device

    def set_vector_envs(self, n_envs: Union[str, int] = "AUTO"):
        """Environments per iteration on the device engine: "AUTO", a count, or 0 to stay on the plugin path."""
        self._vector_envs = n_envs

    def _device_drivers(self, c: RunContext, with_learner: bool = True):
        """(actor, learner) drivers of the device engine for this run, or None (reason in `vector_reason`)."""
        if self._vector_envs == 0:
            self.vector_reason = "set_vector_envs(0)"
            return None
        env = self.make_env()
        self.setup_rl_config()
        c.setup_device()
        if not str(c.used_device_torch).startswith("cuda"):
            self.vector_reason = "the run is not on a GPU device"
            return None
        from simple_distributed_rl_amd.device import vector_runner as vr

        self.vector_reason = vr.why_not_vector(c, env, self.rl_config)
        if self.vector_reason:
            return None
        lanes = self._vector_envs
        if isinstance(lanes, str):
            lanes = 1024 if hasattr(type(env.unwrapped), "device_vector") else 64
        actor = self._vector_actor
        if actor is None or actor.lanes != int(lanes):
            cls = vr.VectorAgent57Actor if vr.engine_kind(self.rl_config) == "agent57_light" else vr.VectorActor
            actor = self._vector_actor = cls(env, self.rl_config, self.make_parameter(), int(lanes))
        return actor, (vr.VectorLearner(actor) if with_learner else None)

    def save_parameter(self, path: str, compress: bool = True):
        self.make_parameter().save(path, compress)

    def load_parameter(self, path: str):
        self.make_parameter().load(path)

    def save_memory(self, path: str, compress: bool = True, **kwargs):
        self.make_memory().save(path, compress, **kwargs)

    def load_memory(self, path: str, **kwargs):
        self.make_memory().load(path, **kwargs)

    # ---- play modes ---------------------------------------------------------------------------
    def _base(self, callbacks, **kw) -> RunContext:
        c = self.context.copy()
        c.env_config, c.rl_config = self.env_config, self.rl_config
        for k, v in kw.items():
            setattr(c, k, max(v, 0) if isinstance(v, (int, float)) and k.startswith(("max_", "timeout")) else v)
        c.callbacks = c.callbacks + list(callbacks)
        return c

    def train(self, max_episodes: int = 0, timeout: float = 0, max_steps: int = 0, max_train_count: int = 0, max_memory: int = 0,
              players: list = [], shuffle_player: bool = True, train_interval: int = 1, train_repeat: int = 1,
              enable_progress: bool = True, callbacks: list = []):
        """runner.py:95-183"""
        c = self._base(callbacks, max_episodes=max_episodes, timeout=timeout, max_steps=max_steps, max_train_count=max_train_count,
                       max_memory=max_memory, players=players, shuffle_player=shuffle_player, train_interval=train_interval, train_repeat=train_repeat)
        c.play_mode, c.run_name = "train", "main"
        c.disable_trainer, c.distributed, c.training, c.train_only, c.rollout = False, False, True, False, False
        drivers = self._device_drivers(c)
        if drivers is not None:
            self.state = run_sequence(c, *drivers)
        else:
            self.state = play(c, env=self.make_env(), worker=self.make_worker(), trainer=self.make_trainer())
        return self.state

    def rollout(self, max_episodes: int = -1, timeout: float = -1, max_steps: int = -1, max_memory: int = -1, players: list = [],
                shuffle_player: bool = True, enable_progress: bool = True, callbacks: list = []):
        """Collect experience without training (runner.py:185-252)."""
        c = self._base(callbacks, max_episodes=max_episodes, timeout=timeout, max_steps=max_steps, max_memory=max_memory, players=players,
                       shuffle_player=shuffle_player)
        c.play_mode = "rollout"
        c.max_train_count = 0
        c.disable_trainer, c.distributed, c.training, c.train_only, c.rollout = True, False, True, False, True
        self.state = play(c, env=self.make_env(), worker=self.make_worker(), trainer=None)
        return self.state

    def train_only(self, timeout: float = -1, max_train_count: int = -1, enable_progress: bool = True, callbacks: list = []):
        """Learner only, on whatever the memory already holds (runner.py:254-308)."""
        c = self._base(callbacks, timeout=timeout, max_train_count=max_train_count)
        c.play_mode = "train_only"
        c.max_episodes = c.max_steps = c.max_memory = 0
        c.disable_trainer, c.distributed, c.training, c.train_only, c.rollout = False, False, True, True, False
        self.state = play_trainer_only(c, trainer=self.make_trainer())
        return self.state

    def evaluate(self, max_episodes: int = 10, timeout: float = -1, max_steps: int = -1, players: list = [], shuffle_player: bool = True,
                 enable_progress: bool = True, callbacks: list = []) -> Union[List[float], List[List[float]]]:
        """runner.py:724-799: rewards per episode (flat list for single-player envs)."""
        c = self._base(callbacks, max_episodes=max_episodes, timeout=timeout, max_steps=max_steps, players=players, shuffle_player=shuffle_player)
        c.play_mode, c.run_name = "evaluate", "eval"
        c.max_train_count = c.max_memory = 0
        c.disable_trainer, c.distributed, c.training, c.train_only, c.rollout = True, False, False, False, False
        self.state = play(c, env=self.make_env(), worker=self.make_worker(), trainer=None)
        if self.make_env().player_num == 1:
            return [r[0] for r in self.state.episode_rewards_list]
        return self.state.episode_rewards_list

    def train_mp(self, actor_num: int = 1, queue_capacity: int = 1000, trainer_parameter_send_interval: float = 1,
                 actor_parameter_sync_interval: float = 1, actor_devices: Union[str, List[str]] = "AUTO", enable_mp_memory: bool = True,
                 timeout: float = -1, max_train_count: int = -1, players: list = [], shuffle_player: bool = True,
                 enable_progress: bool = True, callbacks: list = [], **kwargs):
        """N actor processes -> 1 learner (runner.py:310-402, play_mp.py:471-642)."""
        from simple_distributed_rl_amd.base.run import play_mp

        c = self._base(callbacks, timeout=timeout, max_train_count=max_train_count, players=players, shuffle_player=shuffle_player)
        c.play_mode = "train_mp"
        c.actor_num, c.actor_devices = actor_num, actor_devices
        c.max_episodes = c.max_steps = c.max_memory = 0
        c.disable_trainer, c.distributed, c.training, c.train_only, c.rollout = False, True, True, False, False
        self.setup_rl_config()
        if self._vector_envs != 0:
            c.setup_device()
            if str(c.used_device_torch).startswith("cuda"):
                from simple_distributed_rl_amd.device import vector_runner as vr

                self.vector_reason = vr.why_not_vector(c, self.make_env(), self.rl_config)
                if not self.vector_reason:  # one process per GPU over RCCL (device/mp_runner.py); this process is the learner rank
                    from simple_distributed_rl_amd.device.mp_runner import train_mp_on_engine

                    lanes = self._vector_envs
                    if isinstance(lanes, str):
                        lanes = 1024 if hasattr(type(self.make_env().unwrapped), "device_vector") else 64
                    # memory_device="cuda:k": the reference's enable_mp_memory topology with the replay on a GPU of its own (device/replay_role.py)
                    self.state = train_mp_on_engine(self, c, int(lanes), actor_num, actor_devices, updates_per_step=int(kwargs.get("updates_per_step", 1)),
                                                    sync_interval_steps=int(kwargs.get("sync_interval_steps", 16)), memory_device=kwargs.get("memory_device"),
                                                    prefetch=int(kwargs.get("mem_to_train_queue_capacity", 5)),
                                                    actor_initial_priority=bool(kwargs.get("actor_initial_priority", True)))
                    return self.state
            else:
                self.vector_reason = "the run is not on a GPU device"
        mp_cfg = play_mp.MpConfig(c, [], queue_capacity=queue_capacity, trainer_parameter_send_interval=trainer_parameter_send_interval,
                                  actor_parameter_sync_interval=actor_parameter_sync_interval)
        if enable_mp_memory:  # the reference's default: the replay in a process of its own (play_mp_memory.py:595-796)
            from simple_distributed_rl_amd.base.run import play_mp_memory

            self.state = play_mp_memory.train(mp_cfg, self.make_parameter(), self.make_memory(),
                                              play_mp_memory.MemoryLink(int(kwargs.get("mem_to_train_queue_capacity", 5)), int(kwargs.get("train_to_mem_queue_capacity", 100)),
                                                                        bool(kwargs.get("return_memory_data", False))))
        else:
            self.state = play_mp.train(mp_cfg, self.make_parameter(), self.make_memory())
        self._trainer = None
        return self.state
