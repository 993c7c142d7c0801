"""The device drivers of the sequence loop: `Runner.train()` on the hand-written engine.

`srl.Runner(env, rainbow.Config(...)).train(...)` in the reference plays ONE environment through a Python loop
(srl/runner/runner.py:95-183 -> srl/base/run/core_play.py:115-214).  When the algorithm is the Rainbow family on a
GPU device and the environment produces image frames, the same call here hands the loop
(`base/run/sequence.py`) two drivers that own a `RainbowEngine`:

    VectorActor    `lanes` = E environments per iteration: uint8 frame ring -> matrix-core Q-network ->
                   epsilon-greedy -> environments -> ring commit + PER add, all enqueued, nothing read back
    VectorLearner  the updates the loop owes (`train_interval` / `train_repeat`), enqueued on the engine's learner
                   stream so that they run beside the NEXT lock-step's network pass

so `RunCallback` hooks, stop rules (`max_steps`, `max_train_count`, `max_memory`, `max_episodes`, `timeout`) and
the `RunState` counters behave as in the reference's loop, with `state.total_step` advancing by E per iteration.
Episode results stay on the device (`EpisodeLedger`, libsrlx `srlx_episode_account`): the host learns about them
through a pinned mailbox without synchronising the stream, and reads individual episodes only when somebody asks
(an `on_episode_end` hook, `max_episodes`, the end of the run).

Environments: a class registered under an environment id may offer `device_vector(replay, **kwargs)` returning a
device-resident batch environment (the built-in "SyntheticAtari-v0" does); every other image environment is
stepped on the host in E copies and its frames uploaded each lock-step (`HostVecEnv`).
"""
import ctypes
from typing import List, Optional, Tuple

import numpy as np
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.base.define import SpaceTypes
from simple_distributed_rl_amd.base.run.sequence import ActorDriver, LearnerDriver


# ---------------------------------------------------------------------------------------------
# episode results without leaving HBM
# ---------------------------------------------------------------------------------------------
class EpisodeLedger:
    """Running return/length per environment + a ring of finished episodes + running totals, all in HBM
    (env_run.py:334-352 and core_play.py:200-214 for E environments)."""

    SLOTS = 4

    def __init__(self, n_envs: int, device: torch.device, ring_cap: int = 1 << 16):
        self.E, self.dev, self.cap = int(n_envs), device, int(ring_cap)
        self.lib = N.lib()
        self.ep_return = torch.zeros(self.E, dtype=torch.float32, device=device)
        self.ep_len = torch.zeros(self.E, dtype=torch.int32, device=device)
        self.ring = torch.zeros((self.cap, 2), dtype=torch.float32, device=device)
        self.totals = torch.zeros(4, dtype=torch.int64, device=device)  # episodes, steps, float64 bits of return sum, length sum
        self._mail = torch.zeros((self.SLOTS, 4), dtype=torch.int64).pin_memory()
        self._posted = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self._n_posts = 0
        self._last = (0, 0, 0.0, 0)
        self.read_upto = 0  # episodes already handed to the host

    def clear(self):
        """A new run: counters and the finished-episode ring start over (the lanes' running episodes continue)."""
        torch.cuda.current_stream(self.dev).synchronize()
        self.totals.zero_()
        self._n_posts, self._last, self.read_upto = 0, (0, 0, 0.0, 0), 0

    def account(self, rewards: torch.Tensor, done: torch.Tensor, skip_ptr=None):
        N.check(self.lib.srlx_episode_account(self.E, N.tptr(rewards), N.tptr(done), skip_ptr, N.tptr(self.ep_return), N.tptr(self.ep_len),
                                              N.tptr(self.ring), self.cap, N.tptr(self.totals), N.torch_stream_ptr()))

    def post(self):
        """Enqueue a copy of the totals into the next mailbox slot (no synchronisation)."""
        k = self._n_posts % self.SLOTS
        self._mail[k].copy_(self.totals, non_blocking=True)
        self._posted[k].record()
        self._n_posts += 1

    @staticmethod
    def _decode(row) -> Tuple[int, int, float, int]:
        return int(row[0]), int(row[1]), float(row[2:3].view(torch.float64)[0]), int(row[3])

    def peek(self, wait: bool = False) -> Tuple[int, int, float, int]:
        """(episodes, env steps, return sum, length sum) of the newest mailbox slot whose copy has landed; `wait`
        blocks for the newest post.  A slot whose event has completed has no later write in flight (a later post into the
        same slot re-records its event), so its four values are consistent."""
        for back in range(min(self.SLOTS, self._n_posts)):
            k = (self._n_posts - 1 - back) % self.SLOTS
            if wait and back == 0:
                self._posted[k].synchronize()
            if self._posted[k].query():
                self._last = self._decode(self._mail[k])
                break
        return self._last

    def drain(self) -> List[Tuple[float, int]]:
        """Synchronises and returns the (return, length) records of the episodes that finished since the last drain
        (at most the ring's capacity: older ones were overwritten)."""
        self.post()
        episodes = self.peek(wait=True)[0]
        first = max(self.read_upto, episodes - self.cap)
        out: List[Tuple[float, int]] = []
        if episodes > first:
            idx = torch.arange(first, episodes, device=self.dev) % self.cap
            rows = self.ring[idx].cpu().numpy()
            out = [(float(r), int(l)) for r, l in rows]
        self.read_upto = episodes
        return out


# ---------------------------------------------------------------------------------------------
# environments
# ---------------------------------------------------------------------------------------------
class HostVecEnv:
    """E host copies of an image environment behind the engine's batch-environment contract (`reset()` ->
    uint8 [E, F] first frames; `step(actions)` -> next_obs / rewards / terminated / done device tensors).  A lane whose
    episode ended gets its environment reset on the NEXT lock-step, which then only delivers the new episode's first frame
    (the store's `needs_reset` protocol, srlx_store_commit_step)."""

    capturable = False  # step() synchronises with the host

    def __init__(self, env_config, n_envs: int, device: torch.device, seed: Optional[int] = None, processor=None):
        """processor: an ImageProcessor whose space has been remapped from this environment's (raw uint8 frames, e.g. ALE's 210 x 160 x 3):
        the raw frames are uploaded as they are and ONE srlx_image_preprocess launch per lock-step turns them into the ring's gray frames
        -- the reference runs OpenCV per frame on the host and hands the network float32 (image_processor.py:104-151)."""
        from simple_distributed_rl_amd.base.env.registration import make as make_env_run

        self.envs = [make_env_run(env_config) for _ in range(n_envs)]
        self.E, self.dev = n_envs, device
        self.seed = seed
        self.processor = processor
        sp = self.envs[0].observation_space
        if processor is not None:
            self._raw_shape = tuple(sp.shape)
            self._host_raw = torch.zeros((n_envs,) + self._raw_shape, dtype=torch.uint8).pin_memory()
            self.F = int(processor._out_hw[0] * processor._out_hw[1])
        else:
            self.F = int(np.prod(sp.shape))
        self._scale = 255.0 if float(np.max(sp.high)) <= 1.0 else 1.0  # "0to1" frames (image_processor.py:140-142) back to bytes
        self._host_obs = torch.zeros((n_envs, self.F), dtype=torch.uint8).pin_memory()
        self._host_scal = torch.zeros((n_envs, 3), dtype=torch.float32).pin_memory()
        self.next_obs = torch.zeros((n_envs, self.F), dtype=torch.uint8, device=device)
        self.rewards = torch.zeros(n_envs, dtype=torch.float32, device=device)
        self.terminated = torch.zeros(n_envs, dtype=torch.uint8, device=device)
        self.done = torch.zeros(n_envs, dtype=torch.uint8, device=device)
        self._needs_reset = [False] * n_envs
        self._episodes = 0

    def _bytes(self, frame) -> np.ndarray:
        return np.rint(np.asarray(frame, np.float32).reshape(-1) * self._scale).astype(np.uint8)

    def _take(self, i: int):
        """Lane i's current frame into the staging row: raw bytes when a device processor follows, ring bytes otherwise."""
        if self.processor is not None:
            self._host_raw[i] = torch.from_numpy(np.ascontiguousarray(self.envs[i].state, dtype=np.uint8).reshape(self._raw_shape))
        else:
            self._host_obs.numpy()[i] = self._bytes(self.envs[i].state)

    def _upload(self, dst: torch.Tensor) -> torch.Tensor:
        if self.processor is not None:
            raw = self._host_raw.to(self.dev, non_blocking=True)
            self.processor.preprocess_batch(raw, out_u8=dst.view((self.E,) + tuple(self.processor._out_hw)))
            self._raw_keep = raw  # alive until the kernel has read it
        else:
            dst.copy_(self._host_obs, non_blocking=True)
        return dst

    def _reset_lane(self, i: int):
        seed = None if self.seed is None else self.seed + self._episodes
        self._episodes += 1
        self.envs[i].reset(seed=seed)
        self._take(i)

    def setup(self, context):
        for e in self.envs:
            e.setup(context)

    def teardown(self):
        for e in self.envs:
            e.teardown()

    def reset(self) -> torch.Tensor:
        for i in range(self.E):
            self._reset_lane(i)
        first = self._upload(torch.zeros((self.E, self.F), dtype=torch.uint8, device=self.dev))
        torch.cuda.current_stream(self.dev).synchronize()
        return first

    def step(self, actions: torch.Tensor):
        from simple_distributed_rl_amd.base.define import DoneTypes

        acts = actions.cpu().numpy()  # also orders this call after the previous lock-step's uploads
        scal = self._host_scal.numpy()
        for i, env in enumerate(self.envs):
            if self._needs_reset[i]:
                self._reset_lane(i)
                scal[i] = 0.0
                self._needs_reset[i] = False
                continue
            env.step(int(acts[i]))
            self._take(i)
            scal[i, 0] = env.reward
            scal[i, 1] = 1.0 if env.done_type == DoneTypes.TERMINATED else 0.0
            scal[i, 2] = 1.0 if env.done else 0.0
            self._needs_reset[i] = env.done
        self._upload(self.next_obs)
        dev_scal = self._host_scal.to(self.dev, non_blocking=True)
        self.rewards.copy_(dev_scal[:, 0])
        self.terminated.copy_(dev_scal[:, 1].to(torch.uint8))
        self.done.copy_(dev_scal[:, 2].to(torch.uint8))
        return self.next_obs, self.rewards, self.terminated, self.done


# ---------------------------------------------------------------------------------------------
# eligibility + configuration
# ---------------------------------------------------------------------------------------------
def _image_hw(space) -> Optional[Tuple[int, int]]:
    stype = getattr(space, "stype", None)
    shape = tuple(getattr(space, "shape", ()))
    if stype == SpaceTypes.GRAY_HW and len(shape) == 2:
        return shape
    if stype == SpaceTypes.GRAY_HW1 and len(shape) == 3 and shape[2] == 1:
        return shape[:2]
    return None


def why_not_vector(context, env, rl_config) -> str:
    """Empty string when `Runner.train()` can run on the device engine; otherwise the reason it stays on the plugin path."""
    if not str(context.used_device_torch).startswith("cuda"):
        return "the run is not on a GPU device"
    kind = engine_kind(rl_config)
    if kind is None:
        return f"no device engine for algorithm '{rl_config.get_name()}'"
    if env.player_num != 1:
        return "multi-player environment"
    from simple_distributed_rl_amd.base.spaces.discrete import DiscreteSpace

    if not isinstance(env.action_space, DiscreteSpace) or env.action_space.n > 32:
        return "the engine serves discrete action spaces of at most 32 actions"
    if getattr(rl_config, "_obs_processors", None) and frame_processor(rl_config) is None:
        return "observation processors other than one ImageProcessor over uint8 frames are served by the plugin path"
    hw = _image_hw(frame_space(env, rl_config))
    if hw is None or hw[0] < 8 or hw[1] < 8:
        return "observations are not single-channel image frames (after the config's ImageProcessor, if any)"
    mem = rl_config.memory
    if mem.name not in ("Proportional", "Proportional_cpp", "ReplayBuffer"):
        return f"no device replay for memory '{mem.name}'"
    if mem.enable_demo_memory:
        return "demo memory is served by the plugin memory"
    if kind == "agent57_light":  # torch networks: any DQN-image / dueling shape the plugin builds
        if getattr(rl_config.input_block, "image", None) is None or rl_config.input_block.image.name != "DQN":
            return "input block is not the DQN image block"
        return ""
    if rl_config.window_length != 4:
        return "the matrix-core network reads a window of 4 frames"
    ib = getattr(rl_config.input_block, "image", None)
    if ib is None or ib.name != "DQN" or ib.kwargs.get("filters", 32) != 32 or str(ib.kwargs.get("activation", "relu")).lower() != "relu":
        return "input block is not the DQN image block (32 filters, ReLU)"
    hb = rl_config.hidden_block
    sizes = tuple(hb.kwargs.get("layer_sizes", ()))
    if hb.name != "DuelingNetwork" or len(sizes) != 1 or sizes[0] % 32 != 0 or sizes[0] > 512:
        return "hidden block is not one dueling layer of a multiple of 32 (<= 512) units"
    if hb.kwargs.get("dueling_kwargs", {}).get("dueling_type", "average") not in ("average", ""):
        return "the hand-written gradient step covers the dueling types 'average' and ''"
    if rl_config.batch_size > 64:
        return "the hand-written gradient step covers batches of at most 64"
    return ""


def engine_kind(rl_config) -> Optional[str]:
    """Which device engine serves this algorithm config (None = the plugin classes only)."""
    return {"Rainbow": "rainbow", "Rainbow_no_multisteps": "rainbow", "Agent57_light": "agent57_light"}.get(rl_config.get_name())


def frame_space(env, rl_config):
    """The observation space the algorithm sees per step: the environment's, remapped by the config's observation processors."""
    procs = getattr(rl_config, "_obs_processors", None)
    return procs[-1][2] if procs else env.observation_space


def frame_processor(rl_config):
    """The (single) ImageProcessor of the config that a host-stepped batch environment can run on the device, or None."""
    from simple_distributed_rl_amd.rl.processors.image_processor import ImageProcessor

    procs = getattr(rl_config, "_obs_processors", None) or []
    if len(procs) == 1 and isinstance(procs[0][0], ImageProcessor) and "int" in str(np.dtype(procs[0][1].dtype)):
        return procs[0][0]
    return None


def device_config_from(rl_config, env, n_envs: int, seed: int):
    """rainbow.Config (srl/algorithms/rainbow/rainbow.py:57-114) -> the engine's configuration."""
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

    mem = rl_config.memory
    prop = mem.name != "ReplayBuffer"
    kw = mem.kwargs if prop else {}
    hw = _image_hw(frame_space(env, rl_config))
    hb = rl_config.hidden_block
    return RainbowDeviceConfig(
        batch_size=rl_config.batch_size, epsilon=rl_config.epsilon, test_epsilon=rl_config.test_epsilon, lr=rl_config.lr, discount=rl_config.discount,
        target_model_update_interval=rl_config.target_model_update_interval, enable_reward_clip=rl_config.enable_reward_clip,
        enable_double_dqn=rl_config.enable_double_dqn, enable_noisy_dense=rl_config.enable_noisy_dense, enable_rescale=rl_config.enable_rescale,
        multisteps=rl_config.multisteps, retrace_h=rl_config.retrace_h, window_length=rl_config.window_length,
        memory_capacity=mem.capacity, memory_warmup_size=mem.warmup_size,
        # the uniform ReplayBuffer (priority_memories/replay_buffer.py:10-55) is the alpha = 0 corner of the sum-tree: every leaf
        # weighs 1, so every importance weight is 1 whatever beta is; like random.sample its draws are WITHOUT replacement (a second hit
        # of an item is rejected in draw order, proportional_memory.py:153-157 -- the has_duplicate=False rule of the proportional memory)
        memory_has_duplicate=bool(kw.get("has_duplicate", True)) if prop else False,
        memory_alpha=float(kw.get("alpha", 0.0)), memory_beta_initial=float(kw.get("beta_initial", 0.4)),
        memory_beta_steps=int(kw.get("beta_steps", 1_000_000)), memory_epsilon=float(kw.get("epsilon", 1e-4)),
        hidden_units=int(hb.kwargs["layer_sizes"][0]), dueling_type=hb.kwargs.get("dueling_kwargs", {}).get("dueling_type", "average"),
        filters=32, obs_hw=tuple(hw), n_actions=env.action_space.n, n_envs=n_envs, seed=seed,
    )


class _ReplayFacade:
    """What the loop and callbacks ask of `state.memory` (length for the max_memory rule and progress lines)."""

    def __init__(self, replay):
        self._replay = replay

    def length(self) -> int:
        return self._replay.length()

    def __getattr__(self, item):
        return getattr(self._replay, item)


# ---------------------------------------------------------------------------------------------
# drivers
# ---------------------------------------------------------------------------------------------
class VectorActor(ActorDriver):
    def __init__(self, env_run, rl_config, parameter, n_envs: int, overlap: bool = True, use_graphs: bool = True):
        self.env_run, self.rl_config, self.parameter = env_run, rl_config, parameter
        self.lanes = int(n_envs)
        self.overlap, self.use_graphs = overlap, use_graphs
        self.engine = None
        self._eps_sched = None
        self._iteration = 0
        self._graphs_ready = False

    # -- set-up ---------------------------------------------------------------------------------
    def _make_batch_env(self, replay, context):
        base = self.env_run.unwrapped
        maker = getattr(type(base), "device_vector", None)
        if maker is not None:
            return maker(replay, **self.env_run.config.kwargs)
        env = HostVecEnv(self.env_run.config, self.lanes, replay.dev, context.seed, processor=frame_processor(self.rl_config))
        env.setup(context)
        return env

    def open(self, context, state):
        """(everything happens in `attach`: the engine draws from its own seeded generators, not from the process-wide ones on_start may precede)"""

    def attach(self, context, state):
        from simple_distributed_rl_amd.device.rainbow import RainbowEngine

        dev = torch.device(context.used_device_torch)
        if self.engine is None:  # the engine (replay included) lives as long as the Runner: a second train() continues on the same memory
            seed = 0 if context.seed is None else int(context.seed)
            self.cfg = device_config_from(self.rl_config, self.env_run, self.lanes, seed)
            self.engine = RainbowEngine(self.cfg, dev.index or 0, env=lambda replay: self._make_batch_env(replay, context), overlap=self.overlap)
        eng = self.engine
        self._load_weights()
        if eng.ledger is None:
            eng.ledger = EpisodeLedger(self.lanes, dev)  # same tensors for the engine's lifetime: the captured commit graph points at them
        eng.ledger.clear()
        self._eps_sched = None if self.rl_config.enable_noisy_dense else self.rl_config.epsilon_scheduler.create(self.rl_config.epsilon)
        self._eps_now = None
        self._iteration = 0  # every lane is a worker whose step_in_training restarts with the run (worker_run.py setup)
        self._training = bool(context.training)
        if not self._training:
            eng.eps.fill_(float(self.rl_config.test_epsilon))
        state.env, state.worker, state.workers = self.env_run, None, []
        state.parameter, state.memory = self.parameter, _ReplayFacade(eng.replay)
        state.worker_indices = [0]
        state.episode_count = 0
        self._episodes_announced = 0

    def _load_weights(self):
        eng, p = self.engine, self.parameter
        if p is None:
            return
        online, target = p.q_online.state_dict(), p.q_target.state_dict()
        nets = [(eng.q_online, online), (eng.q_target, target)]
        if eng.q_actor is not eng.q_online:
            nets.append((eng.q_actor, online))
        for net, sd in nets:
            if hasattr(net, "load_reference_state_dict"):
                net.load_reference_state_dict(sd)
            else:
                net.load_state_dict(sd)

    def _store_weights(self):
        eng, p = self.engine, self.parameter
        if p is None:
            return
        torch.cuda.synchronize(eng.dev)
        for mine, theirs in ((eng.q_online, p.q_online), (eng.q_target, p.q_target)):
            sd = mine.reference_state_dict() if hasattr(mine, "reference_state_dict") else mine.state_dict()
            theirs.load_state_dict({k: v.to(next(theirs.parameters()).device) for k, v in sd.items()})

    # -- the loop's calls -------------------------------------------------------------------------
    def _book(self, state, records, hooks, fire: bool):
        for ret, length in records:
            state.episode_count += 1
            state.episode_rewards_list.append([ret])
            state.last_episode_rewards = [ret]
            state.last_episode_step = length
            if fire:
                hooks.fire("on_episode_end")

    def roll_episodes(self, context, state, hooks) -> bool:
        if self._iteration == 0:
            hooks.fire("on_episode_begin")  # the E first episodes begin together
        elif hooks.wants("on_episode_begin"):
            while self._episodes_announced < state.episode_count:  # one per episode that ended: its lane started the next one
                self._episodes_announced += 1
                hooks.fire("on_episode_begin")
        if context.max_episodes > 0 and state.episode_count >= context.max_episodes:
            return False
        return True

    def act(self, context, state, hooks):
        eng = self.engine
        if self._eps_sched is not None and self._training:
            eps = self._eps_sched.update(self._iteration).to_float()  # every lane is a worker at its `_iteration`-th step (rainbow.py:311)
            if eps != getattr(self, "_eps_now", None):
                eng.eps.fill_(float(eps))
                self._eps_now = eps
        eng.actor_front()
        state.action = eng.actions
        hooks.fire("on_step_action_after")
        if eng.overlap:
            eng.join_learner()  # the updates forked after the previous lock-step: they must finish before the replay changes
            eng.refresh_actor_copy()
        eng.actor_commit()
        eng.ledger.post()
        state.total_step += self.lanes
        self._iteration += 1

    def settle(self, context, state, hooks):
        led = self.engine.ledger
        exact = context.max_episodes > 0 or hooks.wants("on_episode_end", "on_episode_begin")
        if exact:
            self._book(state, led.drain(), hooks, fire=True)
        else:
            state.episode_count = led.peek()[0]  # lags the device by at most the mailbox depth; exact at close

    def close(self, context, state):
        eng = self.engine
        if eng is None:
            return
        eng.join_learner()
        torch.cuda.synchronize(eng.dev)
        already = len(state.episode_rewards_list)
        records = eng.ledger.drain()
        total = eng.ledger.peek(wait=True)
        state.episode_count = already  # _book counts up from what was booked one by one
        self._book(state, records, None, fire=False)
        state.episode_count = total[0]
        state.shared_vars["env_steps_exact"] = total[1]  # lanes that only received a reset frame are not environment steps
        self._store_weights()

    # -- graphs -----------------------------------------------------------------------------------
    def ensure_graphs(self):
        """Capture the lock-step's launch-bound parts and the whole update into HIP graphs, once the replay is warm (the
        learner's arenas are sized by one eager update first; that update is a real one and is counted by the caller)."""
        if self._graphs_ready or not self.use_graphs:
            return 0
        eng = self.engine
        before = eng.train_count
        eng.join_learner()
        # a host-stepped batch environment reads the actions back every lock-step: its step cannot live in a graph
        eng.capture_graphs(actor=getattr(eng.env, "capturable", True), learner=True, warm_actor=False, warm_learner=True)
        self._graphs_ready = True
        return eng.train_count - before


class VectorLearner(LearnerDriver):
    """`trainer`-shaped view of the engine's learner: `train_count`, `info`, `train()`."""

    def __init__(self, actor: VectorActor):
        self.actor = actor
        self.info: dict = {}

    @property
    def engine(self):
        return self.actor.engine

    @property
    def train_count(self) -> int:
        return self.engine.train_count if self.engine is not None else 0

    def attach(self, context, state):
        state.trainer = self

    def open(self, context, state):
        pass

    def train(self):
        self.update(1, None)

    def update(self, count: int, state) -> int:
        eng = self.engine
        if eng.replay.is_warmup_needed():
            return 0
        ran = self.actor.ensure_graphs()  # first warm call: one eager update + graph capture
        count -= ran
        if count > 0:
            if eng.overlap:
                ran += eng.fork_learner(count)
            else:
                for _ in range(count):
                    ran += int(eng.learner_step())
        return ran

    def close(self, context, state):
        eng = self.engine
        if eng is not None:
            eng.join_learner()
            torch.cuda.synchronize(eng.dev)
            self.info = eng.info()


class VectorAgent57Actor(VectorActor):
    """`Runner.train()` with Agent57_light on a GPU: E lanes of `Agent57LightEngine` (device/agent57_light.py).  The engine trains the
    Runner's own Parameter object (the five torch networks), so nothing has to be copied back."""

    def attach(self, context, state):
        from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine, why_not_fast
        from simple_distributed_rl_amd.device.agent57_light import Agent57LightEngine

        dev = torch.device(context.used_device_torch)
        if self.engine is None:
            seed = 0 if context.seed is None else int(context.seed)
            # 84 x 84 x 4 configs: every network pass and optimiser step in libsrlx (round 6; the trained networks are written back into the Runner's Parameter at
            # `close`); other geometries: the round-5 engine (image trunks in libsrlx, dense tails in torch), which trains the Parameter's modules in place
            cls = Agent57LightFastEngine if not why_not_fast(self.rl_config) else Agent57LightEngine
            self.engine = cls(self.rl_config, self.lanes, dev.index or 0, seed=seed, env=lambda replay: self._make_batch_env(replay, context), parameter=self.parameter)
        eng = self.engine
        if eng.ledger is None:
            eng.ledger = EpisodeLedger(self.lanes, dev)
        eng.ledger.clear()
        eng.training = bool(context.training)
        self._iteration = 0
        state.env, state.worker, state.workers = self.env_run, None, []
        state.parameter, state.memory = self.parameter, _ReplayFacade(eng.replay)
        state.worker_indices = [0]
        state.episode_count = 0
        self._episodes_announced = 0

    def act(self, context, state, hooks):
        eng = self.engine
        eng.actor_step()
        state.action = eng.actions
        hooks.fire("on_step_action_after")
        eng.ledger.post()
        state.total_step += self.lanes
        self._iteration += 1

    def close(self, context, state):
        eng = self.engine
        if eng is None:
            return
        torch.cuda.synchronize(eng.dev)
        already = len(state.episode_rewards_list)
        records = eng.ledger.drain()
        total = eng.ledger.peek(wait=True)
        state.episode_count = already
        self._book(state, records, None, fire=False)
        state.episode_count = total[0]
        state.shared_vars["env_steps_exact"] = total[1]
        if hasattr(eng, "export_parameter"):  # the all-libsrlx engine trains masters of its own: the Runner's Parameter gets the result
            eng.export_parameter(self.parameter)

    def ensure_graphs(self):
        """The whole update as one HIP graph once the replay is warm (its warm-up updates are real ones and are counted by the caller)."""
        eng = self.engine
        if self._graphs_ready or not self.use_graphs:
            return 0
        before = eng.train_count
        eng.capture_graphs()
        self._graphs_ready = True
        return eng.train_count - before
