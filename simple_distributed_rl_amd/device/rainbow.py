"""Device-resident Rainbow actor/learner (the vectorised engine).

One process drives ONE GPU.  Per step the actor advances E lock-stepped environments
(frame-stack -> Q-network -> epsilon-greedy -> env -> ring commit -> PER add) and the learner does
whole Rainbow updates (PER sample -> n-step gather -> 3 forwards -> fused TD/Huber/priority kernel ->
backward -> Adam -> PER update) without a single device<->host hop; the reference does one hop per
policy() and four per train().

Reference semantics kept (file:line under the reference root):
  srl/algorithms/rainbow/rainbow.py:301-329  Worker.policy          -> VectorActor.step
  srl/algorithms/rainbow/rainbow.py:331-400  Worker.on_step/_add_batch (n-step items, padding) -> DeviceReplay
  srl/algorithms/rainbow/model_torch.py:85-122  Trainer.train       -> Learner.train
  srl/algorithms/rainbow/rainbow.py:185-287  calc_target_q          -> srlx_nstep_td_huber_priority
Field names of RainbowDeviceConfig are those of rainbow.Config (rainbow.py:57-114).
"""
import ctypes
import os
from dataclasses import dataclass, field
from typing import Optional

import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.replay import DeviceReplay
from simple_distributed_rl_amd.device.qnet import DeviceAdam, EngineQNet, QNetInference, check_ranges


@dataclass
class EngineSchedule:
    """How the engine arranges its launches -- nothing here changes a result beyond float32 summation order (the tests compare the variants).  These were
    SRLX_* environment switches through round 5; a schedule is part of the configuration of a run, not of the process environment."""
    fast: Optional[bool] = None  # the round-4/5 lock-step wherever it applies (None), never (False), or raise where it does not (True: as RainbowEngine(fast=True))
    lagged_add: bool = True  # single-GPU fast lock-step: the tree add of lock-step t rides on a side branch of update t + 1 (False: the add behind the join, round 4's order)
    predraw: Optional[bool] = None  # the next update's batch drawn behind this update's write-back (None: on learner-only ranks)
    fc1_neighbour: int = 4  # K splits of the actors' first-dense-layer kernel beside an update = how many CUs its 64 tiles x splits workgroups take.  Re-measure after every change of
    # that kernel: round 6's first float16 version (accumulators shuttled through AGPRs: 8.5 vector instructions per MFMA) was best at 2 (half the chip left to the update: +12 %
    # over 4); once its loop was clean (fragments double-buffered per k-step) 4 is 3 % ahead of 2 and 3 again (profiles/r6_ab_lockstep2.txt, r6_ab_lockstep3.txt).  0: generic count
    dgrad_split: Optional[int] = None  # K splits (1 / 2) of conv3's data-gradient GEMM in the update (None: 2; 1 = the summation order of engines off the fast path.  Beside the actors
    # 2 did nothing while their first dense layer filled the chip (0.4184 / 0.4173 ms, round 5) and is worth 2 % since that layer takes half of it)
    fc1_planes: str = "auto"  # engines off the fast path: operand planes for chip-filling policy passes ("auto": only where no learner shares the GPU; "1" / "0")
    actor_stream: Optional[str] = None  # "low" / "normal" / "high": the actors' side on a stream of that priority level (None: the caller's current stream)
    learner_priority: int = -1  # priority of the learner's launch stream
    fused_adam: bool = True  # Adam inside the launches that finish each gradient (False: gradients written out, srlx_adam_step as a launch -- tests read p.grad)
    fused_adam_rest: bool = True  # ... for the eleven tensors behind the first dense layer's weight too
    fused_td: bool = True  # TD target / Huber / priorities in the backward pass's head kernel (False: a launch of their own)
    autograd_yardstick: bool = False  # the gradient step through torch autograd on float32 pixels: a TEST yardstick, never a fallback
    fused_draw: bool = True  # PER draw + item gather as one launch


@dataclass
class RainbowDeviceConfig:
    # --- rainbow.Config fields (rainbow.py:57-114); defaults = set_atari_config (rainbow.py:116-148)
    batch_size: int = 32
    epsilon: float = 0.1
    test_epsilon: float = 0.0
    lr: float = 0.0000625
    discount: float = 0.99
    target_model_update_interval: int = 32000
    enable_reward_clip: bool = True
    enable_double_dqn: bool = True
    enable_noisy_dense: bool = False
    enable_rescale: bool = False
    multisteps: int = 3
    retrace_h: float = 1.0
    window_length: int = 4
    # --- memory (PriorityReplayBufferConfig, priority_replay_buffer.py:17-60)
    memory_capacity: int = 1_000_000
    memory_warmup_size: int = 80_000
    memory_alpha: float = 0.5
    memory_beta_initial: float = 0.4
    memory_beta_steps: int = 1_000_000
    memory_epsilon: float = 0.0001
    actor_initial_priority: bool = False  # rainbow.py:389-398: new items enter the tree with |n-step target - Q(s_0, a_0)| estimated on the actor side instead of max_priority
    memory_has_duplicate: bool = True  # False: a batch never holds an item twice (the uniform ReplayBuffer's random.sample; has_duplicate=False of the proportional memory)
    # --- model (set_dqn_block + dueling (512,))
    hidden_units: int = 512
    filters: int = 32
    dueling_type: str = "average"  # DuelingNetworkConfig: "average" | "max" | ""
    # --- engine
    obs_hw: tuple = (84, 84)
    n_actions: int = 6
    n_envs: int = 1024
    seed: int = 0
    schedule: EngineSchedule = field(default_factory=EngineSchedule)


class SyntheticAtariVecEnv:
    """E device-resident synthetic environments: uint8 84x84 frames i.i.d.
    U{0..255}, reward in {-1,0,1}, episodes of `episode_len` steps ending `terminated`."""

    def __init__(self, replay: DeviceReplay, episode_len: int = 200):
        self.replay = replay
        self.episode_len = int(episode_len)
        d = replay.dev
        E, F = replay.E, replay.F
        self.next_obs = torch.zeros((E, F), dtype=torch.uint8 if replay.obs_uint8 else torch.float32, device=d)
        self.rewards = torch.zeros(E, dtype=torch.float32, device=d)
        self.terminated = torch.zeros(E, dtype=torch.uint8, device=d)
        self.done = torch.zeros(E, dtype=torch.uint8, device=d)

    def reset(self) -> torch.Tensor:
        g = torch.Generator(device=self.replay.dev)
        g.manual_seed(self.replay.seed)
        if self.replay.obs_uint8:
            return torch.randint(0, 256, self.next_obs.shape, dtype=torch.uint8, device=self.replay.dev, generator=g)
        return torch.rand(self.next_obs.shape, device=self.replay.dev, generator=g) * 2 - 1

    def step(self, actions: torch.Tensor):
        r = self.replay
        # a ring whose tree add lags its commit keeps the device-resident position as the LEARNER's view (it moves with the add, on the learner's stream): the
        # environments belong to the actors' side and take the actors' position -- the host's count of commits -- as a launch argument
        pos = r._steps_committed if r.lagged else -1
        N.check(
            r.lib.srlx_synth_env_step_at(
                r.h_store, pos, self.episode_len, N.tptr(self.next_obs), N.tptr(self.rewards), N.tptr(self.terminated), N.tptr(self.done), N.torch_stream_ptr()
            )
        )
        return self.next_obs, self.rewards, self.terminated, self.done


class RainbowEngine:
    """Actor + learner on one GPU sharing the online network (the reference's sequential `Runner.train`
    topology, core_play.py:115-214, with E environments per iteration).

    Every network pass is hand-written HIP (libsrlx `srlx_qnet_*`): the actors' policy pass, the learner's online /
    target evaluation and the gradient step's forward + backward, with plain dense layers or NoisyLinear ones
    (`enable_noisy_dense`, the reference's own Atari configuration).  `EngineSchedule(autograd_yardstick=True)` swaps the gradient step for
    torch autograd on float32 pixels: a test-only yardstick (tests/test_engine_gpu.py), never a fallback -- shapes the
    kernels do not cover raise."""

    def __init__(self, cfg: RainbowDeviceConfig, device: int = 0, episode_len: int = 200, ring_len: Optional[int] = None, env=None,
                 overlap: bool = False, fast: Optional[bool] = None, actor_stream: Optional[str] = None, role: str = "both", learner_replay: Optional[DeviceReplay] = None):
        """role (device/dist.py): "both" = actors and learner on this GPU; "actor" = a rank that only acts (no target network, no optimiser; its policy passes read
        parameter sets published out of band after every weight broadcast, `on_weights_broadcast`); "learner" = a rank that only learns (never call the actor pieces).
        learner_replay: the replay the LEARNER samples and writes back to when it is not this engine's own ring (a learner rank's global replay); the engine's own ring
        then only stacks frames for its actors: its commits advance the ring position themselves and it gets no tree add.

        overlap=True runs the actor's network pass and the learner update concurrently on two HIP
        streams.  The actor then acts with its own copy of the online network, refreshed after every
        step (the reference's distributed actors do the same on a timer, play_mp.py:121-165), so no
        kernel ever reads weights that another stream is updating.

        actor_stream="low" (the round-4 lock-step only; default: cfg.schedule.actor_stream): the engine creates a LOW-priority HIP stream and makes it the calling thread's
        current stream (torch.cuda.set_stream) -- everything the caller enqueues from now on, the actors' side of the engine included, runs on it.  HIP keeps one
        pool of hardware queues per priority level and replays a graph's branches on normal-priority internal streams, so on a pool of their own the actors never
        queue behind a branch of the update; the update may then run THREE branches wide (the first dense layer's Adam-fused weight gradient on a branch of its own:
        srlx_qnet_set_fc1_branch).  Same kernels, same results; None: the current stream stays what it is and the update stays two branches wide.

        fast (None = cfg.schedule.fast, whose None = wherever it applies; True raises where it does not): the round-4 lock-step for an overlapping engine with
        chip-filling policy passes -- six launches on the actors' stream instead of fifteen and nothing but the PER add behind the join:
          * the policy pass selects its actions in the head kernel (srlx_qnet_forward_u8_policy: `north_star`'s fused policy step);
          * the environments' frames and scalars are one launch, the ring commit is one launch that also writes the next pass's frame-offset table, runs BEFORE
            the join (ring slot p + 1 belongs to no stored item) and leaves the ring position to the PER add (srlx_per_set_add_counters);
          * the actors' "private copy" is one of two published parameter sets (packed filters, first dense layer as bf16 operand planes, small vectors) that the
            UPDATE writes -- the first dense layer from the fused Adam's epilogue, the rest with the filter packing of the learner's own next forward -- so the
            32 MB per-lock-step weight copy and the splitting pass are gone; the actors flip to the new set after the join (a host-side pointer swap).
        Actions, ring and tree are bit-identical to the fifteen-launch path (tests/test_fast_lockstep_gpu.py)."""
        assert role in ("both", "actor", "learner")
        self.cfg = cfg
        self.role = role
        self.learner_replay = learner_replay
        self.overlap = bool(overlap) and role == "both"
        self.dev = torch.device(f"cuda:{device}")
        self.lib = N.lib()
        torch.manual_seed(cfg.seed)
        H, W_ = cfg.obs_hw
        E = cfg.n_envs
        B, n, A = cfg.batch_size, cfg.multisteps, cfg.n_actions
        pad = n + cfg.window_length
        if ring_len is None:
            ring_len = -(-cfg.memory_capacity // E) + pad  # item_len * E >= capacity
        self.noisy = bool(cfg.enable_noisy_dense)
        self.mfma = True  # (kept for callers that used to branch on it: there is no other network path)
        sch = self.schedule = cfg.schedule
        self.autograd_yardstick = bool(sch.autograd_yardstick)
        covered = cfg.filters == 32 and cfg.hidden_units <= 512 and cfg.hidden_units % 32 == 0 and B <= 64 and H == W_ and W_ % 4 == 0 and cfg.dueling_type != "max"
        if not covered and not self.autograd_yardstick:
            raise ValueError("RainbowEngine: the hand-written gradient step covers the DQN image block with 32 filters on square frames (side % 4 == 0), one dueling "
                             f"layer of <= 512 units (average / none) and batches <= 64; got filters={cfg.filters}, hidden={cfg.hidden_units}, batch={B}, "
                             f"frames={cfg.obs_hw}, dueling='{cfg.dueling_type}'.  There is no fallback network path.")
        self.mfma_train = covered and not self.autograd_yardstick
        fused_adam = self.mfma_train and not self.noisy and sch.fused_adam and role != "actor"
        fast_actor = self.fused_convs and not self.noisy and E >= 512 and E % 128 == 0 and (2 * cfg.hidden_units) % 128 == 0
        fast_learner = fused_adam and self.fused_convs and sch.fused_td
        # (actor-side initial priorities on one GPU keep the fifteen-launch lock-step: their tree add runs one lock-step behind the ring commit, which the
        #  deferred-advance commit does not model; a distributed learner -- learner_replay -- commits ring and tree together, two slabs behind: device/dist.py)
        can_fast = {"both": self.overlap and fast_actor and fast_learner and (not cfg.actor_initial_priority or learner_replay is not None),
                    "actor": fast_actor, "learner": fast_learner}[role]
        if fast and not can_fast:
            raise ValueError("RainbowEngine(fast=True): needs overlap, the 84 x 84 x 4 / 32-filter geometry, plain dense layers, >= 512 environments in multiples of 128, "
                             "a hidden layer in multiples of 64 and max-priority adds")
        if fast is None:
            fast = sch.fast
        if fast and not can_fast:
            raise ValueError("RainbowEngine: schedule.fast=True where the fast lock-step does not apply")
        self.fast = can_fast and (bool(fast) if fast is not None else True)
        # the single-GPU round-5 lock-step (actors + learner here, the learner's replay = this ring): the tree add of a lock-step runs one lock-step behind its ring
        # commit, on a side branch of the next update -- the ring gets one spare slot (DeviceReplay(lagged_add=True)).  SRLX_LAGGED_ADD=0: the add behind the join.
        lag = self.fast and role == "both" and learner_replay is None and sch.lagged_add
        self.replay = DeviceReplay(
            E, ring_len + (1 if lag else 0), H * W_, cfg.window_length, n, A, B, True, cfg.enable_reward_clip,
            cfg.memory_alpha, cfg.memory_beta_initial, cfg.memory_beta_steps, cfg.memory_epsilon, cfg.memory_warmup_size, cfg.seed, device,
            has_duplicate=cfg.memory_has_duplicate, lagged_add=lag, fused_draw=sch.fused_draw,
        )
        if env is None:
            self.env = SyntheticAtariVecEnv(self.replay, episode_len)
        else:  # a ready batch environment, or a factory that needs this engine's replay (device/vector_runner.py)
            self.env = env(self.replay) if callable(env) else env

        self.actor_stream = None
        want = actor_stream or sch.actor_stream
        if self.fast and want and want != "default":
            import ctypes

            raw = ctypes.c_void_p()
            N.check(self.lib.srlx_stream_create({"high": -1, "normal": 0, "low": 1}[want], ctypes.byref(raw)))
            self._actor_stream_raw = raw
            self._stream_before = torch.cuda.current_stream(self.dev)  # `close()` hands the thread back to it
            self.actor_stream = torch.cuda.ExternalStream(raw.value, device=self.dev)
            self.actor_stream.wait_stream(self._stream_before)
            torch.cuda.set_stream(self.actor_stream)

        def make_net():
            return EngineQNet(A, cfg.obs_hw, cfg.window_length, cfg.hidden_units, cfg.filters, cfg.dueling_type, noisy=self.noisy).to(self.dev)

        self.q_online = make_net()
        if role == "actor":  # an actor rank: the online network is the broadcast's landing place, nothing trains here
            self.q_target = self.q_actor = self.q_online
            self.inf_actor = QNetInference(self.q_online, E, device, noise_seed=cfg.seed * 3 + 0xA11CE)
            self.inf_online = self.inf_target = None
            if self.fast:
                self.inf_online = QNetInference(self.q_online, 64, device)  # (packs the filters a publish writes: srlx_qnet_publish is a call on the source handle)
                self.inf_actor.enable_fc1_planes(private_weights=True)
                self.inf_actor.enable_actor_sets()
                self.inf_actor.set_fc1_neighbour(0)  # nobody shares the GPU: CU-filling workgroups (83 against 97-130 us at 1024 rows)
                self._planes_ptr = [self.inf_actor.set_planes_ptr(0), self.inf_actor.set_planes_ptr(1)]
                self._set, self._published, self._seen_versions, self._fresh_set, self._learner_planes = 0, None, None, None, False
            elif not self.noisy and E >= 512 and E % 128 == 0 and (2 * cfg.hidden_units) % 128 == 0:
                self.inf_actor.enable_fc1_planes(private_weights=True)  # (re-split by `on_weights_broadcast`)
            self.mfma_train = False
            self.optimizer = None
            self.actor_priority = bool(cfg.actor_initial_priority)
            self._init_common(cfg, E, B, A, H, W_)
            if self.fast:
                self._publish_out_of_band()
            return
        self.q_target = make_net()
        self.q_target.eval()
        self.q_target.load_state_dict(self.q_online.state_dict())  # model_torch.py:41-42
        self.q_online.train()
        if self.overlap:
            if self.fast:  # the actors read a PUBLISHED set, never these tensors: no second module
                self.q_actor = self.q_online
            else:
                self.q_actor = make_net()
                self.q_actor.load_state_dict(self.q_online.state_dict())
            # high priority: the learner's many small kernels slot in between the actor's chip-filling GEMMs
            self.s_learner = torch.cuda.Stream(device=self.dev, priority=int(sch.learner_priority))
            self._ev_fork = torch.cuda.Event()
            self._ev_join = torch.cuda.Event()
        else:
            self.q_actor = self.q_online
            if learner_replay is not None:  # `run_updates`: a graph three branches wide (online | target | ingest) must be launched from a high-priority stream (tools/README.md, 9)
                self.s_learner = torch.cuda.Stream(device=self.dev, priority=-1)
        # one inference handle per concurrent user (each owns its activation buffers and, for noisy layers, its noise stream)
        self.inf_actor = QNetInference(self.q_actor, E, device, noise_seed=cfg.seed * 3 + 0xA11CE)
        planes = str(sch.fc1_planes)
        if role == "learner":
            pass  # (never acts: no operand planes, no parameter sets)
        elif self.fast:
            self.inf_actor.enable_fc1_planes(private_weights=True)  # (the activation planes; the weight planes the passes read are the published sets')
            self.inf_actor.enable_actor_sets()
            # the passes run BESIDE the update: half-CU workgroups in a steady stream instead of one CU-filling workgroup per CU for the whole launch
            # (same-box A/B of the lock-step, tools/_r4_probe3.sh: 0.508 ms with 4 K splits -- 256 workgroups that leave half of every CU to the update --, 0.515
            # with 8, 0.528 with 16, 0.545 with the CU-filling kernel; SRLX_FC1_NEIGHBOUR=0 selects that one, = k the K splits)
            self.inf_actor.set_fc1_neighbour(int(sch.fc1_neighbour))
        elif not self.noisy and E >= 512 and E % 128 == 0 and (2 * cfg.hidden_units) % 128 == 0 and (planes == "1" or (planes == "auto" and not overlap)):
            # Chip-filling policy passes with the first dense layer on pre-split bf16 operand planes (srlx_fc1_planes.hip): the GEMM itself is 1.6x faster
            # (85 against 137 us at 1024 rows), but beside a learner it LOSES: same-box A/B (tools/_ab_lockstep.sh) 0.565 against 0.511 ms per lock-step --
            # the refresh of the actors' copy also has to split the 32 MB weight (22 us on the serial tail every lock-step) and the planes kernel's 144 KB /
            # 512-thread workgroups keep the learner's dependent kernels waiting for compute units like the convolution kernel does.  So: on for an engine
            # that only acts (no update shares the GPU: the actor ranks of device/dist.py), off when actors and learner overlap on one GPU.  SRLX_FC1_PLANES=1 / 0
            # forces it.
            self.inf_actor.enable_fc1_planes(private_weights=self.q_actor is not self.q_online)
        self.inf_online = QNetInference(self.q_online, max(B * (n + 1), 64), device, noise_seed=cfg.seed * 3 + 0x0B0E)
        self.inf_target = QNetInference(self.q_target, B * n, device, noise_seed=cfg.seed * 3 + 0x7A26)
        if self.mfma_train:
            # hand-written training pass (srlx_qnet_backward_u8): one forward over s_0..s_n of every item, gradients of the
            # s_0 rows written straight into p.grad -- no autograd graph, no float32 copy of s_0
            self.inf_online.enable_training(B)
            # the target network's pass is independent of the online pass until the TD kernel: its own stream
            self.s_target = torch.cuda.Stream(device=self.dev, priority=-1)
            self._ev_t0, self._ev_t1 = torch.cuda.Event(), torch.cuda.Event()
            self.optimizer = DeviceAdam(self.inf_online._params(), lr=cfg.lr)  # model_torch.py:71 as one libsrlx launch over all parameter tensors
        else:
            self.optimizer = torch.optim.Adam(self.q_online.parameters(), lr=cfg.lr, capturable=True, fused=True)
        self.actor_priority = bool(cfg.actor_initial_priority) and self.mfma
        self._init_common(cfg, E, B, A, H, W_)
        d = self.dev
        self.train_count_dev = torch.zeros(1, dtype=torch.int64, device=d)
        self._fused_td = bool(sch.fused_td)  # TD / Huber / priorities inside the backward's head kernel
        self.lreplay.count_updates_in(self.train_count_dev)  # train_count += 1 rides on the priority write-back's launch
        if fused_adam:
            # the 32 MB first dense layer takes its Adam step inside the backward pass, beside the convolution gradients (A/B switch for measurements)
            self.optimizer.fuse_first_dense(self.inf_online, self.train_count_dev)
        if self.fast:
            if role == "both":
                self._planes_ptr = [self.inf_actor.set_planes_ptr(0), self.inf_actor.set_planes_ptr(1)]
            self._set, self._published = 0, None
            self._seen_versions = None
            self.inf_target.set_pack_sticky(True)  # the target network's packed filters change at a sync only
            self._learner_planes = False
            self._fresh_set = None  # the set whose planes equal the online network's current weight (None: some update did not publish)
            # the priority write-back leaves the update's critical path: it needs the head kernel's priorities only, so it is the FIRST launch of the backward pass's
            # weight-gradient branch (srlx_qnet_set_priority_sink; no new branch in the graph -- as a branch of its own it put the update on the actors' hardware queue:
            # tools/README.md findings 3, 5); the step count it used to advance moves to the update's LAST launch (the packing / publishing one).
            # (a learner whose batches are SERVED by a replay GPU -- device/replay_role.py -- has no tree to write into: its priorities leave in a message, behind the update)
            self._update_side = self._fused_td and getattr(self.lreplay, "h_per", None) is not None
            if self.actor_stream is not None and want == "low":  # the actors cannot queue behind a branch of the update: it may run three wide
                N.check(self.lib.srlx_qnet_set_fc1_branch(self.inf_online.h, 2))
            # (a learner-only rank: the first dense layer's weight gradient LAST on the weight-gradient branch, order 0 -- with the critical chain recorded first,
            # 0.300 ms per period against 0.343 on a branch of its own; same box, profiles/r5_ab_ingest_order.txt)
            N.check(self.lib.srlx_qnet_set_main_first(self.inf_online.h, 1))
            if sch.dgrad_split is None or int(sch.dgrad_split) == 2:  # conv3's data-gradient GEMM as twice the workgroups, each half as long (-2 % per period / per lock-step)
                N.check(self.lib.srlx_qnet_set_dgrad_split(self.inf_online.h, 2))
            if self._update_side:
                N.check(self.lib.srlx_per_set_update_counter(self.lreplay.h_per, None))
            if fused_adam and not self.noisy and sch.fused_adam_rest:
                # no optimiser launch on the update's tail: the remaining eleven tensors take their steps in the launches that finish their gradients and in the
                # packing launch (`publish_to` follows every backward pass of a fast engine)
                self.optimizer.fuse_rest(self.inf_online)
            if role == "both" and learner_replay is None:
                self.replay.enable_deferred_advance()
            self._publish_out_of_band()
        self.target = torch.zeros(B, dtype=torch.float32, device=d)
        self.loss = torch.zeros(1, dtype=torch.float32, device=d)
        self.grad_q0 = torch.zeros((B, A), dtype=torch.float32, device=d)
        self.priorities = torch.zeros(B, dtype=torch.float32, device=d)
        self._learner_graph = None
        self._learner_graphs = {}
        self._learner_pending = False
        self._capturing = self._in_capture = False
        # a learner rank's ingest (device/dist.py): the commit of transitions that arrived from other ranks runs on a side stream between the update's draw and
        # its priority write-back -- `ingest` = (key, callable issuing the launches) for the NEXT update only
        self.ingest = None
        self.s_ingest = torch.cuda.Stream(device=self.dev, priority=-1) if (learner_replay is not None or self.replay.lagged) else None
        self._ev_drawn, self._ev_ingested, self._ev_sunk, self._ev_predrawn = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        for ev in (self._ev_drawn, self._ev_ingested, self._ev_sunk, self._ev_predrawn):
            ev.record()  # (torch creates the HIP event at the first record: the library keeps raw handles of two of them)
        # pre-draw: the NEXT update's batch is drawn right behind this update's write-back, into the replay's other buffer set, instead of at the head of the next
        # update's chain.  On for a learner rank's replay, where the update's own chain is the period (0.338 -> 0.323 ms per period alone on a GPU, same box); beside
        # this GPU's own actors the period is contention-bound and the move changed nothing (0.457 against 0.454 ms, same box; 0.441 against 0.434 for the 2-GPU
        # topology's acting learner rank): on for learner-ONLY ranks unless SRLX_PREDRAW says otherwise.
        self._predraw = bool(self.fast and self._update_side and self.s_ingest is not None and getattr(self.lreplay, "two_sets", False)
                             and (sch.predraw if sch.predraw is not None else (learner_replay is not None and role == "learner")))
        self._bset, self._drawn, self._drawn_at = 0, None, 0
        self._seen_commits, self._unnoted = 0, 0  # ring commits issued on the device that the host's count (`note_commit`) has not caught up with
        self.s_predraw = torch.cuda.Stream(device=self.dev, priority=-1) if self._predraw else None

    @property
    def lreplay(self) -> DeviceReplay:
        """The replay the learner samples and writes back to."""
        return self.learner_replay if self.learner_replay is not None else self.replay

    def _init_common(self, cfg, E, B, A, H, W_):
        """What every role needs: the actors' buffers and the first observations."""
        d = self.dev
        # ---- actor-side initial priorities (cfg.actor_initial_priority; the reference's distributed worker, rainbow.py:389-398) ----
        # The Q rows of the last n + 1 acting passes are kept; one lock-step after an item was committed -- its last state s_n has then been evaluated by
        # the pass that acts on it -- the existing fused TD kernel turns the cached rows + the item's stored actions / rewards into |target - Q(s_0, a_0)|
        # (ONE launch: srlx_store_actor_td).  On one GPU the item's leaf is ADDED then (the add of a lock-step is deferred by one) with (|td| + eps)^alpha; an
        # actor rank ships the estimates with its next slab and the learner rank commits ring and tree together, two slabs behind (device/dist.py).  The cached online
        # rows stand in for the target network too (an actor holds no target network here; equal right after a target sync) and are as old as the pass that
        # produced them (the reference re-evaluates all n + 1 states under the current weights: identical while the weights stand still, tests/test_engine_gpu.py).
        # Items whose window touches an episode end keep max_priority: the state after a terminal / truncated step is never evaluated by an actor.
        if self.actor_priority:
            n1 = cfg.multisteps + 1
            self.q_hist = torch.zeros((n1, E, cfg.n_actions), dtype=torch.float32, device=d)
            self._passes, self._ap_first_slot = 0, None
            self._ap = dict(pri=torch.zeros(E, dtype=torch.float32, device=d), mask=torch.zeros(E, dtype=torch.uint8, device=d))
        self._select_graph = None
        self._commit_graph = None
        self._learner_pending = False
        self.train_count = 0
        self.sync_count = 0
        self.total_env_steps = 0
        # noisy nets act greedily (rainbow.py:305-309): epsilon-greedy with epsilon = 0
        self.eps = torch.full((E,), 0.0 if self.noisy else float(cfg.epsilon), dtype=torch.float32, device=d)
        self.actions = torch.zeros(E, dtype=torch.int32, device=d)
        self.u_policy = torch.zeros(2 * E, dtype=torch.float64, device=d)
        self.policy_counter = torch.zeros(1, dtype=torch.int64, device=d)
        self._img = (cfg.window_length, H, W_)
        self.before_env = None  # optional hook between the policy pass and the environments' step
        self.ledger = None  # optional EpisodeLedger (device/vector_runner.py): per-episode returns without leaving HBM
        self._own_ring_only = self.learner_replay is not None or self.role == "actor"  # the engine's ring only stacks frames: no tree add, the commit moves the position
        self.first_obs = self.env.reset()
        self.replay.reset_all(self.first_obs)

    def close(self):
        """Hands the calling thread back to the stream it was on before the engine took it to its actors' stream (`actor_stream=`), and destroys that stream."""
        if getattr(self, "actor_stream", None) is not None:
            torch.cuda.synchronize(self.dev)
            torch.cuda.set_stream(self._stream_before)
            self.actor_stream = None
            N.check(self.lib.srlx_stream_destroy(self._actor_stream_raw))

    # ---- actor (rainbow.py:301-329 + 331-400 for E envs) --------------------------------------
    def _actor_net(self, obs=None, events=None):
        """Q-values of all E environments: frame-offset table + srlx_qnet_forward_u8 straight from the uint8 ring
        (`events` bracket the network kernels)."""
        off = self.replay.frame_table_current()
        if events is not None:
            events[0].record()
        q = self.inf_actor.forward_u8(self.replay.obs_base, off)
        if events is not None:
            events[1].record()
        if self.actor_priority:  # the rows the initial priorities are estimated from
            self.q_hist[self._passes % self.q_hist.shape[0]].copy_(q[: self.cfg.n_envs])
        return q

    def _actor_select(self, q):
        """epsilon-greedy over the Q rows (epsilon = 0 for noisy nets), then the environments step: writes nothing shared."""
        cfg = self.cfg
        st = N.torch_stream_ptr()
        N.check(self.lib.srlx_rng_uniform(cfg.seed ^ 0xAC7, N.tptr(self.policy_counter), self.u_policy.numel(), N.tptr(self.u_policy), st))
        N.check(self.lib.srlx_policy_epsilon_greedy(cfg.n_envs, cfg.n_actions, N.tptr(q), N.tptr(self.eps), N.tptr(self.u_policy), None, N.tptr(self.actions), st))
        self.env.step(self.actions)

    def _actor_commit(self):
        """ring commit + PER add of the lock-step the front produced (the only actor writes to the replay)."""
        e = self.env
        if self.ledger is not None:  # before the commit: the store's needs_reset view still marks the lanes that only received a first frame
            self.ledger.account(e.rewards, e.done, self.replay.needs_reset_ptr)
        self.replay.commit(self.actions, e.rewards, e.terminated, e.done, e.next_obs, defer_add=self.actor_priority)

    # ---- the round-4 lock-step (self.fast) -----------------------------------------------------------------------------------------------------
    def _publish_out_of_band(self):
        """Set `self._set` := the online network as it stands (packed filters, small vectors AND a splitting pass over the first dense layer): start-up and after
        weights were loaded from outside (a state dict, a broadcast); the target handle re-packs too.  Runs on the current stream, nothing of the engine in flight."""
        self.inf_online.weights_changed()
        if self.role == "learner":
            self.inf_online.publish_to(None)
        else:
            self.inf_online.publish_to(self.inf_actor, self._set, with_fc1=True)
            self.inf_actor.select_set(self._set)
        if self.inf_target is not None:
            self.inf_target.weights_changed()
            self.inf_target.publish_to(None)
        self._fresh_set = self._set
        self._published = None
        self._seen_versions = (self.q_online.weights_version, self.q_target.weights_version)

    def on_weights_broadcast(self):
        """A broadcast (device/dist.py) has just overwritten the parameters in place: whatever was derived from them is rebuilt (fast: the published set; else
        the actor handle's operand planes and packed filters)."""
        if self.fast:
            if self.role != "actor":
                self.join_learner()
            self._publish_out_of_band()
        else:
            self.inf_actor.weights_changed()

    def _check_versions(self):
        if self._seen_versions != (self.q_online.weights_version, self.q_target.weights_version):  # a state dict was loaded behind the engine's back
            self.join_learner()
            self._publish_out_of_band()

    def actor_commit_ring(self):
        """fast: the ring half of the commit -- one launch (frames, scalars, item mask, the NEXT pass's frame-offset table, the policy generator's counter); touches
        nothing a running learner reads, so it goes BEFORE the join."""
        e = self.env
        if self.ledger is not None:
            self.ledger.account(e.rewards, e.done, self.replay.needs_reset_ptr)
        self.replay.commit(self.actions, e.rewards, e.terminated, e.done, e.next_obs, defer_add=True, next_table=True, bump=self.policy_counter)
        if self.actor_priority:  # what the estimates of the NEXT pass will need (the items this commit completed)
            r = self.replay
            self._ap["mask"].copy_(r.item_mask)
            self._ap_first_slot = ((r._steps_committed - 1) * self.cfg.n_envs) % r.capacity
            self._passes += 1

    def actor_commit_tree(self):
        """fast: the tree half -- the PER add at max_priority, which also moves the ring position; behind the join (the learner writes the tree and reads the position).
        Nothing where the engine's ring only stacks frames (an actor rank; a learner rank's own actors): that ring's commit has moved the position itself."""
        if not self._own_ring_only and not self.replay.lagged:  # (lagged: the next fork takes the add with it)
            self.replay.add_masked()
        self.total_env_steps += self.cfg.n_envs

    def actor_td_estimates(self):
        """float32 [E] for the items the PREVIOUS lock-step committed, from the cached Q rows (see __init__; ONE launch, srlx_store_actor_td): |n-step target -
        Q(s_0, a_0)|, or -1 where the item's window touches an episode end (the adder uses max_priority), or -2 where the lock-step completed no item for the lane.
        Call after this lock-step's network pass.  None before the first commit."""
        if self._ap_first_slot is None:
            return None
        cfg, r, a = self.cfg, self.replay, self._ap
        n1, T = cfg.multisteps + 1, self._passes  # pass T has just been stored; the items were committed at lock-step T - 1: states s_{T-n} .. s_T
        if T < cfg.multisteps:  # the first lock-steps: some of the window's states were never evaluated (random filling came before) -> max_priority
            a["pri"].copy_(torch.where(a["mask"] != 0, -1.0, -2.0))
            self._ap_first_slot = None
            return a["pri"]
        N.check(self.lib.srlx_store_actor_td(r.h_store, self._ap_first_slot, r.capacity, N.tptr(self.q_hist), (T - cfg.multisteps) % n1, N.tptr(a["mask"]), float(cfg.discount),
                                             float(cfg.retrace_h), int(cfg.enable_double_dqn), int(cfg.enable_rescale), N.tptr(a["pri"]), N.torch_stream_ptr()))
        self._ap_first_slot = None
        return a["pri"]

    def _add_with_actor_priorities(self):
        """The deferred PER add of the PREVIOUS lock-step's items with actor-side initial priorities (srlx_per_add, SRLX_PRIO_EST_F32).  Call after this lock-step's
        network pass and after the learner has been joined (it writes the tree)."""
        est = self.actor_td_estimates()
        if est is not None:
            r = self.replay
            N.check(self.lib.srlx_per_add(r.h_per, r.E, N.tptr(est), N.PRIO_EST_F32, 1, N.torch_stream_ptr()))

    def actor_step(self):
        """One eager lock-step of the actors (no graphs, no learner)."""
        if self.fast:
            self.actor_front()
            self.actor_commit_ring()
            if not self._own_ring_only:
                self.replay.add_masked()  # (lagged: launches the pending add now)
            return
        if self.actor_priority:  # the deferred add and its bookkeeping (mask, first slot, pass count) live in the piecewise calls: take exactly that path
            self.actor_front()
            self._add_with_actor_priorities()
            self.actor_commit()
            self.total_env_steps -= self.cfg.n_envs  # (actor_commit counted the lock-step; callers of actor_step count it themselves)
            return
        self._actor_select(self._actor_net())
        self._actor_commit()

    def random_front(self):
        """Uniformly random actions (epsilon = 1, no network) + the environments' step: the front of a lock-step that fills the replay."""
        cfg = self.cfg
        E = cfg.n_envs
        st = N.torch_stream_ptr()
        if not hasattr(self, "_ones"):
            self._ones = torch.ones(E, dtype=torch.float32, device=self.dev)
            self._zq = torch.zeros((E, cfg.n_actions), dtype=torch.float32, device=self.dev)
        N.check(self.lib.srlx_rng_uniform(cfg.seed ^ 0xF111, N.tptr(self.policy_counter), self.u_policy.numel(), N.tptr(self.u_policy), st))
        N.check(self.lib.srlx_policy_epsilon_greedy(E, cfg.n_actions, N.tptr(self._zq), N.tptr(self._ones), N.tptr(self.u_policy), None, N.tptr(self.actions), st))
        if self.before_env is not None:
            self.before_env()
        return self.env.step(self.actions)

    def _random_rest(self):
        """One lock-step with uniformly random actions: used to fill the replay."""
        next_obs, rewards, terminated, done = self.random_front()
        self.replay.commit(self.actions, rewards, terminated, done, next_obs)

    def enable_lazy_capture(self):
        """From now on every update variant (published set x ingest key) is captured into a HIP graph the first time it runs and replayed afterwards."""
        self._capturing = True

    def prefill(self, randomise_priorities: bool = True):
        """Untimed set-up of the benchmark state: a random-policy rollout until every PER leaf holds an
        item, then |delta| ~ U(0,1) priorities (speedtest.py:40-41)."""
        r, cfg = self.replay, self.cfg
        steps = r.item_len + cfg.multisteps - 1
        for _ in range(steps):
            self._random_rest()
        self.total_env_steps += steps * cfg.n_envs
        if randomise_priorities:
            g = torch.Generator(device=self.dev)
            g.manual_seed(cfg.seed + 1)
            pri = torch.rand(r.capacity, dtype=torch.float32, device=self.dev, generator=g)
            N.check(self.lib.srlx_per_set_range(r.h_per, 0, r.capacity, N.tptr(pri), N.PRIO_F32, 1, N.torch_stream_ptr()))
        torch.cuda.synchronize(self.dev)

    def actor_forward_flops(self) -> float:
        """Algorithmic fp32 FLOPs of one policy-step network pass over E environments (19.97 M MAC per sample for A = 6)."""
        c = self.cfg
        F1 = c.filters
        h1 = (c.obs_hw[0] + 6 - 8) // 4 + 1
        w1 = (c.obs_hw[1] + 6 - 8) // 4 + 1
        h2, w2 = (h1 + 4 - 4) // 2 + 1, (w1 + 4 - 4) // 2 + 1
        mac = h1 * w1 * F1 * c.window_length * 64 + h2 * w2 * 2 * F1 * F1 * 16 + h2 * w2 * 2 * F1 * 2 * F1 * 9
        mac += h2 * w2 * 2 * F1 * 2 * c.hidden_units + c.hidden_units * (1 + c.n_actions)
        return 2.0 * mac * c.n_envs

    def conv_gemm_flops(self, with_conv1: bool = False) -> float:
        """Algorithmic fp32 FLOPs of conv2 + conv3 (+ conv1: the fused kernel computes all three) of one policy-step pass over E environments."""
        c = self.cfg
        F1 = c.filters
        h1 = (c.obs_hw[0] + 6 - 8) // 4 + 1
        w1 = (c.obs_hw[1] + 6 - 8) // 4 + 1
        h2, w2 = (h1 + 4 - 4) // 2 + 1, (w1 + 4 - 4) // 2 + 1
        mac = h2 * w2 * 2 * F1 * F1 * 16 + h2 * w2 * 2 * F1 * 2 * F1 * 9
        if with_conv1:
            mac += h1 * w1 * F1 * c.window_length * 64
        return 2.0 * mac * c.n_envs

    @property
    def fused_convs(self) -> bool:
        """Does the policy pass take the one-kernel conv1 -> conv2 -> conv3 path (srlx_qnet_fused.hip)?"""
        c = self.cfg
        return tuple(c.obs_hw) == (84, 84) and c.window_length == 4 and c.filters == 32 and os.environ.get("SRLX_NO_FUSED_CONV", "0") != "1"

    # ---- learner (model_torch.py:85-122) -----------------------------------------------------
    def _learner_body(self, publish: Optional[int] = None, ingest=None, bset: Optional[int] = None, have_batch: bool = False, predraw: bool = False):
        """One Rainbow update.  fast engines: `publish` = the actor set (0 / 1) this update also writes -- the first dense layer as operand planes from the fused
        Adam's epilogue, packed filters and small vectors with the packing launch that follows the optimiser step (None: that launch only packs for this handle's
        own next forward).  `ingest`: a callable issuing the launches that add committed transitions to the tree (a learner rank's arrived slab, device/dist.py; the
        single-GPU engine's previous lock-step); they run on a side stream and the priority write-back waits for them.
        Pre-draw (round 5): `bset` = the replay's buffer set this update trains on, `have_batch` = the PREVIOUS update drew it already (no draw at the head of this
        update's chain), `predraw` = this update draws the next one's batch into the other set right behind its priority write-back, beside the rest of its backward
        pass.  The tree sees the same sequence of operations either way: ... add, write-back(u), draw(u + 1), add, write-back(u + 1), draw(u + 2) ..."""
        cfg, r = self.cfg, self.lreplay
        B, n, A = cfg.batch_size, cfg.multisteps, cfg.n_actions
        pe = getattr(self, "_phase_mark", None)  # tools/lockstep_phases.py: timing events recorded inside the (captured) update; None in production

        def mark(i):
            if pe is not None:
                pe(i)

        def fork_ingest(cur):
            if ingest is None:
                return
            self._ev_drawn.record(cur)
            self.s_ingest.wait_event(self._ev_drawn)
            with torch.cuda.stream(self.s_ingest):
                ingest()
                self._ev_ingested.record(self.s_ingest)

        mark(0)
        if bset is not None:
            r.use_set(bset)
        step_dev = r.rng_counter if self._predraw else self.train_count_dev  # (pre-draw: the draw's own number is the update's number; srlx_per.hip:sample_wg_body)
        if self.fast:
            self.inf_online.fuse_adam_planes(self._planes_ptr[publish] if publish is not None else None)
        if self.mfma_train:
            b = r.batch if have_batch else r.sample_items(step_dev, all_states=True)
            mark(1)
            cur = torch.cuda.current_stream(self.dev)
            # Where the ingest is enqueued decides which hardware queue gets its first kernel when (same-box A/B, profiles/r5_ab_ingest_order.txt): a learner rank's
            # ingest (ring commit + 7168-leaf add: 150 us, the write-back waits for it) goes FIRST -- period alone 0.326 ms against 0.427 behind the target fork and
            # 0.467 behind both passes' launches; the single-GPU engine's (one 16 us add) goes behind the online PASS ITSELF (dependent on it): 0.432 ms per lock-step
            # against 0.454 first, 0.444 behind the target fork, 0.493 behind the launches but independent of them
            # (a rank that acts AND ingests other ranks' slabs -- the 2-GPU topology's rank 0 -- keeps the single-GPU placement: 0.429 against 0.455 ms per lock-step
            # with the ingest first, tools/dist_one_rank_probe.py)
            early = self.learner_replay is not None and self.role == "learner"
            if early:
                fork_ingest(cur)
            self._ev_t0.record(cur)
            self.s_target.wait_event(self._ev_t0)
            on_target = (not early) and ingest is not None  # (... and on the target pass's own stream, behind that pass: -0.5 % against a stream of its own behind the online pass)
            with torch.cuda.stream(self.s_target):  # fork: target network (rainbow.py:221) alongside the online network
                q_tg_next = self.inf_target.forward_u8(r.obs_base, r.frame_off_next.view(B * n, cfg.window_length))
                self._ev_t1.record(self.s_target)
                if on_target:
                    ingest()
                    self._ev_ingested.record(self.s_target)
            q_all = self.inf_online.forward_u8(r.obs_base, r.frame_off_all.view(B * (n + 1), cfg.window_length))
            if not early and not on_target:
                fork_ingest(cur)
            if self.noisy:
                # the reference evaluates q_online(s_1..s_n) (rainbow.py:220) and q_online(s_0) (model_torch.py:103) in two forward
                # calls, i.e. under two noise draws: re-evaluate the dense layers of the s_0 rows under a fresh one
                self.inf_online.redraw_rows(B, n + 1, out=q_all)
            q_all = q_all.view(B, n + 1, A)
            mark(2)
            cur.wait_event(self._ev_t1)  # join before the TD kernel
            # rainbow.py:220 + model_torch.py:103: the TD arithmetic reads s_0 and s_1..s_n rows straight out of the one forward;
            # model_torch.py:107-109 without autograd: every p.grad is (over)written by the backward kernels
            if self.fast and self._update_side:
                self.inf_online.set_priority_sink(r, b.indices, self.priorities)
                # a learner rank's write-back follows the slab's ingest on the ingest's own stream: on the weight-gradient branch it would hold that branch back
                # until the 150 us ingest is through
                sink_on_ingest = early and ingest is not None
                self.inf_online.set_sink_stream(self.s_ingest if sink_on_ingest else None)
                self.inf_online.set_sink_wait(self._ev_ingested if ingest is not None and not sink_on_ingest else None)
                self.inf_online.set_sink_done(self._ev_sunk if predraw or sink_on_ingest else None)
            if self._fused_td:  # ... in the prologue of the backward's first kernel
                self.inf_online.backward_td_u8(r.obs_base, r.frame_off_all, n, q_all, q_tg_next, b.actions, b.rewards, b.terminated, b.weights, cfg.discount,
                                               cfg.retrace_h, cfg.enable_double_dqn, cfg.enable_rescale, self.target, self.loss, self.grad_q0, self.priorities)
            else:
                N.check(
                    self.lib.srlx_nstep_td_huber_priority_packed(
                        B, n, A, N.tptr(q_all), N.tptr(q_tg_next), N.tptr(b.actions), N.tptr(b.rewards), N.tptr(b.terminated), None,
                        N.tptr(b.weights), float(cfg.discount), float(cfg.retrace_h), int(cfg.enable_double_dqn), int(cfg.enable_rescale),
                        N.tptr(self.target), N.tptr(self.loss), N.tptr(self.grad_q0), N.tptr(self.priorities), N.torch_stream_ptr(),
                    )
                )
                self.inf_online.backward_u8(r.obs_base, r.frame_off_all, self.grad_q0, sample_stride=n + 1)
            if predraw:  # the NEXT update's batch: behind this update's write-back (and the add it waited for), beside the rest of this backward pass
                self.s_predraw.wait_event(self._ev_sunk)
                with torch.cuda.stream(self.s_predraw):
                    r.use_set(1 - bset)
                    r.sample_items(step_dev, all_states=True)
                    r.use_set(bset)
                    self._ev_predrawn.record(self.s_predraw)
            mark(4)
            self.optimizer.step(self.train_count_dev)
            mark(5)
            if ingest is not None:
                cur.wait_event(self._ev_ingested)  # (the side stream joins: a capture must see it come back; a write-back on this stream must follow the add)
            if predraw:
                cur.wait_event(self._ev_predrawn)
            elif self.fast and self._update_side and early and ingest is not None:
                cur.wait_event(self._ev_sunk)  # (the ingest stream carries the write-back behind the add: it joins here)
            if self.fast:  # the new weights' packed filters: for the next online forward and, with `publish`, for the actors (+ the small vectors); train_count_dev += 1
                self.inf_online.publish_to(self.inf_actor if publish is not None else None, publish or 0,
                                           bump=self.train_count_dev if self._update_side else None)
                mark(6)
                if self._update_side:
                    return
        else:  # SRLX_TORCH_BACKWARD=1: the test yardstick -- matrix-core evaluation of s_1..s_n, autograd for the gradient step
            b = r.sample_items(self.train_count_dev)
            cur = torch.cuda.current_stream(self.dev)
            fork_ingest(cur)
            foff = r.frame_off_next.view(B * n, cfg.window_length)
            q_on_next = self.inf_online.forward_u8(r.obs_base, foff)  # rainbow.py:220
            q_tg_next = self.inf_target.forward_u8(r.obs_base, foff)  # rainbow.py:221
            q0 = self.q_online(r.obs0.view(B, *self._img))  # model_torch.py:103 (autograd)
            N.check(
                self.lib.srlx_nstep_td_huber_priority(
                    B, n, A, N.tptr(q_on_next), N.tptr(q_tg_next), N.tptr(q0), N.tptr(b.actions), N.tptr(b.rewards), N.tptr(b.terminated),
                    None, N.tptr(b.weights), float(cfg.discount), float(cfg.retrace_h), int(cfg.enable_double_dqn), int(cfg.enable_rescale),
                    N.tptr(self.target), N.tptr(self.loss), N.tptr(self.grad_q0), N.tptr(self.priorities), N.torch_stream_ptr(),
                )
            )
            self.optimizer.zero_grad(set_to_none=False)
            q0.backward(self.grad_q0)  # model_torch.py:107-109: d loss / d q seeds autograd
            self.optimizer.step()
            if ingest is not None:
                cur.wait_event(self._ev_ingested)
        r.update(b.indices, self.priorities)  # model_torch.py:113-114; train_count_dev += 1 in the same launch (count_updates_in)

    def learner_step(self, publish: Optional[int] = None) -> bool:
        """Returns False while the replay is below warm-up (priority_replay_buffer.py:228-230).  A pending `self.ingest` rides on this update."""
        r = self.lreplay
        if r.is_warmup_needed():
            return False
        if self.fast:
            self._check_versions()  # (a state dict loaded behind the engine's back: re-pack before anything trains on stale filters)
        ing, self.ingest = self.ingest, None
        ing_key, ing_fn = (ing[0], ing[1]) if ing is not None else (None, None)
        bset, have, pre = None, False, False
        if self._predraw:
            bset, pre = self._bset, True
            # the set holds a batch the previous update drew -- unless a ring commit has run on the device since that draw (updates paused while slabs kept
            # arriving): this update's own ingest is then the SECOND commit between the draw and its frame reads, and the margin is one slot
            have = self._drawn == bset and self._device_commits() == self._drawn_at
            if self._drawn is not None and not have:
                r.rng_counter.sub_(1)  # the stale draw is dropped: this update draws under the same number (the draw's number is the update's number)
        key = (publish, ing_key, bset, have)
        g = self._learner_graphs.get(key)
        if g is None and self._capturing and not self._in_capture:  # a combination first seen after `capture_graphs`: captured now, replayed from then on
            torch.cuda.current_stream(self.dev).synchronize()
            g = self._capture_learner(key, ing_fn, pre)
        if g is not None:
            g.replay()
        else:
            self._learner_body(publish, ing_fn, bset, have, pre)
        if ing is not None:
            self._note_issued_commit()
        if self._predraw:
            r.use_set(bset)  # (host view: `batch`, `used`, ... name what THIS update trained on)
            self._drawn, self._drawn_at, self._bset = 1 - bset, self._device_commits(), 1 - bset  # (the pre-draw ran behind this update's own ingest)
        if self.fast:
            self._fresh_set = publish  # the planes of that set now hold the online weight (None: no set does)
        # model_torch.py:117-119 (fires at train_count 0 too)
        if self.train_count % self.cfg.target_model_update_interval == 0:
            self.sync_target()
        self.train_count += 1
        return True

    def _device_commits(self) -> int:
        """Ring commits the device has been handed: the host's count plus the ingests issued since it last moved (a slab's `note_commit` follows the updates)."""
        r = self.lreplay
        if r._steps_committed != self._seen_commits:
            self._seen_commits, self._unnoted = r._steps_committed, 0
        return r._steps_committed + self._unnoted

    def _note_issued_commit(self):
        self._device_commits()
        self._unnoted += 1

    def _capture_learner(self, key, ingest_fn, predraw: bool = False):
        g = torch.cuda.CUDAGraph()
        self._in_capture = True
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):  # other threads (the RCCL watchdog) may touch the runtime meanwhile
                self._learner_body(key[0], ingest_fn, key[2] if len(key) > 2 else None, key[3] if len(key) > 3 else False, predraw)
        finally:
            self._in_capture = False
        self._learner_graphs[key] = g
        return g

    def sync_target(self):
        with torch.no_grad():
            torch._foreach_copy_(list(self.q_target.parameters()), list(self.q_online.parameters()))
        if self.fast:  # the target handle keeps its packed filters between syncs: re-pack them now (current stream: the learner's)
            self.inf_target.weights_changed()
            self.inf_target.publish_to(None)
        self.sync_count += 1

    # ---- the pieces of a step (the Runner's vectorised loop drives them one by one: device/vector_runner.py) --------
    def fork_point(self):
        """Marks the point of the current stream the next fork_learner(..., marked=True) is ordered after (the replay as of now): lets the host enqueue more work on
        the current stream -- the actors' pass -- BEFORE it spends ~100 us inside the update graph's launch, without that work becoming a dependency of the update."""
        self._ev_fork.record(torch.cuda.current_stream(self.dev))

    def fork_learner(self, updates: int, marked: bool = False) -> int:
        """overlap=True: enqueue `updates` learner updates on the learner's stream, ordered after everything enqueued on the
        current stream so far (they see the replay as of now; marked=True: as of the last fork_point()).  Returns how many ran (0 below the warm-up)."""
        if self.fast:
            self._check_versions()
        if not marked:
            self.fork_point()
        if self.ingest is None and self.replay.lagged:  # the tree add of the previous lock-step rides on this fork's first update (or runs alone below)
            self.ingest = self.replay.take_pending_add()
        self.s_learner.wait_event(self._ev_fork)
        ran = 0
        with torch.cuda.stream(self.s_learner):
            for k in range(updates):
                if self.fast:
                    pub = 1 - self._set if k == updates - 1 else None  # the last update of the lock-step publishes into the set the actors are NOT reading
                    ok = self.learner_step(pub)
                    if ok and pub is not None:
                        self._published = pub
                else:
                    ok = self.learner_step()
                ran += int(ok)
            if self.ingest is not None:  # no update took the pending ingest with it (warm-up, or none asked for): commit it here, in stream order
                ing, self.ingest = self.ingest, None
                ing[1]()
                self._note_issued_commit()
            self._ev_join.record(self.s_learner)
        self._learner_pending = True
        return ran

    def run_updates(self, updates: int) -> int:
        """`updates` learner updates NOT beside this engine's actors (a learner-only rank; a learner rank without overlap): on the learner's launch stream, ordered
        after the current stream and joined back to it.  A pending `ingest` rides on the first update or runs by itself."""
        cur = torch.cuda.current_stream(self.dev)
        self.s_learner.wait_stream(cur)
        ran = 0
        with torch.cuda.stream(self.s_learner):
            for _ in range(updates):
                ran += int(self.learner_step())
            if self.ingest is not None:
                ing, self.ingest = self.ingest, None
                ing[1]()
                self._note_issued_commit()
        cur.wait_stream(self.s_learner)
        return ran

    def join_learner(self):
        """The current stream waits for the forked updates (before the next write to the replay)."""
        if self._learner_pending:
            torch.cuda.current_stream(self.dev).wait_event(self._ev_join)
            self._learner_pending = False

    def refresh_actor_copy(self):
        """overlap=True: one multi-tensor copy online -> the actor's private network."""
        if self.fast:  # the joined update wrote the other set: the next passes read it (a pointer swap on the host)
            if self._published is not None:
                self._set, self._published = self._published, None
                self.inf_actor.select_set(self._set)
        elif self.q_actor is not self.q_online:
            self.inf_actor.refresh_from(self.q_online)

    def actor_front(self, events=None):
        """Network pass + action selection + environments of one lock-step: reads the ring, writes nothing shared."""
        if self.fast:  # network pass + selection in one call (3 launches), the environments in one
            self._check_versions()
            off = self.replay.frame_table_current()  # (no launch: the last commit wrote it)
            if events is not None:
                events[0].record()
            self.inf_actor.forward_u8_policy(self.replay.obs_base, off, self.eps, self.cfg.seed ^ 0xAC7, self.policy_counter, self.actions,
                                             q_copy=self.q_hist[self._passes % self.q_hist.shape[0]] if self.actor_priority else None)
            if events is not None:
                events[1].record()
            if self.before_env is not None:  # (device/dist.py: the exchange of the previous lock-step must be over before the environments overwrite what it ships)
                self.before_env()
            self.env.step(self.actions)
            return
        q = self._actor_net(None, events)  # eager launches, bracketed by the events
        if self.before_env is not None:
            self.before_env()
        if self._select_graph is not None:
            self._select_graph.replay()
        else:
            self._actor_select(q)

    def actor_commit(self):
        """Ring commit + PER add of the lock-step `actor_front` produced: the only actor writes to the replay."""
        if self.fast:
            self.actor_commit_ring()
            self.actor_commit_tree()
            return
        if self._own_ring_only:  # the ring only stacks frames for this engine's actors
            e = self.env
            if self.ledger is not None:
                self.ledger.account(e.rewards, e.done, self.replay.needs_reset_ptr)
            self.replay.commit(self.actions, e.rewards, e.terminated, e.done, e.next_obs, defer_add=True)
        elif self._commit_graph is not None:
            self._commit_graph.replay()
            self.replay._steps_committed += 1
        else:
            self._actor_commit()
        if self.actor_priority:  # what the deferred add of this lock-step will need
            r = self.replay
            self._ap["mask"].copy_(r.item_mask)
            self._ap_first_slot = ((r._steps_committed - 1) * self.cfg.n_envs) % r.capacity
            self._passes += 1
            if not self._own_ring_only:
                # the commit has just replaced the items these E leaves stood for, and their successors are added one lock-step from now: until then the leaves
                # hold NOTHING (priority 0: never drawn) instead of the old items' priorities on the new items' data
                if not hasattr(self, "_zero_leaves"):
                    self._zero_leaves = torch.zeros(self.cfg.n_envs, dtype=torch.float64, device=self.dev)
                N.check(self.lib.srlx_per_set_range(r.h_per, self._ap_first_slot, self.cfg.n_envs, N.tptr(self._zero_leaves), N.PRIO_RAW, 1, N.torch_stream_ptr()))
        self.total_env_steps += self.cfg.n_envs

    def step(self, learner_updates: int = 1, events=None):
        """One engine step: E environment steps and `learner_updates` Rainbow updates.  `events` = (start, end)
        torch events recorded around the dominant hand-written kernel group of the actor on its launch
        stream: the matrix-core network pass (or, on the torch path, the frame-stack kernel)."""
        if self.fast:
            self.fork_learner(learner_updates)
            self.actor_front(events)
            self.actor_commit_ring()  # before the join: nothing the learner reads
            self.join_learner()
            self.actor_commit_tree()
            self.refresh_actor_copy()
            return
        if self.overlap:  # the learner sees the replay as of the end of the previous step
            self.fork_learner(learner_updates)
        self.actor_front(events)
        if self.overlap:
            self.join_learner()
        if self.actor_priority:
            self._add_with_actor_priorities()
        self.actor_commit()
        if self.overlap:
            self.refresh_actor_copy()
        else:
            for _ in range(learner_updates):
                self.learner_step()

    # ---- HIP graphs -------------------------------------------------------------------------
    def capture_graphs(self, actor: bool = True, learner: bool = True, warm_actor: bool = True, warm_learner: bool = True):
        """Captures the actor step and the learner step into HIP graphs (launch-bound inner loops).
        Call after warm-up: arenas are sized and the replay is past its warm-up gate.  `warm_actor=False` skips the
        extra eager actor step (a distributed wrapper has already stepped, and an un-pushed step would desynchronise
        the learner's global ring from this rank's environments).  Update variants that were not captured here (a learner rank's ingest keys) are captured the
        first time they run."""
        torch.cuda.synchronize(self.dev)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            if actor and warm_actor and self.role != "learner":
                self.actor_step()
                self.total_env_steps += self.cfg.n_envs
            if learner and warm_learner and self.role != "actor" and not self.lreplay.is_warmup_needed() and self.ingest is None:
                self.learner_step()  # a real update (eager: sizes the arenas), target sync and counters included
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        if learner and self.role != "actor":
            self._capturing = True
        if self.fast:  # the actors' launches stay eager; the update is captured per variant: publishing into set 0 / set 1 / not at all
            if learner and self.role != "actor" and not self.lreplay.is_warmup_needed() and self.learner_replay is None and not self.replay.lagged:
                for key in ([(None, None, None, False), (0, None, None, False), (1, None, None, False)] if self.role == "both" else [(None, None, None, False)]):
                    self._capture_learner(key, None)
                self._learner_graph = self._learner_graphs[(None, None, None, False)]
            torch.cuda.synchronize(self.dev)
            return
        if actor and self.role != "learner" and not self._own_ring_only:
            q = self._actor_net(None)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):  # other threads (the RCCL watchdog) may touch the runtime meanwhile
                self._actor_select(q)
            self._select_graph = g
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):  # other threads (the RCCL watchdog) may touch the runtime meanwhile
                self._actor_commit()
            self.replay._steps_committed -= 1  # capture does not execute
            self._commit_graph = g
        if learner and self.role != "actor" and not self.lreplay.is_warmup_needed() and self.learner_replay is None:
            self._learner_graph = self._capture_learner((None, None, None, False), None)
        torch.cuda.synchronize(self.dev)

    def refresh_host_mirrors(self):
        self.join_learner()
        self.replay.flush_pending_add()
        N.check(self.lib.srlx_per_refresh(self.replay.h_per, N.torch_stream_ptr()))

    def info(self):
        self.join_learner()
        self.replay.flush_pending_add()
        self.lreplay.check_draws()
        check_ranges()
        return dict(loss=float(self.loss.item()), train_count=self.train_count, sync=self.sync_count, memory=self.lreplay.length())
