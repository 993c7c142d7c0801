"""Agent57_light on one GPU with every network pass, every optimiser step and every per-lane / per-batch array operation in libsrlx (round 6).

BASELINE.json configs[3] workload; replaces for E lock-stepped environments + one learner (reference paths under the reference root):

    srl/algorithms/agent57_light/agent57_light.py:271-471   Worker.on_reset / policy / on_step (UVFA inputs, UCB arm, intrinsic reward, item fields)
    srl/algorithms/agent57_light/agent57_light.py:165-268   change_batches_format, calc_target_q
    srl/algorithms/agent57_light/model_torch.py:18-117      QNetwork (UVFA), _EmbeddingNetwork, _LifelongNetwork
    srl/algorithms/agent57_light/model_torch.py:244-443     Trainer: four Adams, _update_q x 2, embedding loss, RND loss, mixed priorities, target sync

`device/agent57_light.py` (rounds 2-5) evaluates the image trunks in libsrlx and everything behind them in torch (hipBLASLt GEMMs, ATen elementwise / indexing
launches, four torch.optim.Adam): ~150 launches per lock-step on the actors' side, a ~400-node update graph, 62 % of the kernel time in library kernels.  Here:

    the two UVFA Q-networks      srlx_qnet handles with UVFA columns (srlx_qnet_bind_uvfa): conv trunk -> first dense layer on the bf16 matrix pipe (exact split
                                 products) -> head kernel that adds the rank-1 UVFA terms; no concatenated input is ever built.  Update = Rainbow's, n_step = 1:
                                 ONE pass over [s_0, s_1] rows, target network beside it, TD / Huber / gradient seed in the backward's head kernel with the sampled
                                 actor's discount (srlx_qnet_set_td_extras), Adam fused into the launches that finish each gradient.
    embedding / RND networks     srlx_qnet handles in head_mode 1 (trunk + one dense layer); the embedding network's classifier tail and the RND LayerNorm + MSE:
                                 one single-workgroup launch each, forward + backward + Adam (srlx_agent57_emb_tail / _rnd_tail).
    actors' bookkeeping          srlx_agent57_policy / _post_step / _begin_episodes (one launch each) + the UCB bank + the NGU kNN kernels of rounds 2-3.
    batch assembly               srlx_per_sample_gather_train + srlx_store_locate + srlx_agent57_gather_inputs.

Actors and learner overlap like the Rainbow engine's fast lock-step (device/rainbow.py): the update is ONE captured HIP graph on the learner's stream, the actors
read one of two PUBLISHED parameter sets per network (packed filters, first dense layer as bf16 operand planes, small vectors, UVFA columns) that the update writes
-- the first dense layers from their fused Adam epilogues -- and flip after the join.  Needs E >= 512 in multiples of 128; smaller engines (tests) run the update
behind the actors on one stream and read the master parameters.

The plugin's torch modules (algorithms/agent57_light.py: the reference's state_dict keys) are the import / export format: `load_parameter` / `export_parameter`.
There is no torch network on the path and no fallback: unsupported shapes raise.
"""
import ctypes
from typing import Optional

import numpy as np
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.algorithms._device_ops import NguOps
from simple_distributed_rl_amd.device.agent57_light import UcbBank
from simple_distributed_rl_amd.device.qnet import DeviceAdam, EngineHiddenNet, EngineQNet, QNetInference, check_ranges
from simple_distributed_rl_amd.device.replay import DeviceReplay
from simple_distributed_rl_amd.rl import functions as funcs

_EMB_TAIL_KEYS = ("out_block.hidden_layers.0.weight", "out_block.hidden_layers.0.bias", "out_block_normalize.weight", "out_block_normalize.bias",
                  "out_block_out1.weight", "out_block_out1.bias")
_RND_TAIL_KEYS = ("hidden_normalize.weight", "hidden_normalize.bias")


def _single(block_cfg, what):
    sizes = tuple(block_cfg.kwargs.get("layer_sizes", ()))
    if len(sizes) != 1 or block_cfg.kwargs.get("activation", "relu").lower() != "relu":
        raise ValueError(f"Agent57LightFastEngine: {what} must be ONE ReLU layer (got layer_sizes={sizes}); there is no fallback network path")
    return int(sizes[0])


def why_not_fast(rl_config) -> str:
    """Empty string when `Agent57LightFastEngine` covers this (set-up) config; otherwise the reason (the Runner then takes the round-5 engine with torch tails,
    device/agent57_light.py, which serves every DQN-image / dueling shape the plugin builds)."""
    c = rl_config
    shape = tuple(int(x) for x in c.observation_space.shape)
    if len(shape) != 3 or shape != (84, 84, 4):
        return "observations are not 84 x 84 x 4 frame stacks"
    if c.input_block.image.name != "DQN" or c.input_block.image.kwargs.get("filters", 32) != 32:
        return "input block is not the DQN image block with 32 filters"
    dk = c.hidden_block.kwargs
    sizes = tuple(dk.get("layer_sizes", ()))
    if c.hidden_block.name != "DuelingNetwork" or len(sizes) != 1 or sizes[0] % 64 or sizes[0] > 512:
        return "hidden block is not one dueling layer of a multiple of 64 (<= 512) units"
    if dk.get("dueling_kwargs", {}).get("dueling_type", "average") not in ("average", ""):
        return "dueling type is neither 'average' nor ''"
    if c.batch_size > 32:
        return "batch size above 32"
    if c.enable_intrinsic_reward:
        for blk in (c.episodic_emb_block, c.episodic_out_block, c.lifelong_hidden_block):
            if len(tuple(blk.kwargs.get("layer_sizes", ()))) != 1 or str(blk.kwargs.get("activation", "relu")).lower() != "relu":
                return "an embedding / lifelong block is not ONE ReLU layer"
        D, Hd, A = int(c.episodic_emb_block.kwargs["layer_sizes"][0]), int(c.episodic_out_block.kwargs["layer_sizes"][0]), int(c.action_space.n)
        if D > 128 or c.batch_size * (2 * D + 3 * Hd + A + 1) + 256 > 16384:
            return "the embedding network's tail does not fit one workgroup's 64 KB of LDS (batch x (2 x embedding + 3 x classifier layer))"
        if int(c.lifelong_hidden_block.kwargs["layer_sizes"][0]) != 128:
            return "the lifelong networks' layer is not 128 units wide (its LayerNorm is fused into a 128-unit tile)"
    return ""


class _Net:
    """One trained network: master module (kernel layouts), learner handle, optimiser, the actors' handle; for a Q-network also its target."""

    def __init__(self, name):
        self.name = name
        self.module = self.inf = self.opt = self.actor = self.target = self.inf_target = None
        self.planes_ptr = None


class Agent57LightFastEngine:
    def __init__(self, rl_config, n_envs: int, device: int = 0, episode_len: int = 200, seed: int = 0, env=None, parameter=None, ring_len: Optional[int] = None,
                 overlap: Optional[bool] = None, fc1_neighbour: int = 3, fused_adam: bool = True, role: str = "both", learner_replay: Optional[DeviceReplay] = None,
                 actor_stream: Optional[str] = None):
        """rl_config: a set-up algorithms.agent57_light.Config (84 x 84 x window-4 image observations); parameter: its Parameter (the five torch networks: the
        initial weights are taken from it, `export_parameter()` writes the trained ones back).  overlap (None = wherever it applies): the update beside the
        actors on published parameter sets; needs n_envs >= 512 in multiples of 128.  fused_adam=False (tests): every gradient is written to `p.grad` and the
        optimiser steps are launches of their own (srlx_adam_step) -- the same arithmetic, with the gradients left to look at.
        role (device/dist.py): "both" = actors and learner on this GPU; "actor" = a rank that only acts (no target networks, no optimisers; its passes read the
        master parameters a weight broadcast lands in, through sets published out of band: `on_weights_broadcast`); "learner" = a rank that only learns.
        learner_replay: the replay the LEARNER samples when it is not this engine's own ring (a learner rank's global replay over every actor rank's environments);
        the engine's own ring then only stacks frames for its actors."""
        from simple_distributed_rl_amd.device.rainbow import SyntheticAtariVecEnv

        c = self.cfg = rl_config
        assert c.is_setup(), "rl_config.setup(env) first: the networks are built from the negotiated spaces"
        assert role in ("both", "actor", "learner")
        self.role, self.acts, self.learns = role, role != "learner", role != "actor"
        self.learner_replay = learner_replay
        self.dev = torch.device(f"cuda:{device}")
        self.device_index = int(device)
        self.lib = N.lib()
        self.E, self.seed = int(n_envs), int(seed)
        shape = c.observation_space.shape
        H, W_, Wn = int(shape[0]), int(shape[1]), int(shape[2])
        self.hw, self.Wn, self.A = (H, W_), Wn, int(c.action_space.n)
        E, B, A = self.E, int(c.batch_size), self.A
        dk = c.hidden_block.kwargs
        ok = (c.hidden_block.name == "DuelingNetwork" and len(tuple(dk.get("layer_sizes", ()))) == 1 and dk.get("dueling_kwargs", {}).get("dueling_type", "average") in ("average", "")
              and (H, W_, Wn) == (84, 84, 4) and c.input_block.image.kwargs.get("filters", 32) == 32 and B <= 32)
        if not ok:
            raise ValueError("Agent57LightFastEngine covers 84 x 84 x 4 frames, the DQN image block with 32 filters, ONE dueling layer (average / none) and batches "
                             "<= 32; there is no fallback network path (the round-5 engine with torch tails is device/agent57_light.py, a test yardstick)")
        why = why_not_fast(c)
        if why:
            raise ValueError(f"Agent57LightFastEngine: {why}; there is no fallback network path")
        self.hidden = int(tuple(dk["layer_sizes"])[0])
        self.dueling = dk.get("dueling_kwargs", {}).get("dueling_type", "average")
        self.intrinsic = bool(c.enable_intrinsic_reward)
        self.fused_adam = bool(fused_adam)
        chip_filling = E >= 512 and E % 128 == 0
        can_overlap = chip_filling and self.fused_adam and role == "both"
        if overlap and not can_overlap:
            raise ValueError("Agent57LightFastEngine(overlap=True): needs >= 512 environments in multiples of 128 (the published sets feed the chip-filling kernels)")
        self.overlap = can_overlap if overlap is None else bool(overlap)
        self.sets = self.overlap or (role == "actor" and chip_filling)  # an actor rank's passes read sets published out of band after every weight broadcast
        self._own_ring_only = role == "actor" or learner_replay is not None  # the engine's own ring only stacks frames: no tree add, the commit moves the position
        mem = c.memory
        kw = mem.kwargs if mem.name != "ReplayBuffer" else {}
        if ring_len is None:
            ring_len = -(-mem.capacity // E) + 1 + Wn
        if self.acts:
            self.replay = DeviceReplay(E, ring_len, H * W_, Wn, 1, A, B, True, False, float(kw.get("alpha", 0.0)), float(kw.get("beta_initial", 0.4)),
                                       int(kw.get("beta_steps", 1_000_000)), float(kw.get("epsilon", 1e-4)), mem.warmup_size, self.seed, device)
            self.env = SyntheticAtariVecEnv(self.replay, episode_len) if env is None else (env(self.replay) if callable(env) else env)
        else:
            assert learner_replay is not None, "role='learner' needs the replay it learns from"
            self.replay, self.env = learner_replay, None
        self.L = self.replay.L
        c._set_device(str(self.dev))
        if parameter is None:
            parameter = c.make_parameter()
        self.parameter = parameter
        d = self.dev
        Na = int(c.actor_num)
        # ---- UVFA column layout (model_torch.py:52-62: ext reward, int reward, one-hot action, one-hot actor) ----
        col = 0
        c_ext = c_int = c_act = -1
        if c.input_ext_reward:
            c_ext, col = col, col + 1
        if c.input_int_reward and self.intrinsic:
            c_int, col = col, col + 1
        if c.input_action:
            c_act, col = col, col + A
        c_actor, col = col, col + Na
        self.uvfa_layout = (c_ext, c_int, c_act, A if c_act >= 0 else 0, c_actor, Na)
        self.X = col
        self.D_emb = _single(c.episodic_emb_block, "episodic_emb_block") if self.intrinsic else 0
        self.H_emb = _single(c.episodic_out_block, "episodic_out_block") if self.intrinsic else 0
        self.D_rnd = _single(c.lifelong_hidden_block, "lifelong_hidden_block") if self.intrinsic else 0
        self.train_count_dev = torch.zeros(1, dtype=torch.int64, device=d)
        self.beta_list = torch.tensor(np.array(funcs.create_beta_list(Na), np.float32), device=d)
        self.discount_list = torch.tensor(np.array(funcs.create_discount_list(Na), np.float32), device=d)
        self.eps_list = torch.tensor(np.array(funcs.create_epsilon_list(Na), np.float32), device=d)
        # the actors' side on a stream of its own priority level: HIP keeps one pool of hardware queues per level and runs a graph's internal branches on
        # normal-priority streams -- on the caller's (normal) stream the actors' chip-filling launches share a hardware queue with a branch of the update, which then
        # runs BEHIND them instead of beside them (tools/a57_trace.py: no overlap at all).  Default: "low" for an overlapped engine; `close()` hands the thread back
        self.actor_stream = None
        want = actor_stream if actor_stream is not None else ("low" if (self.overlap and role == "both" and learner_replay is None) else "default")
        if want != "default":
            raw = ctypes.c_void_p()
            N.check(self.lib.srlx_stream_create({"high": -1, "normal": 0, "low": 1}[want], ctypes.byref(raw)))
            self._actor_stream_raw = raw
            self._stream_before = torch.cuda.current_stream(self.dev)
            self.actor_stream = torch.cuda.ExternalStream(raw.value, device=self.dev)
            self.actor_stream.wait_stream(self._stream_before)
            torch.cuda.set_stream(self.actor_stream)
        self._build_networks(B, fc1_neighbour)
        self.load_parameter(parameter)
        z = lambda dt, *sh: torch.zeros(sh, dtype=dt, device=d)  # noqa: E731
        self._graphs, self._capturing, self._in_capture = {}, False, False
        self._set, self._published = 0, None
        self._learner_pending = False
        self.train_count = self.sync_count = self.total_env_steps = 0
        self.ledger, self.training, self.ingest, self.before_env = None, True, None, None
        self._trunks_fresh, self._q_ready = False, False
        # the five image blocks of a lock-step as ONE launch (srlx_qnet_forward_convs_multi_u8): 3 % faster for the actors alone and -4 % per lock-step beside the
        # update (1.41 against 1.47 ms, three interleaved repetitions) since the update's forward passes left its critical chain; beside the round's earlier
        # 1.1 ms update chain the single launch LOST 3-4 % (it leaves the update's small kernels no launch boundary to slot into)
        self.multi_trunk = True
        if self.learns:
            self._init_learner(B, A, z)
        if self.acts:
            self._init_actors(E, A, Na, z)
            self.first_obs = self.env.reset()
            self.replay.reset_all(self.first_obs)
        self._publish_out_of_band()
        if self.acts:
            self._begin_all()

    def _init_actors(self, E, A, Na, z):
        """Per-environment actor state (the reference keeps these on the worker object, :288-311)."""
        c, d = self.cfg, self.dev
        self.ucb = UcbBank(E, Na, c.ucb_window_size, c.ucb_epsilon, c.ucb_beta, d, self.seed)
        self.episode_reward, self.prev_r_ext, self.prev_r_int = z(torch.float32, E), z(torch.float32, E), z(torch.float32, E)
        self.prev_action, self.actions, self.zero_arm = z(torch.int32, E), z(torch.int32, E), z(torch.int32, E)
        self.reset_lane, self.live_lane = z(torch.uint8, E), torch.ones(E, dtype=torch.uint8, device=d)
        self.policy_counter = z(torch.int64, 1)
        self.q_ext, self.q_int, self.q = z(torch.float32, E, A), z(torch.float32, E, A), z(torch.float32, E, A)
        L = self.L
        self.x_r_int, self.x_prev_r_ext, self.x_prev_r_int = z(torch.float32, L, E), z(torch.float32, L, E), z(torch.float32, L, E)
        self.x_actor, self.x_prev_action = z(torch.int32, L, E), z(torch.int32, L, E)
        self.ngu = None
        if self.intrinsic:
            self.ngu = NguOps(d, E, self.D_emb, c.episodic_memory_capacity, c.episodic_count_max, c.episodic_epsilon, c.episodic_cluster_distance, c.episodic_pseudo_counts)
            self.emb_out, self.rnd_t_out, self.rnd_p_out = z(torch.float32, E, self.D_emb), z(torch.float32, E, self.D_rnd), z(torch.float32, E, self.D_rnd)
            self.episodic, self.lifelong = z(torch.float32, E), z(torch.float32, E)
        self.record = z(torch.uint8, (10 + 4 * 5) * E)  # this lock-step as one packed record (srlx_agent57_pack_record: what an actor rank ships)

    @property
    def lreplay(self) -> DeviceReplay:
        """The replay the learner samples and writes back to."""
        return self.learner_replay if self.learner_replay is not None else self.replay

    def _init_learner(self, B, A, z):
        d = self.dev
        if self.learner_replay is not None:  # the item fields of the global replay's environments, [ring slot][environment] (filled by `ingest_fields`)
            Lg, Eg = self.learner_replay.L, self.learner_replay.E
            self.lx = dict(r_int=z(torch.float32, Lg, Eg), prev_r_ext=z(torch.float32, Lg, Eg), prev_r_int=z(torch.float32, Lg, Eg), actor=z(torch.int32, Lg, Eg),
                           prev_action=z(torch.int32, Lg, Eg))
        self.loc_env, self.loc_slot = z(torch.int64, B), z(torch.int64, B)
        self.on_r_ext, self.on_r_int, self.on_action, self.on_actor = z(torch.float32, 2 * B), z(torch.float32, 2 * B), z(torch.int32, 2 * B), z(torch.int32, 2 * B)
        self.tg_r_ext, self.tg_r_int, self.tg_action, self.tg_actor = z(torch.float32, B), z(torch.float32, B), z(torch.int32, B), z(torch.int32, B)
        self.b_discount, self.b_r_int = z(torch.float32, B), z(torch.float32, B)
        self.out = {k: dict(q_all=z(torch.float32, 2 * B, A), q_tg=z(torch.float32, B, A), target=z(torch.float32, B), loss=z(torch.float32, 1), grad_q0=z(torch.float32, B, A),
                            pri=z(torch.float32, B), td=z(torch.float32, B)) for k in ("q_ext", "q_int")}
        self.priorities = z(torch.float32, B)
        if self.intrinsic:
            self.l_emb, self.g_emb, self.emb_loss = z(torch.float32, 2 * B, self.D_emb), z(torch.float32, 2 * B, self.D_emb), z(torch.float32, 1)
            self.l_rnd_p, self.l_rnd_t, self.g_rnd, self.rnd_loss = z(torch.float32, 2 * B, self.D_rnd), z(torch.float32, 2 * B, self.D_rnd), z(torch.float32, B, self.D_rnd), z(torch.float32, 1)
        self.s_target = torch.cuda.Stream(device=d, priority=-1)
        self._ev_fwd = {k: torch.cuda.Event() for k in ("q_ext", "q_int", "emb", "rnd")}  # a network's forward passes (on s_target) are through
        self._ev_fork_fwd = torch.cuda.Event()
        self.hoist_forwards = True  # every network's forward on the side stream, beside the previous network's backward (False: one network after the other)
        self.s_learner = torch.cuda.Stream(device=d, priority=-1)
        self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
        # a learner rank's ingest (device/dist.py): the commit of a slab that arrived from the actor ranks runs on a side stream between the update's draw and its
        # priority write-back -- `self.ingest` = (key, callable issuing the launches) for the NEXT update only
        self.s_ingest = torch.cuda.Stream(device=d, priority=-1) if self.learner_replay is not None else None  # (no stream the single-GPU engine does not use: one
        # more stream in the process changes which hardware queues the others land on)
        self._ev_drawn, self._ev_ingested = torch.cuda.Event(), torch.cuda.Event()
        if self.overlap and not self._own_ring_only:
            self.replay.enable_deferred_advance()

    # ---- networks ----------------------------------------------------------------------------------------------------------------------------------------------
    def _build_networks(self, B, fc1_neighbour):
        c, d, dev, E = self.cfg, self.dev, self.device_index, self.E
        self.nets = {}
        lrs = dict(q_ext=c.lr_ext, q_int=c.lr_int, emb=c.episodic_lr, rnd=c.lifelong_lr)

        def actor_handle(module, uv=None):
            if not self.acts:
                return None
            h = QNetInference(module, E, dev, uvfa_layout=uv)
            if self.sets:
                h.enable_fc1_planes(private_weights=True)
                h.enable_actor_sets()
                # beside an update: half-CU workgroups (an actor rank has the GPU to itself: CU-filling ones); a 128-unit layer is ONE column tile: 32 K splits make
                # 256 short workgroups of it (4 would be E / 128 x 4 serial chains of 60 K-slabs: the latency of the 1024-unit layer for an eighth of its work)
                h.set_fc1_neighbour((fc1_neighbour if self.role == "both" else 0) if uv is not None else 32)
            return h

        def trainable(net, max_train, max_rows, uv=None):
            net.inf = QNetInference(net.module, max_rows, dev, uvfa_layout=uv)  # (an actor rank: only packs / publishes what a broadcast brought)
            if self.learns:
                net.inf.enable_training(max_train)
                net.opt = DeviceAdam(net.inf._params(), lr=lrs[net.name])
                if self.fused_adam:
                    net.opt.fuse_first_dense(net.inf, self.train_count_dev)
                    net.opt.fuse_rest(net.inf)
                N.check(self.lib.srlx_qnet_set_main_first(net.inf.h, 1))
            net.actor = actor_handle(net.module, uv)
            if self.sets:
                net.planes_ptr = [net.actor.set_planes_ptr(0), net.actor.set_planes_ptr(1)]

        for name in ("q_ext", "q_int"):
            n = self.nets[name] = _Net(name)
            mk = lambda: EngineQNet(self.A, self.hw, self.Wn, self.hidden, 32, self.dueling, uvfa_cols=self.X).to(d)  # noqa: E731
            n.module = mk()
            trainable(n, B, 2 * B, self.uvfa_layout)
            if self.learns:
                n.target = mk()
                n.inf_target = QNetInference(n.target, B, dev, uvfa_layout=self.uvfa_layout)
                n.inf_target.set_pack_sticky(True)
        if not self.intrinsic:
            return
        n = self.nets["emb"] = _Net("emb")
        Hd, De, A = self.H_emb, self.D_emb, self.A
        n.module = EngineHiddenNet(De, self.hw, self.Wn, 32, tail_shapes=[(Hd, 2 * De), (Hd,), (Hd,), (Hd,), (A, Hd), (A,)]).to(d)
        trainable(n, 2 * B, 2 * B)
        n.inf.set_head_mode(1, De)
        if n.actor is not None:
            n.actor.set_head_mode(1, De)
        n.tail_g = [torch.zeros_like(t) for t in n.module.tail]
        n.tail_m = [torch.zeros_like(t) for t in n.module.tail]
        n.tail_v = [torch.zeros_like(t) for t in n.module.tail]
        n = self.nets["rnd"] = _Net("rnd")
        Dr = self.D_rnd
        n.module = EngineHiddenNet(Dr, self.hw, self.Wn, 32, tail_shapes=[(Dr,), (Dr,)]).to(d)
        n.target = EngineHiddenNet(Dr, self.hw, self.Wn, 32, tail_shapes=[(Dr,), (Dr,)]).to(d)  # the fixed random network (never trained)
        trainable(n, B, 2 * B)
        n.inf.set_head_mode(1, Dr)
        n.tail_g = [torch.zeros_like(t) for t in n.module.tail]
        n.tail_m = [torch.zeros_like(t) for t in n.module.tail]
        n.tail_v = [torch.zeros_like(t) for t in n.module.tail]
        # the actors' copies of the predictor's LayerNorm parameters, one pair per published set (the RND tail writes the pair of the set the update publishes into)
        n.ln_sets = [[torch.ones(Dr, device=d), torch.zeros(Dr, device=d)] for _ in range(2)]
        n.actor_target = None
        if self.learns:
            n.inf_target = QNetInference(n.target, 2 * B, dev)
            n.inf_target.set_pack_sticky(True)
        if self.acts:
            n.actor_target = QNetInference(n.target, E, dev)
            n.actor_target.set_pack_sticky(True)
            if self.sets:
                n.actor_target.enable_fc1_planes(private_weights=True)
                n.actor_target.set_fc1_neighbour(32)
        self._set_tail_pointers()

    def _set_tail_pointers(self):
        """Everything that holds the ADDRESS of a tail tensor (the embedding tail's pointer tables, the LayerNorm pointers of the RND target handles): again after the
        parameters were re-homed (`rebind`)."""
        if not self.intrinsic:
            return
        e, n, Dr = self.nets["emb"], self.nets["rnd"], self.D_rnd
        tab = lambda ts: (N.c_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
        e.tail_tabs = (tab(list(e.module.tail)), tab(e.tail_g), tab(e.tail_m), tab(e.tail_v))
        for h in (n.inf_target, n.actor_target):
            if h is not None:
                h.set_head_mode(1, Dr, n.target.tail[0], n.target.tail[1])

    def modules(self):
        """The five networks a weight broadcast carries (model_torch.py:148-156: both online Q-networks, the embedding network, the RND target and predictor)."""
        ms = [self.nets["q_ext"].module, self.nets["q_int"].module]
        if self.intrinsic:
            ms += [self.nets["emb"].module, self.nets["rnd"].target, self.nets["rnd"].module]
        return ms

    def rebind(self):
        """The parameters were re-homed (device/dist.py:flatten_parameters): every handle, optimiser and pointer table reads their addresses again."""
        for n in self.nets.values():
            for h in (n.inf, n.actor, n.inf_target, getattr(n, "actor_target", None)):
                if h is not None:
                    h.bind()
            if n.opt is not None:
                n.opt.bind()
        self._set_tail_pointers()
        self._publish_out_of_band()

    def on_weights_broadcast(self):
        """A broadcast has just overwritten the master parameters in place: whatever was derived from them is rebuilt."""
        self.join_learner()
        self._publish_out_of_band()

    def load_parameter(self, p):
        """The five networks := the plugin Parameter's torch modules (the reference's state_dict keys)."""
        self.join_learner()
        for name, src in (("q_ext", p.q_ext_online), ("q_int", p.q_int_online)):
            n = self.nets[name]
            n.module.load_reference_state_dict(src.state_dict())
            if n.target is not None:
                n.target.load_reference_state_dict(getattr(p, name + "_target").state_dict())
        if self.intrinsic:
            self.nets["emb"].module.load_reference(p.emb_network.state_dict(), "emb_block.hidden_layers.0", _EMB_TAIL_KEYS)
            self.nets["rnd"].module.load_reference(p.lifelong_train.state_dict(), "hidden_block.hidden_layers.0", _RND_TAIL_KEYS)
            self.nets["rnd"].target.load_reference(p.lifelong_target.state_dict(), "hidden_block.hidden_layers.0", _RND_TAIL_KEYS)
        if hasattr(self, "_published"):
            self._publish_out_of_band()

    def export_parameter(self, p=None):
        """Writes the trained networks back into the plugin Parameter's modules (backup / evaluation through the plugin surface)."""
        p = self.parameter if p is None else p
        self.join_learner()
        torch.cuda.synchronize(self.dev)
        for name, dst in (("q_ext", p.q_ext_online), ("q_int", p.q_int_online)):
            n = self.nets[name]
            dst.load_state_dict({k: v.to(next(dst.parameters()).device) for k, v in n.module.reference_state_dict().items()})
            if n.target is not None:
                getattr(p, name + "_target").load_state_dict({k: v.to(next(dst.parameters()).device) for k, v in n.target.reference_state_dict().items()})
        if self.intrinsic:
            to = lambda sd, m: m.load_state_dict({k: v.to(next(m.parameters()).device) for k, v in sd.items()})  # noqa: E731
            to(self.nets["emb"].module.reference_tensors("emb_block.hidden_layers.0", _EMB_TAIL_KEYS), p.emb_network)
            to(self.nets["rnd"].module.reference_tensors("hidden_block.hidden_layers.0", _RND_TAIL_KEYS), p.lifelong_train)
            to(self.nets["rnd"].target.reference_tensors("hidden_block.hidden_layers.0", _RND_TAIL_KEYS), p.lifelong_target)
        return p

    def _publish_out_of_band(self):
        """Everything derived from the master parameters rebuilt on the current stream (start-up, after `load_parameter`): the learner handles' packed filters, the
        target handles', and -- with sets -- set `self._set` of every actor handle (packed filters, small vectors, UVFA columns, first dense layer split into planes)."""
        for n in self.nets.values():
            n.inf.weights_changed()
            if self.sets:
                n.inf.publish_to(n.actor, self._set, with_fc1=True)
                n.actor.select_set(self._set)
            else:
                n.inf.publish_to(None)
                if n.actor is not None:
                    n.actor.weights_changed()
            if n.inf_target is not None:
                n.inf_target.weights_changed()
                n.inf_target.publish_to(None)
        if self.intrinsic and self.acts:
            r = self.nets["rnd"]
            r.actor_target.weights_changed()
            if self.sets:
                with torch.no_grad():
                    r.ln_sets[self._set][0].copy_(r.module.tail[0])
                    r.ln_sets[self._set][1].copy_(r.module.tail[1])
            self._select_rnd_ln()
        self._published = None

    def _select_rnd_ln(self):
        r = self.nets["rnd"]
        w, b = (r.ln_sets[self._set] if self.sets else (r.module.tail[0], r.module.tail[1]))
        r.actor.set_head_mode(1, self.D_rnd, w, b)

    # ---- actors ------------------------------------------------------------------------------------------------------------------------------------------------
    def arm(self) -> torch.Tensor:
        return self.ucb.arm if self.training else self.zero_arm

    def _begin_all(self):
        """on_reset of every lane (:288-311): arm from the lane's UCB controller, random previous action, zero previous rewards."""
        self.ucb.step(None, self.episode_reward)
        N.check(self.lib.srlx_agent57_begin_episodes(self.E, self.A, None, self.seed ^ 0xBE61, N.tptr(self.policy_counter), N.tptr(self.prev_action), N.tptr(self.prev_r_ext),
                                                     N.tptr(self.prev_r_int), N.tptr(self.episode_reward), N.tptr(self.reset_lane), N.tptr(self.live_lane), N.torch_stream_ptr()))

    def policy_q(self):
        """q_ext, q_int, q = q_ext + beta[arm] * q_int of every lane in its current state (:355-363); selects the actions too."""
        r, st = self.replay, N.torch_stream_ptr()
        off = r.frame_table_current()
        arm = self.arm()
        ready, self._q_ready = self._q_ready, False  # the last lock-step evaluated both Q-networks on this state already (`_next_q`)
        if not ready:
            for name, out in (("q_ext", self.q_ext), ("q_int", self.q_int)):
                h = self.nets[name].actor
                h.set_uvfa_inputs(self.prev_r_ext, self.prev_r_int, self.prev_action, arm)
                h.forward_u8(r.obs_base, off, out=out)
        c = self.cfg
        N.check(self.lib.srlx_agent57_policy(self.E, self.A, N.tptr(self.q_ext), N.tptr(self.q_int), N.tptr(arm) if self.training else None, N.tptr(self.beta_list),
                                             N.tptr(self.eps_list), float(c.test_beta), float(c.test_epsilon), self.seed ^ 0xAC7, N.tptr(self.policy_counter),
                                             N.tptr(self.actions), N.tptr(self.q), st))
        return self.q_ext, self.q_int, self.q

    def actor_step(self):
        """One lock-step of the E environments (policy -> environments -> ring commit -> intrinsic reward -> bookkeeping -> [join] tree add)."""
        c, r, st = self.cfg, self.replay, N.torch_stream_ptr()
        E = self.E
        arm = self.arm()
        self.policy_q()
        if self.before_env is not None:  # (device/dist.py: the previous slab's frames must have left before the environments overwrite them)
            self.before_env()
        next_obs, rewards, terminated, done = self.env.step(self.actions)
        slot = r._steps_committed % self.L
        if self.ledger is not None:
            self.ledger.account(rewards, done, r.needs_reset_ptr)
        # ring commit (slot p + 1 and the scalars of p belong to no stored item: beside a running update), the next pass's frame table, the policy generator's counter
        r.commit(self.actions, rewards, terminated, done, next_obs, defer_add=True, next_table=True, bump=self.policy_counter)
        epi = lif = None
        if self.intrinsic:  # :383-391 on s_{t+1}, the state the commit has just made current
            off = r.frame_table_current()
            e, rn = self.nets["emb"], self.nets["rnd"]
            if self.sets and self.multi_trunk:
                # ALL FIVE networks evaluate this state -- the embedding / RND networks now, the two Q-networks at the next policy step (their UVFA inputs enter in
                # the head kernel, behind the image block): their image blocks as ONE launch of 5 E workgroups (one ramp and one tail instead of five)
                hs = [e.actor, rn.actor_target, rn.actor, self.nets["q_ext"].actor, self.nets["q_int"].actor]
                QNetInference.forward_convs_multi(hs, r.obs_base, off)
                e.actor.forward_dense(E, out=self.emb_out)
                rn.actor_target.forward_dense(E, out=self.rnd_t_out)
                rn.actor.forward_dense(E, out=self.rnd_p_out)
                self._trunks_fresh = True
            else:
                e.actor.forward_u8(r.obs_base, off, out=self.emb_out)
                rn.actor_target.forward_u8(r.obs_base, off, out=self.rnd_t_out)
                rn.actor.forward_u8(r.obs_base, off, out=self.rnd_p_out)
            N.check(self.lib.srlx_ngu_episodic_reward(self.ngu.h, N.tptr(self.emb_out), N.tptr(self.reset_lane), N.tptr(self.live_lane), N.tptr(self.episodic), st))
            N.check(self.lib.srlx_ngu_lifelong_reward(E, self.D_rnd, N.tptr(self.rnd_t_out), N.tptr(self.rnd_p_out), float(c.lifelong_max), N.tptr(self.lifelong), st))
            epi, lif = self.episodic, self.lifelong
        row = lambda t: N.c_p(t.data_ptr() + slot * E * t.element_size())  # noqa: E731
        self._last_slot = slot
        N.check(self.lib.srlx_agent57_post_step(E, N.tptr(self.actions), N.tptr(arm), N.tptr(rewards), N.tptr(self.reset_lane), N.tptr(epi), N.tptr(lif), N.tptr(self.prev_action),
                                                N.tptr(self.prev_r_ext), N.tptr(self.prev_r_int), N.tptr(self.episode_reward), row(self.x_r_int), row(self.x_prev_r_ext),
                                                row(self.x_prev_r_int), row(self.x_actor), row(self.x_prev_action), st))
        if self.training:  # lanes whose episode just ended: book it with their UCB controller, draw the next arm, on_reset
            self.ucb.step(done, self.episode_reward)
            N.check(self.lib.srlx_agent57_begin_episodes(E, self.A, N.tptr(done), self.seed ^ 0xBE61, N.tptr(self.policy_counter), N.tptr(self.prev_action), N.tptr(self.prev_r_ext),
                                                         N.tptr(self.prev_r_int), N.tptr(self.episode_reward), N.tptr(self.reset_lane), N.tptr(self.live_lane), st))
        else:
            self.reset_lane.copy_(done)
            torch.bitwise_xor(done, 1, out=self.live_lane)
        self.total_env_steps += E
        self._next_q()
        if self._own_ring_only:  # the ring only stacks frames for this engine's actors: its commit moved the position itself, there is no tree here
            return
        self.join_learner()
        r.add_masked()
        self._flip()

    def _next_q(self):
        """The Q-networks' dense layers for the NEXT policy step, on the operand planes the multi-network launch of this lock-step left: everything they need exists once
        the lock-step's bookkeeping is done (the state, the previous action / rewards, the arm).  Run here -- before the join -- they use the same parameter set as
        their image blocks (the flip to the set the joined update wrote comes behind them), and they are more work beside the update."""
        if not self._trunks_fresh:
            return
        self._trunks_fresh = False
        arm = self.arm()
        for name, out in (("q_ext", self.q_ext), ("q_int", self.q_int)):
            h = self.nets[name].actor
            h.set_uvfa_inputs(self.prev_r_ext, self.prev_r_int, self.prev_action, arm)
            h.forward_dense(self.E, out=out)
        self._q_ready = True

    def pack_record(self) -> torch.Tensor:
        """The lock-step `actor_step` has just taken as ONE packed record (uint8: action, reward, flags and the five item fields of every lane): what an actor rank
        ships to the learner rank beside its frames (`env.next_obs`)."""
        e, E, slot = self.env, self.E, self._last_slot
        row = lambda t: N.c_p(t.data_ptr() + slot * E * t.element_size())  # noqa: E731
        N.check(self.lib.srlx_agent57_pack_record(E, N.tptr(self.actions), N.tptr(e.rewards), N.tptr(e.terminated), N.tptr(e.done), row(self.x_r_int), row(self.x_actor),
                                                  row(self.x_prev_action), row(self.x_prev_r_ext), row(self.x_prev_r_int), N.tptr(self.record), N.torch_stream_ptr()))
        return self.record

    def _flip(self):
        """The joined update wrote the other set of every network: the next passes read it (host-side pointer swaps)."""
        if self.sets and self._published is not None and self.acts:
            self._set, self._published = self._published, None
            for n in self.nets.values():
                n.actor.select_set(self._set)
            if self.intrinsic:
                self._select_rnd_ln()

    # ---- learner -----------------------------------------------------------------------------------------------------------------------------------------------
    def _forward_q(self, n: _Net, target_only: bool = False, online_only: bool = False):
        """The passes of one Q-network's update (model_torch.py:384-443) on the CURRENT stream: the target network on s_1, the online network over the interleaved
        [s_0, s_1] rows."""
        r = self.lreplay
        B, W = r.B, self.Wn
        o = self.out[n.name]
        if not online_only:
            n.inf_target.set_uvfa_inputs(self.tg_r_ext, self.tg_r_int, self.tg_action, self.tg_actor)
            n.inf_target.forward_u8(r.obs_base, r.frame_off_next.view(B, W), out=o["q_tg"])
        if not target_only:
            n.inf.set_uvfa_inputs(self.on_r_ext, self.on_r_int, self.on_action, self.on_actor)
            n.inf.forward_u8(r.obs_base, r.frame_off_all.view(2 * B, W), out=o["q_all"])

    def _backward_q(self, n: _Net, rewards, publish, bump):
        """TD target (per-actor discount) / Huber / gradient seed in the backward pass's head kernel, Adam inside the gradient launches, publish."""
        c, r = self.cfg, self.lreplay
        B = r.B
        o = self.out[n.name]
        n.inf.set_td_extras(self.b_discount, o["td"])
        if self.sets:
            n.inf.fuse_adam_planes(n.planes_ptr[publish] if publish is not None else None)
        b = r.batch
        n.inf.backward_td_u8(r.obs_base, r.frame_off_all, 1, o["q_all"].view(B, 2, self.A), o["q_tg"], b.actions, rewards, b.terminated, b.weights, 0.0, 1.0,
                             c.enable_double_dqn, c.enable_rescale, o["target"], o["loss"], o["grad_q0"], o["pri"])
        n.opt.step(self.train_count_dev)  # (fused: nothing left to launch)
        n.inf.publish_to(n.actor if publish is not None else None, publish or 0, bump=bump)

    def _learner_body(self, publish: Optional[int] = None, drawn: bool = False, ingest=None):
        """One update of the four trained networks (model_torch.py:263-381) -- device work only, capturable.  publish: the actor set (0 / 1) it also writes.
        drawn=True (tests): the batch, its frame tables and its UVFA inputs are already in the engine's buffers.  ingest: a callable issuing the launches that commit
        a slab of arrived transitions (ring + item fields + tree; device/dist.py): they run on a side stream BEHIND the draw, and the priority write-back waits for
        them -- the tree sees draw, add, write-back in that order, and the commit hides beside the networks' passes."""
        r = self.lreplay
        cur = torch.cuda.current_stream(self.dev)

        def pre():
            b = r.sample_items(self.train_count_dev, all_states=True)
            self._gather_inputs(b, N.torch_stream_ptr())

        if not drawn:
            pre()
        if ingest is not None:  # behind the draw AND the gather of the drawn items' fields (the ingest overwrites a ring slot's fields: one no stored item points at)
            self._ev_drawn.record(cur)
            self.s_ingest.wait_event(self._ev_drawn)
            with torch.cuda.stream(self.s_ingest):
                ingest()
                self._ev_ingested.record(self.s_ingest)
        self._update_networks(r.batch, publish, wait=self._ev_ingested if ingest is not None else None)

    def _lx(self):
        """The learner's view of the item fields: the global replay's arrays on a learner rank, this engine's own otherwise."""
        if self.learner_replay is not None:
            x = self.lx
            return x["r_int"], x["prev_r_ext"], x["prev_r_int"], x["actor"], x["prev_action"]
        return self.x_r_int, self.x_prev_r_ext, self.x_prev_r_int, self.x_actor, self.x_prev_action

    def ingest_fields(self, records: torch.Tensor, envs_per_record: int):
        """Learner rank: the five item fields of a slab of packed records -> row (ring position mod L) of the global replay's field arrays.  Launch BETWEEN the ring
        commit of the same slab (deferred advance) and the tree add that moves the position."""
        rp = self.learner_replay
        xr, xpe, xpi, xa, xpa = self._lx()
        N.check(self.lib.srlx_agent57_unpack_fields(records.shape[0], int(envs_per_record), N.tptr(records), records.shape[1], rp._views[0], rp.L, N.tptr(xr), N.tptr(xa),
                                                    N.tptr(xpa), N.tptr(xpe), N.tptr(xpi), N.torch_stream_ptr()))

    def _gather_inputs(self, b, st):
        r, B = self.lreplay, self.lreplay.B
        xr, xpe, xpi, xa, xpa = self._lx()
        N.check(self.lib.srlx_store_locate(r.h_store, B, N.tptr(b.indices), N.tptr(self.loc_env), N.tptr(self.loc_slot), None, st))
        N.check(self.lib.srlx_agent57_gather_inputs(B, r.E, N.tptr(self.loc_env), N.tptr(self.loc_slot), N.tptr(b.actions), N.tptr(b.rewards), N.tptr(xr),
                                                    N.tptr(xpe), N.tptr(xpi), N.tptr(xa), N.tptr(xpa), N.tptr(self.discount_list),
                                                    N.tptr(self.on_r_ext), N.tptr(self.on_r_int), N.tptr(self.on_action), N.tptr(self.on_actor), N.tptr(self.tg_r_ext),
                                                    N.tptr(self.tg_r_int), N.tptr(self.tg_action), N.tptr(self.tg_actor), N.tptr(self.b_discount), N.tptr(self.b_r_int), st))

    def _forward_emb(self):
        """The inverse-dynamics embedding (:341-348): rows 2 b = f(s), 2 b + 1 = f(s'), all with gradient; its dense tail (forward, loss, backward to the
        embeddings, the tail's own Adam steps) in one launch behind the pass."""
        c, r, e = self.cfg, self.lreplay, self.nets["emb"]
        B, b = r.B, r.batch
        e.inf.forward_u8(r.obs_base, r.frame_off_all.view(2 * B, self.Wn), out=self.l_emb)
        tp, tg, tm, tv = e.tail_tabs
        N.check(self.lib.srlx_agent57_emb_tail(B, self.D_emb, self.H_emb, self.A, N.tptr(self.l_emb), N.tptr(b.actions), ctypes.cast(tp, N.c_p), ctypes.cast(tg, N.c_p),
                                               ctypes.cast(tm, N.c_p), ctypes.cast(tv, N.c_p), 1e-5, float(c.episodic_lr), 0.9, 0.999, 1e-8, N.tptr(self.train_count_dev),
                                               N.tptr(self.emb_loss), N.tptr(self.g_emb), N.torch_stream_ptr()))

    def _backward_emb(self, publish):
        r, e = self.lreplay, self.nets["emb"]
        if self.sets:
            e.inf.fuse_adam_planes(e.planes_ptr[publish] if publish is not None else None)
        e.inf.backward_u8(r.obs_base, r.frame_off_all.view(2 * r.B, self.Wn), self.g_emb, sample_stride=1)
        e.opt.step(self.train_count_dev)
        e.inf.publish_to(e.actor if publish is not None else None, publish or 0)

    def _forward_rnd(self, publish):
        """RND (:353-362): the predictor and the fixed target network (its rows 0, 2, ... = s_0 are what the loss reads), and the predictor's LayerNorm tail
        (forward, loss, backward, its Adam steps, the published set's copy of its parameters) in one launch behind them."""
        c, r, rn = self.cfg, self.lreplay, self.nets["rnd"]
        B = r.B
        off2 = r.frame_off_all.view(2 * B, self.Wn)
        rn.inf_target.forward_u8(r.obs_base, off2, out=self.l_rnd_t)
        rn.inf.forward_u8(r.obs_base, off2, out=self.l_rnd_p)
        mw, mb = (rn.ln_sets[publish] if (self.sets and publish is not None) else (None, None))
        N.check(self.lib.srlx_agent57_rnd_tail(B, self.D_rnd, 2 * self.D_rnd, N.tptr(self.l_rnd_p), N.tptr(self.l_rnd_t), N.tptr(rn.module.tail[0]), N.tptr(rn.module.tail[1]),
                                               N.tptr(rn.tail_g[0]), N.tptr(rn.tail_g[1]), N.tptr(rn.tail_m[0]), N.tptr(rn.tail_v[0]), N.tptr(rn.tail_m[1]),
                                               N.tptr(rn.tail_v[1]), N.tptr(mw), N.tptr(mb), 1e-5, float(c.lifelong_lr), 0.9, 0.999, 1e-8, N.tptr(self.train_count_dev),
                                               N.tptr(self.rnd_loss), N.tptr(self.g_rnd), N.torch_stream_ptr()))

    def _backward_rnd(self, publish, bump=None):
        r, rn = self.lreplay, self.nets["rnd"]
        if self.sets:
            rn.inf.fuse_adam_planes(rn.planes_ptr[publish] if publish is not None else None)
        rn.inf.backward_u8(r.obs_base, r.frame_off_all.view(2 * r.B, self.Wn), self.g_rnd, sample_stride=2)
        rn.opt.step(self.train_count_dev)
        rn.inf.publish_to(rn.actor if publish is not None else None, publish or 0, bump=bump)

    def _update_networks(self, b, publish, wait=None):
        """The four networks' updates, then the mixed priorities and their write-back.  A network's FORWARD passes do not depend on the other networks' updates:
        they (and the embedding / RND tails behind them) all run on one side stream (`s_target`, forked once from the update's own stream), the backward passes -- each forks its weight-gradient stream -- one
        after the other on the update's stream, each behind its forward's event: the forward of network k + 1 runs beside the backward of network k.  (A BACKWARD pass on the side stream -- RND's, with its weight-gradient launches in line -- made the update 2.24 ms instead of 0.81; a whole
        update per branch is not possible either: a captured stream that forks again faults in hipStreamEndCapture -- tools/capture_probe.py --, and a graph per network
        on streams of their own runs 2.5 ms instead of 1.1 once the process owns a low-priority stream, which the actors need: profiles/NOTES.md, round 6.)"""
        c, r = self.cfg, self.lreplay
        cur, sf = torch.cuda.current_stream(self.dev), self.s_target
        last = "rnd" if self.intrinsic else "q_ext"
        q_ext = self.nets["q_ext"]
        ev = self._ev_fwd
        self._ev_fork_fwd.record(cur)
        sf.wait_event(self._ev_fork_fwd)
        with torch.cuda.stream(sf):
            self._forward_q(q_ext, target_only=True)
            ev["q_ext"].record(sf)
            if self.intrinsic and self.hoist_forwards:
                self._forward_q(self.nets["q_int"])
                ev["q_int"].record(sf)
                self._forward_emb()
                ev["emb"].record(sf)
                self._forward_rnd(publish)
                ev["rnd"].record(sf)
        self._forward_q(q_ext, online_only=True)
        cur.wait_event(ev["q_ext"])
        self._backward_q(q_ext, b.rewards, publish, self.train_count_dev if last == "q_ext" else None)
        if self.intrinsic:
            hoist = self.hoist_forwards
            cur.wait_event(ev["q_int"]) if hoist else self._forward_q(self.nets["q_int"])
            self._backward_q(self.nets["q_int"], self.b_r_int, publish, None)
            cur.wait_event(ev["emb"]) if hoist else self._forward_emb()
            self._backward_emb(publish)
            cur.wait_event(ev["rnd"]) if hoist else self._forward_rnd(publish)
            self._backward_rnd(publish, bump=self.train_count_dev)
        # ---- mixed priorities (:367-373) and their write-back ----
        use_int = self.intrinsic and not c.disable_int_priority
        N.check(self.lib.srlx_agent57_priority(r.B, self.A, N.tptr(self.out["q_ext"]["td"]), None, N.tptr(self.out["q_int"]["td"]) if use_int else None, None, None,
                                               N.tptr(self.tg_actor), N.tptr(self.beta_list), None, None, N.tptr(self.priorities), N.torch_stream_ptr()))
        if wait is not None:
            cur.wait_event(wait)
        r.update(b.indices, self.priorities)

    def _after_update(self):
        """The host-side tail of an update: target sync every `target_model_update_interval` updates (fires at 0 too, :376-379), counters."""
        if self.train_count % self.cfg.target_model_update_interval == 0:
            with torch.no_grad():
                for name in ("q_ext", "q_int"):
                    n = self.nets[name]
                    torch._foreach_copy_(list(n.target.parameters()), list(n.module.parameters()))
                    n.inf_target.weights_changed()
                    n.inf_target.publish_to(None)
            self.sync_count += 1
        self.train_count += 1

    def learner_step(self, publish: Optional[int] = None) -> bool:
        """Returns False while the replay is below warm-up.  A pending `self.ingest` rides on this update."""
        if self.lreplay.is_warmup_needed():
            return False
        ing, self.ingest = self.ingest, None
        ing_key, ing_fn = (ing[0], ing[1]) if ing is not None else (None, None)
        key = (publish, ing_key)
        g = self._graphs.get(key)
        if g is None and self._capturing and not self._in_capture:
            torch.cuda.current_stream(self.dev).synchronize()
            g = torch.cuda.CUDAGraph()
            self._in_capture = True
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (other threads -- the RCCL watchdog -- may touch the runtime meanwhile)
                    self._learner_body(publish, ingest=ing_fn)
            finally:
                self._in_capture = False
            self._graphs[key] = g
        if g is not None:
            g.replay()
        else:
            self._learner_body(publish, ingest=ing_fn)
        self._after_update()
        return True

    def run_updates(self, updates: int) -> int:
        """`updates` updates NOT beside this engine's actors (a learner-only rank): on the learner's stream, ordered after the current stream and joined back to it.  A
        pending `ingest` rides on the first update or runs by itself."""
        cur = torch.cuda.current_stream(self.dev)
        self.s_learner.wait_stream(cur)
        ran = 0
        with torch.cuda.stream(self.s_learner):
            for _ in range(updates):
                ran += int(self.learner_step(None))
            if self.ingest is not None:
                ing, self.ingest = self.ingest, None
                ing[1]()
        cur.wait_stream(self.s_learner)
        return ran

    def fork_learner(self, updates: int) -> int:
        """overlap: `updates` updates on the learner's stream, ordered after everything enqueued on the current stream so far; the last one publishes."""
        if (updates <= 0 or self.lreplay.is_warmup_needed()) and self.ingest is None:
            return 0
        self._ev_fork.record(torch.cuda.current_stream(self.dev))
        self.s_learner.wait_event(self._ev_fork)
        ran = 0
        with torch.cuda.stream(self.s_learner):
            for k in range(updates):
                pub = 1 - self._set if k == updates - 1 else None
                if self.learner_step(pub):
                    ran += 1
                    if pub is not None:
                        self._published = pub
            if self.ingest is not None:  # no update took the pending ingest with it (warm-up, or none asked for): commit it here, in stream order
                ing, self.ingest = self.ingest, None
                ing[1]()
            self._ev_join.record(self.s_learner)
        self._learner_pending = True
        return ran

    def join_learner(self):
        if getattr(self, "_learner_pending", False):
            torch.cuda.current_stream(self.dev).wait_event(self._ev_join)
            self._learner_pending = False

    def capture_graphs(self, warm_updates: int = 1):
        """From now on every update variant (publishing into set 0 / 1 / none) is captured into a HIP graph the first time it runs.  Call once the replay is warm:
        `warm_updates` eager updates run first (library scratch, event creation)."""
        if self._capturing or self.lreplay.is_warmup_needed():
            return
        self.join_learner()
        torch.cuda.synchronize(self.dev)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(warm_updates):
                self.learner_step(None)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        if self.sets:
            self._publish_out_of_band()  # (the warm updates trained the master without publishing)
        self._capturing = True

    def step(self, learner_updates: int = 1, events=None):
        if self.overlap:
            self.fork_learner(learner_updates)
        if events is not None:
            events[0].record()
        self.actor_step()  # (joins the forked updates before its tree add)
        if events is not None:
            events[1].record()
        if not self.overlap:
            for _ in range(learner_updates):
                self.learner_step(None)

    def prefill(self, randomise_priorities: bool = True):
        """Random-policy rollout until every PER leaf holds an item (untimed benchmark set-up)."""
        saved = self.eps_list
        self.eps_list = torch.ones_like(saved)
        for _ in range(self.replay.item_len):
            self.actor_step()
        self.eps_list = saved
        if randomise_priorities:
            g = torch.Generator(device=self.dev)
            g.manual_seed(self.seed + 1)
            pri = torch.rand(self.replay.capacity, dtype=torch.float32, device=self.dev, generator=g)
            N.check(self.lib.srlx_per_set_range(self.replay.h_per, 0, self.replay.capacity, N.tptr(pri), N.PRIO_F32, 1, N.torch_stream_ptr()))
        torch.cuda.synchronize(self.dev)

    def losses(self) -> dict:
        out = {"ext_loss": float(self.out["q_ext"]["loss"].item()), "sync": self.sync_count}
        if self.intrinsic:
            out.update(int_loss=float(self.out["q_int"]["loss"].item()), emb_loss=float(self.emb_loss.item()), lifelong_loss=float(self.rnd_loss.item()))
        return out

    def info(self) -> dict:
        self.join_learner()
        check_ranges()
        d = dict(train_count=self.train_count, memory=self.lreplay.length())
        if self.train_count > 0:
            d.update(self.losses())
            d["loss"] = d["ext_loss"]
        return d

    def close(self):
        """Joins the learner; hands the calling thread back to the stream it was on before the engine took it to its actors' stream."""
        self.join_learner()
        torch.cuda.synchronize(self.dev)
        if getattr(self, "actor_stream", None) is not None:
            torch.cuda.set_stream(self._stream_before)
            self.actor_stream = None
            N.check(self.lib.srlx_stream_destroy(self._actor_stream_raw))
