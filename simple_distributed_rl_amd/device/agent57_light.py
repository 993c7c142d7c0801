"""Device-resident Agent57_light actor/learner: E lock-stepped environments on one GPU (BASELINE.json configs[3] workload).

The reference runs ONE environment per actor process (srl/algorithms/agent57_light/agent57_light.py:271-471) and trains from
pickled 11-field items (:420-432, model_torch.py:263-443).  Here every per-actor quantity becomes a per-environment device
array and every list comprehension a kernel:

    UVFA inputs (prev action / prev extrinsic reward / prev intrinsic reward / one-hot arm)   per-environment tensors
    sliding-window UCB meta-controller (:317-353), one per environment                          srlx_agent57_ucb_step
    epsilon-greedy on q_ext + beta[arm] * q_int (:355-375)                                      srlx_policy_epsilon_greedy
    frame stacking, item storage (state, next state, action, reward, undone)                    srlx_store_* (uint8 ring, n_step = 1)
    the other item fields (intrinsic reward, arm, previous action / rewards)                    [ring slot][env] arrays, gathered
                                                                                                by (slot, env) from srlx_store_locate
    episodic novelty: kNN over the episode's embeddings (:473-513)                              srlx_ngu_episodic_reward (E memories)
    lifelong novelty: clipped RND error (:515-529)                                              srlx_ngu_lifelong_reward
    per-arm-discount double-DQN targets, Huber + gradient seed, |td_ext + beta td_int|          srlx_dqn_target, fused TD kernel,
                                                                                                srlx_agent57_priority
    proportional replay                                                                         srlx_per_*

The five networks (two UVFA Q-networks, the inverse-dynamics embedding, RND target / predictor) keep the plugin's torch
module trees (algorithms/agent57_light.py: the reference's, so parameters stay interchangeable), but on this engine their
IMAGE BLOCKS never run through torch: actors and learner evaluate them with libsrlx's fused convolution kernel straight from
the uint8 ring (device/qnet.py:ImageTrunk / TrainableImageTrunk, weights bound by address) and the learner differentiates
them with the hand-written backward (srlx_qnet_backward_convs_u8); only the dense tails and Adam are torch (hipBLASLt).
SRLX_A57_TORCH_LEARNER=1 puts the learner back on torch's convolutions (a test yardstick).  `Agent57LightLearner` is shared
with the single-environment plugin trainer.  With overlap=True the update runs on its own stream beside the next lock-step's
actors, which act on copies of the networks they need (refreshed after the join), like the Rainbow engine.
"""
import ctypes
import functools
import os
from typing import Optional

import numpy as np
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.algorithms._device_ops import NguOps, TdOps
from simple_distributed_rl_amd.device.replay import DeviceReplay
from simple_distributed_rl_amd.rl import functions as funcs


def q_values(net, state_cf, r_ext, r_int, onehot_action, onehot_actor, features=None):
    """QNetwork.forward (agent57_light/model_torch.py:35-64) on a channels-first float32 stack (or on `features` = the image block's
    flattened output computed elsewhere: device/qnet.py:ImageTrunk)."""
    parts = [net.in_block(state_cf, channels_first=True) if features is None else features]
    if net.input_ext_reward:
        parts.append(r_ext)
    if net.input_int_reward:
        parts.append(r_int)
    if net.input_action:
        parts.append(onehot_action)
    parts.append(onehot_actor)
    return net.hidden_block(torch.cat(parts, dim=1))


def embed(net, state_cf, features=None):
    return net.emb_block(net.in_block(state_cf, channels_first=True) if features is None else features)


def rnd(net, state_cf, features=None):
    return net.hidden_normalize(net.hidden_block(net.in_block(state_cf, channels_first=True) if features is None else features))


class UcbBank:
    """E sliding-window UCB meta-controllers in HBM (agent57_light.py:317-353)."""

    def __init__(self, n_envs: int, n_arms: int, window: int, epsilon: float, beta: float, device: torch.device, seed: int):
        self.E, self.n_arms, self.window, self.epsilon, self.beta, self.dev, self.seed = n_envs, n_arms, window, epsilon, beta, device, seed
        self.lib = N.lib()
        d = device
        self.ring_arm = torch.zeros((n_envs, window), dtype=torch.int32, device=d)
        self.ring_reward = torch.zeros((n_envs, window), dtype=torch.float32, device=d)
        self.head = torch.zeros(n_envs, dtype=torch.int32, device=d)
        self.n_recent = torch.zeros(n_envs, dtype=torch.int32, device=d)
        self.count = torch.ones((n_envs, n_arms), dtype=torch.int32, device=d)
        self.sum = torch.zeros((n_envs, n_arms), dtype=torch.float64, device=d)
        self.arm = torch.full((n_envs,), -1, dtype=torch.int32, device=d)
        self.u = torch.zeros((n_envs, 3), dtype=torch.float64, device=d)
        self.counter = torch.zeros(1, dtype=torch.int64, device=d)

    def step(self, done: Optional[torch.Tensor], episode_reward: torch.Tensor, uniforms: Optional[torch.Tensor] = None):
        st = N.torch_stream_ptr()
        if uniforms is None:
            N.check(self.lib.srlx_rng_uniform(self.seed ^ 0x0C8, N.tptr(self.counter), self.u.numel(), N.tptr(self.u), st))
            uniforms = self.u
        N.check(self.lib.srlx_agent57_ucb_step(self.E, self.n_arms, self.window, N.tptr(self.ring_arm), N.tptr(self.ring_reward), N.tptr(self.head),
                                               N.tptr(self.n_recent), N.tptr(self.count), N.tptr(self.sum), N.tptr(self.arm), N.tptr(done),
                                               N.tptr(episode_reward), N.tptr(uniforms), float(self.epsilon), float(self.beta), st))
        return self.arm


class Agent57LightLearner:
    """One Agent57_light update on device tensors (model_torch.py:263-443): both Q-networks, the embedding and the RND
    predictor, the mixed priorities.  Used by the plugin Trainer (one host batch put on the GPU) and by the engine."""

    def __init__(self, config, parameter, device: torch.device, channels_first: bool):
        c, p = config, parameter
        self.config, self.parameter, self.device, self.cf = c, p, device, channels_first
        self.ops = TdOps(device)
        # multi-tensor Adam with its step count on the device: one launch per optimiser, and the whole update can live in a HIP graph
        kw = dict(capturable=True, fused=True)
        self.q_ext_optimizer = torch.optim.Adam(p.q_ext_online.parameters(), lr=c.lr_ext, **kw)
        self.q_int_optimizer = torch.optim.Adam(p.q_int_online.parameters(), lr=c.lr_int, **kw)
        self.emb_optimizer = torch.optim.Adam(p.emb_network.parameters(), lr=c.episodic_lr, **kw)
        self.lifelong_optimizer = torch.optim.Adam(p.lifelong_train.parameters(), lr=c.lifelong_lr, **kw)
        self.beta_list = torch.tensor(np.array(funcs.create_beta_list(c.actor_num), np.float32), device=device)
        self.discount_list = torch.tensor(np.array(funcs.create_discount_list(c.actor_num), np.float32), device=device)
        self.actor_eye = torch.eye(c.actor_num, dtype=torch.float32, device=device)
        self.action_eye = torch.eye(c.action_space.n, dtype=torch.float32, device=device)
        self.train_count = 0
        self.sync_count = 0
        self.info: dict = {}

    def _q(self, net, inputs, feat=None):
        if feat is not None:  # the image block's output computed elsewhere (device/qnet.py:TrainableImageTrunk)
            return q_values(net, None, *inputs[1:], features=feat)
        return q_values(net, *inputs) if self.cf else net(inputs)

    def _update_q(self, online, target_net, optimizer, rewards, next_inputs, cur_inputs, undone, discount, inv, action, w, feats=None):
        """model_torch.py:384-443 with the arithmetic around the three forwards in libsrlx.  feats = (features of the current states WITH gradient,
        of the next states by the online network, of the next states by the target network) or None (the torch image blocks)."""
        cfg = self.config
        f_cur, f_next_on, f_next_tg = feats if feats is not None else (None, None, None)
        with torch.no_grad():  # agent57_light.py:241-257
            online.eval()
            q_tg_next = self._q(target_net, next_inputs, f_next_tg)
            q_on_next = self._q(online, next_inputs, f_next_on) if cfg.enable_double_dqn else None
        target = self.ops.dqn_target(q_on_next, q_tg_next, rewards, undone, inv, 0.0, cfg.enable_double_dqn, cfg.enable_rescale, False, discount_per_sample=discount)
        online.train()
        q = self._q(online, cur_inputs, f_cur)
        _, loss, grad, _ = self.ops.huber(target, q, action, w)
        optimizer.zero_grad()
        q.backward(grad)
        optimizer.step()
        return target, q.detach(), loss

    def update(self, states, n_states, action, r_ext, r_int, undone, prev_action, prev_r_ext, prev_r_int, actor, weights, invalid=None):
        """states / n_states float32 (channels-first stacks for the engine, the reference's layout for the plugin); action / prev_action /
        actor int tensors [B]; the rest float32 [B].  Returns the new priorities (device float32 [B])."""
        priorities = self.update_networks(states, n_states, action, r_ext, r_int, undone, prev_action, prev_r_ext, prev_r_int, actor, weights, invalid)
        self.after_update()
        return priorities

    def after_update(self):
        """The host-side tail of an update: target sync every `target_model_update_interval` updates (fires at 0 too, :376-379), counters."""
        cfg, p = self.config, self.parameter
        if self.train_count % cfg.target_model_update_interval == 0:
            with torch.no_grad():
                for tgt, src in ((p.q_ext_target, p.q_ext_online), (p.q_int_target, p.q_int_online)):
                    torch._foreach_copy_(list(tgt.parameters()), list(src.parameters()))
                    torch._foreach_copy_(list(tgt.buffers()), list(src.buffers())) if list(tgt.buffers()) else None
            self.sync_count += 1
        self.train_count += 1

    def update_networks(self, states, n_states, action, r_ext, r_int, undone, prev_action, prev_r_ext, prev_r_int, actor, weights, invalid=None, features=None):
        """The device part of an update (everything of `update` but the target sync and the counters): only enqueues work, capturable.
        features: the image blocks' outputs from hand-written trunks (Agent57LightEngine.learner_features; states / n_states are then unused) --
        {"q_ext": (cur with grad, next online, next target), "q_int": ..., "emb": (cur, next: both with grad), "ll": (train with grad, target)}."""
        cfg, p = self.config, self.parameter
        ft = features or {}
        B = action.shape[0]
        actor_onehot = self.actor_eye[actor.long()]
        discount = self.discount_list[actor.long()]  # model_torch.py:287
        onehot_action = self.action_eye[action.long()]
        next_inputs = [n_states, r_ext.view(B, 1), r_int.view(B, 1), onehot_action, actor_onehot]  # :294-299
        cur_inputs = [states, prev_r_ext.view(B, 1), prev_r_int.view(B, 1), self.action_eye[prev_action.long()], actor_onehot]  # :427-433
        action32 = action.to(torch.int32)
        tgt_e, q_e, ext_loss = self._update_q(p.q_ext_online, p.q_ext_target, self.q_ext_optimizer, r_ext, next_inputs, cur_inputs, undone, discount, invalid,
                                              action32, weights, ft.get("q_ext"))
        self.ext_loss = ext_loss
        tgt_i = q_i = None
        if cfg.enable_intrinsic_reward:
            tgt_i, q_i, int_loss = self._update_q(p.q_int_online, p.q_int_target, self.q_int_optimizer, r_int, next_inputs, cur_inputs, undone, discount, invalid,
                                                  action32, weights, ft.get("q_int"))
            self.int_loss = int_loss
            # inverse-dynamics embedding (:341-348)
            p.emb_network.train()
            if "emb" in ft:
                h = torch.cat([embed(p.emb_network, None, ft["emb"][0]), embed(p.emb_network, None, ft["emb"][1])], dim=1)
                probs = torch.softmax(p.emb_network.out_block_out1(p.emb_network.out_block_normalize(p.emb_network.out_block(h))), dim=1)
            elif self.cf:
                h = torch.cat([embed(p.emb_network, states), embed(p.emb_network, n_states)], dim=1)
                probs = torch.softmax(p.emb_network.out_block_out1(p.emb_network.out_block_normalize(p.emb_network.out_block(h))), dim=1)
            else:
                probs = p.emb_network([states, n_states])
            emb_loss = torch.nn.functional.mse_loss(probs, onehot_action)
            self.emb_optimizer.zero_grad()
            emb_loss.backward()
            self.emb_optimizer.step()
            self.emb_loss = emb_loss.detach()
            # RND (:353-362)
            with torch.no_grad():
                if "ll" in ft:
                    lifelong_target_val = rnd(p.lifelong_target, None, ft["ll"][1])
                else:
                    lifelong_target_val = rnd(p.lifelong_target, states) if self.cf else p.lifelong_target(states)
            p.lifelong_train.train()
            if "ll" in ft:
                lifelong_train_val = rnd(p.lifelong_train, None, ft["ll"][0])
            else:
                lifelong_train_val = rnd(p.lifelong_train, states) if self.cf else p.lifelong_train(states)
            lifelong_loss = torch.nn.functional.mse_loss(lifelong_target_val, lifelong_train_val)
            self.lifelong_optimizer.zero_grad()
            lifelong_loss.backward()
            self.lifelong_optimizer.step()
            self.lifelong_loss = lifelong_loss.detach()
        use_int = cfg.enable_intrinsic_reward and not cfg.disable_int_priority  # :367-372
        self.td_ext, self.td_int, priorities = self.ops.agent57_priority(tgt_e, q_e, tgt_i if use_int else None, q_i if use_int else None, action32,
                                                                         actor.to(torch.int32), self.beta_list)
        return priorities

    def losses(self) -> dict:
        out = {"ext_loss": float(self.ext_loss.item()), "sync": self.sync_count}
        if self.config.enable_intrinsic_reward:
            out.update(int_loss=float(self.int_loss.item()), emb_loss=float(self.emb_loss.item()), lifelong_loss=float(self.lifelong_loss.item()))
        return out


def _miopen_find(fn):
    """Runs an engine entry point with torch.backends.cudnn.benchmark on (MIOpen find mode) and restores the previous setting."""

    @functools.wraps(fn)
    def scoped(self, *a, **kw):
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):
            return fn(self, *a, **kw)

    return scoped


class _GraphLauncher:
    """A helper thread that enqueues the update (its HIP-graph launch keeps the calling thread inside hipGraphLaunch for milliseconds; torch releases the GIL there,
    so the main thread goes on issuing the actors' launches).  One request at a time: submit, then wait."""

    def __init__(self, device: torch.device):
        import queue
        import threading

        self._q, self._done, self._err, self._dev = queue.SimpleQueue(), threading.Event(), None, device
        self._done.set()
        self._thread = threading.Thread(target=self._run, name="srlx-a57-launcher", daemon=True)
        self._thread.start()

    def close(self):
        """Ends the helper thread (a sentinel request); the launcher cannot be used afterwards."""
        if self._thread is not None:
            self.wait()
            self._q.put(None)
            self._thread.join(timeout=5)
            self._thread = None

    def _run(self):
        torch.cuda.set_device(self._dev)
        while True:
            fn = self._q.get()
            if fn is None:
                return
            try:
                fn()
            except BaseException as e:  # surfaces in wait()
                self._err = e
            self._done.set()

    def submit(self, fn):
        self.wait()
        self._done.clear()
        self._q.put(fn)

    def wait(self):
        self._done.wait()
        if self._err is not None:
            e, self._err = self._err, None
            raise e


class Agent57LightEngine:
    """E environments + learner on one GPU.  `rl_config`: a set-up algorithms.agent57_light.Config (image observations, window 4);
    `parameter`: its Parameter (the five networks), created here when not given."""

    def __init__(self, rl_config, n_envs: int, device: int = 0, episode_len: int = 200, seed: int = 0, env=None, parameter=None, ring_len: Optional[int] = None,
                 overlap: bool = False):
        """overlap=True (round 4): the update runs on its own stream BESIDE the actors' lock-step, forked before the policy pass and joined before the ring commit,
        exactly like RainbowEngine's overlap: the actors act on private copies of the two Q-networks (refreshed after every join; the reference's distributed actors
        poll the trainer's parameter board on a timer, play_mp.py:151-165).  The update's HIP graph is launched by a helper thread: hipGraphLaunch keeps its caller
        busy for the whole ~2.4 ms of that graph (tools/_r4_a57.sh), during which the main thread issues the actors' launches."""
        from simple_distributed_rl_amd.device.rainbow import SyntheticAtariVecEnv

        c = self.cfg = rl_config
        assert c.is_setup(), "rl_config.setup(env) first: the networks are built from the negotiated spaces"
        self.dev = torch.device(f"cuda:{device}")
        self.lib = N.lib()
        # MIOpen's immediate mode falls back to im2col-per-image / naive fp32 convolutions on gfx950 (profiles/r1_kernel_stats_before_find.csv):
        # the engine's entry points run under `_miopen_find` (solver benchmarking once per shape), scoped so that the process-wide flag --
        # and with it the convolution algorithms every other torch user of the process gets -- is left as it was
        self.E, self.seed = int(n_envs), int(seed)
        shape = c.observation_space.shape  # (H, W, window)
        H, W_, Wn = int(shape[0]), int(shape[1]), int(shape[2])
        assert Wn == c.window_length
        self.hw, self.Wn, self.A = (H, W_), Wn, c.action_space.n
        E, B = self.E, c.batch_size
        mem = c.memory
        kw = mem.kwargs if mem.name != "ReplayBuffer" else {}
        pad = 1 + Wn
        if ring_len is None:
            ring_len = -(-mem.capacity // E) + pad
        self.replay = DeviceReplay(E, ring_len, H * W_, Wn, 1, self.A, B, True, False, float(kw.get("alpha", 0.0)), float(kw.get("beta_initial", 0.4)),
                                   int(kw.get("beta_steps", 1_000_000)), float(kw.get("epsilon", 1e-4)), mem.warmup_size, self.seed, device)
        self.L = self.replay.L
        if env is None:
            self.env = SyntheticAtariVecEnv(self.replay, episode_len)
        else:
            self.env = env(self.replay) if callable(env) else env
        c._set_device(str(self.dev))
        if parameter is None:
            parameter = c.make_parameter()
        parameter.to_device(self.dev)
        self.parameter = p = parameter
        self.learner = Agent57LightLearner(c, p, self.dev, channels_first=True)
        d = self.dev
        Na = c.actor_num
        self.beta_list, self.discount_list = self.learner.beta_list, self.learner.discount_list
        self.eps_list = torch.tensor(np.array(funcs.create_epsilon_list(Na), np.float32), device=d)
        self.actor_eye, self.action_eye = self.learner.actor_eye, self.learner.action_eye
        self.ucb = UcbBank(E, Na, c.ucb_window_size, c.ucb_epsilon, c.ucb_beta, d, self.seed)
        # The actors' five image blocks (two UVFA Q-networks, embedding, RND target / predictor) straight from the uint8 ring through libsrlx's
        # convolution kernels (device/qnet.py:ImageTrunk) -- MIOpen's fp32 convolutions + replication-pad + layout transposes were 60 % of a
        # lock-step; the dense parts and the whole learner stay torch.  SRLX_A57_TORCH_TRUNKS=1: the all-torch pass (A/B, tests).
        from simple_distributed_rl_amd.device.qnet import ImageTrunk

        self.overlap = bool(overlap)
        self._act_q = {"q_ext": p.q_ext_online, "q_int": p.q_int_online}  # the networks the actors evaluate
        if c.enable_intrinsic_reward:
            self._act_q.update(emb=p.emb_network, rnd_train=p.lifelong_train)
        if self.overlap:
            import copy

            # private copies of everything the actors evaluate AND the update trains (the RND target network is never trained: shared)
            self._act_src = dict(self._act_q)
            self._act_q = {k: copy.deepcopy(v).to(self.dev) for k, v in self._act_q.items()}
            for v in self._act_q.values():
                v.eval()
            self.s_learner = torch.cuda.Stream(device=self.dev, priority=-1)
            self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
            self._learner_pending = False
            self._launcher = _GraphLauncher(self.dev)
        self._trunks = {}
        nets = {"q_ext": self._act_q["q_ext"], "q_int": self._act_q["q_int"]}
        if c.enable_intrinsic_reward:
            nets.update(emb=self._act_q["emb"], rnd_target=p.lifelong_target, rnd_train=self._act_q["rnd_train"])
        if os.environ.get("SRLX_A57_TORCH_TRUNKS", "0") != "1":
            for name, net in nets.items():
                blk = getattr(getattr(net, "in_block", None), "image_block", None)
                if blk is not None and getattr(net.in_block, "out_flatten", False) and ImageTrunk.supported(blk):
                    self._trunks[name] = ImageTrunk(blk, shape[:2], E, device)
        self._all_fused = len(self._trunks) == len(nets)
        if self.overlap:
            assert self._all_fused, "Agent57LightEngine(overlap=True): every image block must run on the libsrlx trunks (the float32 stack kernel reads the ring position)"
            # the ring commit leaves the ring position to the PER add (which follows the join): everything between the two -- next state's features, intrinsic
            # reward, per-lane bookkeeping -- runs beside the update, off the frame-offset table the commit itself writes
            self.replay.enable_deferred_advance()
        # The LEARNER's image blocks, forward and backward, hand-written too (device/qnet.py:TrainableImageTrunk): one handle per network that trains
        # (its forward keeps the activations, its backward writes the six convolution gradients) and one per network that is only evaluated (targets).
        # No MIOpen on the update path: its convolutions were 3.3 ms of the update and chose their solvers per engine instance by timing, so that one
        # instance in four learned on a different trajectory.  SRLX_A57_TORCH_LEARNER=1: the torch image blocks (A/B, tests).
        self._ltrunks = None
        if self._all_fused and os.environ.get("SRLX_A57_TORCH_LEARNER", "0") != "1" and 2 * B <= 64 and nets["q_ext"].in_block.image_block.image_layers[0].out_channels == 32:
            from simple_distributed_rl_amd.device.qnet import TrainableImageTrunk

            blk = lambda net: net.in_block.image_block  # noqa: E731
            lt = {"q_ext": TrainableImageTrunk(blk(p.q_ext_online), shape[:2], 2 * B, device, B), "q_ext_t": ImageTrunk(blk(p.q_ext_target), shape[:2], B, device),
                  "q_int": TrainableImageTrunk(blk(p.q_int_online), shape[:2], 2 * B, device, B), "q_int_t": ImageTrunk(blk(p.q_int_target), shape[:2], B, device)}
            if c.enable_intrinsic_reward:
                lt.update(emb=TrainableImageTrunk(blk(p.emb_network), shape[:2], 2 * B, device, 2 * B), ll=TrainableImageTrunk(blk(p.lifelong_train), shape[:2], B, device, B),
                          ll_t=ImageTrunk(blk(p.lifelong_target), shape[:2], B, device))
            self._ltrunks = lt
        self.ngu = None
        if c.enable_intrinsic_reward:
            self.ngu = NguOps(d, E, p.emb_network.emb_block.out_size, c.episodic_memory_capacity, c.episodic_count_max, c.episodic_epsilon,
                              c.episodic_cluster_distance, c.episodic_pseudo_counts)
        # per-environment actor state (the reference keeps these on the worker object, :288-311)
        self.episode_reward = torch.zeros(E, dtype=torch.float32, device=d)
        self.prev_action = torch.zeros(E, dtype=torch.int64, device=d)
        self.prev_r_ext = torch.zeros(E, dtype=torch.float32, device=d)
        self.prev_r_int = torch.zeros(E, dtype=torch.float32, device=d)
        self.actions = torch.zeros(E, dtype=torch.int32, device=d)
        self.u_policy = torch.zeros(2 * E, dtype=torch.float64, device=d)
        self.policy_counter = torch.zeros(1, dtype=torch.int64, device=d)
        self.reset_lane = torch.zeros(E, dtype=torch.uint8, device=d)  # lanes whose lock-step only delivers a new episode's first frame
        self.gen = torch.Generator(device=d)
        self.gen.manual_seed(self.seed + 17)
        # the item fields the store does not keep, [ring slot][env]
        L = self.L
        self.x_r_int = torch.zeros((L, E), dtype=torch.float32, device=d)
        self.x_actor = torch.zeros((L, E), dtype=torch.int64, device=d)
        self.x_prev_action = torch.zeros((L, E), dtype=torch.int64, device=d)
        self.x_prev_r_ext = torch.zeros((L, E), dtype=torch.float32, device=d)
        self.x_prev_r_int = torch.zeros((L, E), dtype=torch.float32, device=d)
        self.train_count_dev = torch.zeros(1, dtype=torch.int64, device=d)
        self.loc_env = torch.zeros(B, dtype=torch.int64, device=d)
        self.loc_slot = torch.zeros(B, dtype=torch.int64, device=d)
        self.total_env_steps = 0
        self.ledger = None
        self._learner_graph = None
        self.training = True
        self.first_obs = self.env.reset()
        self.replay.reset_all(self.first_obs)
        self._begin_episodes(None)
        self.state = self._stack()

    # ---- helpers --------------------------------------------------------------------------------
    @property
    def train_count(self) -> int:
        return self.learner.train_count

    def _stack(self) -> Optional[torch.Tensor]:
        """The policy / intrinsic-reward input after a commit: the image features of every fused trunk (from the frame-offset table), and the
        float32 stack itself only when some network still needs it."""
        self._feat = {}
        if self._trunks:
            self._frame_off = self.replay.frame_table_current()  # (valid until the next commit)
            with torch.no_grad():
                for name, trunk in self._trunks.items():
                    if not name.startswith("q_"):  # embedding / RND: used right away; the Q-networks' at policy time (an update may come first)
                        self._feat[name] = trunk(self.replay.obs_base, self._frame_off)
        if self._all_fused:
            return None
        return self.replay.stack_current().view(self.E, self.Wn, *self.hw)

    def _begin_episodes(self, done: Optional[torch.Tensor]):
        """on_reset (:288-311) for the lanes in `done` (None = all): next arm from the lane's UCB controller, random previous action, zero
        previous rewards; the arm's (beta, epsilon, discount) are looked up when used."""
        self.ucb.step(done, self.episode_reward)
        rnd_a = torch.randint(0, self.A, (self.E,), device=self.dev, generator=self.gen)
        if done is None:
            self.prev_action.copy_(rnd_a)
            self.prev_r_ext.zero_()
            self.prev_r_int.zero_()
            self.episode_reward.zero_()
        else:
            m = done.bool()
            self.prev_action = torch.where(m, rnd_a, self.prev_action)
            self.prev_r_ext = torch.where(m, torch.zeros_like(self.prev_r_ext), self.prev_r_ext)
            self.prev_r_int = torch.where(m, torch.zeros_like(self.prev_r_int), self.prev_r_int)
            self.episode_reward = torch.where(m, torch.zeros_like(self.episode_reward), self.episode_reward)

    def arm(self) -> torch.Tensor:
        return self.ucb.arm.long() if self.training else torch.zeros(self.E, dtype=torch.int64, device=self.dev)

    # ---- actor ----------------------------------------------------------------------------------
    @_miopen_find
    def policy_q(self):
        """q_ext, q_int and q = q_ext + beta[arm] * q_int of every lane in its current state (:355-363)."""
        p, arm = self.parameter, self.arm()
        inputs = (self.state, self.prev_r_ext.view(-1, 1), self.prev_r_int.view(-1, 1), self.action_eye[self.prev_action], self.actor_eye[arm])
        with torch.no_grad():
            fe = self._trunks["q_ext"](self.replay.obs_base, self._frame_off) if "q_ext" in self._trunks else None
            q_ext = q_values(self._act_q["q_ext"], *inputs, features=fe)
            fi = self._trunks["q_int"](self.replay.obs_base, self._frame_off) if "q_int" in self._trunks else None
            q_int = q_values(self._act_q["q_int"], *inputs, features=fi)
        beta = self.beta_list[arm] if self.training else torch.full((self.E,), float(self.cfg.test_beta), device=self.dev)
        return q_ext, q_int, (q_ext + beta.view(-1, 1) * q_int).contiguous()

    @_miopen_find
    def actor_net(self):
        """First half of a lock-step: the two Q-networks over every lane's current state.  Reads the ring and the per-lane UVFA state only --
        none of the tensors a transition exchange in flight still holds (device/dist.py) -- so it may run while that exchange travels."""
        arm = self.arm()
        _, _, q = self.policy_q()
        return arm, q

    def actor_step(self):
        self.actor_rest(*self.actor_net())

    @_miopen_find
    def actor_rest(self, arm, q):
        """Second half: action selection, environments, ring commit, intrinsic reward, per-lane bookkeeping (overwrites actions / rewards / flags /
        next_obs: an exchange that still reads them must have been waited for)."""
        c, r, st = self.cfg, self.replay, N.torch_stream_ptr()
        E = self.E
        eps = (self.eps_list[arm] if self.training else torch.full((E,), float(c.test_epsilon), device=self.dev)).contiguous()
        N.check(self.lib.srlx_rng_uniform(self.seed ^ 0xAC7, N.tptr(self.policy_counter), self.u_policy.numel(), N.tptr(self.u_policy), st))
        N.check(self.lib.srlx_policy_epsilon_greedy(E, self.A, N.tptr(q), N.tptr(eps), N.tptr(self.u_policy), None, N.tptr(self.actions), st))
        next_obs, rewards, terminated, done = self.env.step(self.actions)
        slot = r._steps_committed % self.L  # the ring slot this lock-step's transition goes to
        self.x_actor[slot] = arm
        self.x_prev_action[slot] = self.prev_action
        self.x_prev_r_ext[slot] = self.prev_r_ext
        self.x_prev_r_int[slot] = self.prev_r_int
        if self.ledger is not None:
            self.ledger.account(rewards, done, r.needs_reset_ptr)
        if self.overlap:  # ring commit now (slot p + 1 and the scalars of p belong to no stored item), tree add + ring position after the join at the end
            r.commit(self.actions, rewards, terminated, done, next_obs, defer_add=True, next_table=True)
        else:
            r.commit(self.actions, rewards, terminated, done, next_obs)
        self.state = self._stack()  # s_{t+1}: the next lock-step's policy input AND the intrinsic reward's argument
        live = (self.reset_lane == 0)
        r_int = torch.zeros(E, dtype=torch.float32, device=self.dev)
        if c.enable_intrinsic_reward:
            p = self.parameter
            with torch.no_grad():
                emb_net, rnd_train = self._act_q["emb"], self._act_q["rnd_train"]
                emb_net.eval()
                rnd_train.eval()
                f = self._feat
                episodic = self.ngu.episodic(embed(emb_net, self.state, f.get("emb")), reset=self.reset_lane, active=live.to(torch.uint8))
                lifelong = self.ngu.lifelong(rnd(p.lifelong_target, self.state, f.get("rnd_target")), rnd(rnd_train, self.state, f.get("rnd_train")),
                                             c.lifelong_max)
            r_int = torch.where(live, episodic * lifelong, r_int)  # :383-391
        self.x_r_int[slot] = r_int
        # the worker's bookkeeping (:393-417): lanes in a reset lock-step took no action
        self.prev_action = torch.where(live, self.actions.long(), self.prev_action)
        self.prev_r_ext = torch.where(live, rewards, self.prev_r_ext)
        self.prev_r_int = torch.where(live, r_int, self.prev_r_int)
        self.episode_reward = self.episode_reward + torch.where(live, rewards, torch.zeros_like(rewards))
        if self.training:
            self._begin_episodes(done)  # lanes whose episode just ended: book it with their UCB controller, draw the next arm
        self.reset_lane = done.clone()
        self.total_env_steps += E
        if self.overlap:  # the forked updates read the ring position and write the tree: they finish before the add (which moves the position)
            self.join_learner()
            r.add_masked()

    # ---- learner --------------------------------------------------------------------------------
    def learner_features(self, rp):
        """The image blocks of one update through the hand-written trunks, straight from `rp`'s uint8 ring through the frame-offset table of the batch
        just drawn (`rp.sample_items(all_states=True)`: rows [b][s_0, s_1][window]).  An online Q-network evaluates s_0 (with gradient) and s_1
        (without: the double-DQN argmax, agent57_light.py:241-257) in ONE pass of 2 B rows whose backward touches rows 0, 2, 4, ... only."""
        lt, B, W = self._ltrunks, rp.B, self.Wn
        base = rp.obs_base
        off01 = rp.frame_off_all.view(2 * B, W)
        off0, off1 = rp.frame_off_all[:, 0].contiguous(), rp.frame_off_all[:, 1].contiguous()
        ft = {}
        for name in ("q_ext", "q_int"):
            x = lt[name].features(base, off01, 2)
            ft[name] = (x[0::2], x[1::2].detach(), lt[name + "_t"](base, off1))
        if "emb" in lt:
            x = lt["emb"].features(base, off01, 1)
            ft["emb"] = (x[0::2], x[1::2])
            ft["ll"] = (lt["ll"].features(base, off0, 1), lt["ll_t"](base, off0))
        return ft

    def _learner_body(self):
        """PER sample -> gather (frames by the store, the other fields by (slot, env)) -> the five networks' update -> PER update: device work only."""
        r = self.replay
        hand = self._ltrunks is not None
        b = r.sample_items(self.train_count_dev, all_states=True) if hand else r.sample(self.train_count_dev)
        N.check(self.lib.srlx_store_locate(r.h_store, r.B, N.tptr(b.indices), N.tptr(self.loc_env), N.tptr(self.loc_slot), None, N.torch_stream_ptr()))
        e, s = self.loc_env, self.loc_slot
        obs = None if hand else b.obs.view(r.B, 2, self.Wn, *self.hw)
        pri = self.learner.update_networks(None if hand else obs[:, 0], None if hand else obs[:, 1], b.actions.view(-1), b.rewards.view(-1), self.x_r_int[s, e],
                                           1.0 - b.terminated.view(-1), self.x_prev_action[s, e], self.x_prev_r_ext[s, e], self.x_prev_r_int[s, e], self.x_actor[s, e],
                                           b.weights, features=self.learner_features(r) if hand else None)
        r.update(b.indices, pri)
        self.train_count_dev.add_(1)

    def _learner_step_raw(self) -> bool:
        if self.replay.is_warmup_needed():
            return False
        if self._learner_graph is not None:
            self._learner_graph.replay()
        else:
            self._learner_body()
        self.learner.after_update()
        return True

    @_miopen_find
    def learner_step(self) -> bool:
        return self._learner_step_raw()

    def fork_learner(self, updates: int):
        """overlap: `updates` updates on the learner's stream, ordered after everything enqueued on the current stream so far, launched by the helper thread."""
        if updates <= 0 or self.replay.is_warmup_needed():
            return 0
        main = torch.cuda.current_stream(self.dev)
        self._ev_fork.record(main)

        def body():
            self.s_learner.wait_event(self._ev_fork)
            with torch.cuda.stream(self.s_learner):
                for _ in range(updates):
                    self._learner_step_raw()
                self._ev_join.record(self.s_learner)

        self._learner_pending = True
        if self._learner_graph is not None:
            self._launcher.submit(body)
        else:  # (eager updates call MIOpen-free torch ops from this thread)
            body()
        return updates  # (as RainbowEngine.fork_learner: how many updates were enqueued)

    def close(self):
        """Joins a pending update and ends the launcher thread."""
        if self.overlap and getattr(self, "_launcher", None) is not None:
            self.join_learner()
            torch.cuda.synchronize(self.dev)
            self._launcher.close()
            self._launcher = None

    @_miopen_find
    def capture_graphs(self, warm_updates: int = 3):
        """The whole update (sampling, gathers, four forward/backward passes, four fused Adam steps, priority write-back) as ONE HIP graph:
        eager it is ~400 small launches and the host is the bottleneck (7.9 ms per update at B = 32).  Call once the replay is warm."""
        if self._learner_graph is not None or self.replay.is_warmup_needed():
            return
        torch.cuda.synchronize(self.dev)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # warm-up on a side stream: MIOpen picks its solvers, the optimisers create their state
            for _ in range(warm_updates):
                self._learner_body()
                self.learner.after_update()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._learner_body()
        self._learner_graph = g
        torch.cuda.synchronize(self.dev)

    def step(self, learner_updates: int = 1, events=None):
        if self.overlap:
            self.fork_learner(learner_updates)
        if events is not None:
            events[0].record()
        self.actor_step()  # (joins the forked updates before its ring commit)
        if events is not None:
            events[1].record()
        if self.overlap:
            self.refresh_actor_copy()
        else:
            for _ in range(learner_updates):
                self.learner_step()

    def join_learner(self):
        """The current stream waits for the forked updates (overlap; otherwise updates run on the caller's stream and there is nothing to join)."""
        if self.overlap and self._learner_pending:
            self._launcher.wait()
            torch.cuda.current_stream(self.dev).wait_event(self._ev_join)
            self._learner_pending = False

    def refresh_actor_copy(self):
        """overlap: the actors' private Q-networks := the online ones (one multi-tensor copy each; call with the learner joined)."""
        if self.overlap:
            with torch.no_grad():
                for k, src in self._act_src.items():
                    torch._foreach_copy_(list(self._act_q[k].parameters()), list(src.parameters()))
                    if list(src.buffers()):
                        torch._foreach_copy_(list(self._act_q[k].buffers()), list(src.buffers()))

    def prefill(self, randomise_priorities: bool = True):
        """Random-policy rollout until every PER leaf holds an item (untimed benchmark set-up)."""
        saved = self.eps_list
        self.eps_list = torch.ones_like(saved)
        steps = self.replay.item_len
        for _ in range(steps):
            self.actor_step()
        self.eps_list = saved
        if randomise_priorities:
            g = torch.Generator(device=self.dev)
            g.manual_seed(self.seed + 1)
            pri = torch.rand(self.replay.capacity, dtype=torch.float32, device=self.dev, generator=g)
            N.check(self.lib.srlx_per_set_range(self.replay.h_per, 0, self.replay.capacity, N.tptr(pri), N.PRIO_F32, 1, N.torch_stream_ptr()))
        torch.cuda.synchronize(self.dev)

    def info(self) -> dict:
        from simple_distributed_rl_amd.device.qnet import check_ranges

        check_ranges()
        d = dict(train_count=self.train_count, memory=self.replay.length())
        if self.train_count > 0:
            d.update(self.learner.losses())
            d["loss"] = d["ext_loss"]
        return d
