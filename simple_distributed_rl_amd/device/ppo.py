"""PPO on the vectorised path (BASELINE.json config 5: "PPO continuous (Pendulum-shaped obs), 4096
vectorized envs, fused GAE kernel, data-parallel").

E lock-stepped environments live on the GPU.  A rollout of T steps costs 3 launches per step around the network
(`srlx_ppo_normal_act` = sample + log-prob, `srlx_pendulum_step` = environment + auto-reset, buffer writes are slices
of preallocated [T][E] tensors); the advantages of all E x T transitions come from ONE `srlx_gae_scan`; every
minibatch update is forward -> ONE `srlx_ppo_loss_normal` (losses + gradient seeds for loc / log_scale / v) ->
backward -> global-norm clip -> Adam.  Nothing crosses to the host inside an iteration.

Reference semantics kept (srl/algorithms/ppo/ppo.py): the Normal head with the stable-gradient log-scale clip,
log-prob floor log(1e-6) (:322), GAE with no bootstrap at an episode end, truncation included (:389-404), the SAME
GAE value used as v_target and as advantage with `baseline_type="advantage"` subtracting V again (:214-215,121-122)
-- `v_target="return"` selects the textbook target (advantage + V) instead --, clipped surrogate, value clipping,
entropy bonus on the taken action's log-prob (:166), global gradient clipping (:240-241).

Round 6: with the reference's default blocks (hidden (64, 64), value (64,), policy (64,)) the network itself is libsrlx code too (`fused`, the default then;
csrc/srlx_ppo_net.hip): the WHOLE rollout of an iteration -- T network passes, policy samples, environment steps, the buffers, V(s_T), the GAE scan -- is one
launch, a minibatch update is three (forward + loss + backward; gradient reduction; clip + Adam), and an iteration is ~55 launches instead of ~1700 framework
kernels.  The parameters are one flat float32 vector; the `ActorCritic` module's tensors are views of it.  `fused=False` keeps the torch-autograd path as the
yardstick the fused one is tested against.

Data parallel (config 5): `DistributedPPO` gives every rank its own E environments and averages the gradients of
every minibatch with one all-reduce of the flat ~52 KB gradient vector (latency-bound; RCCL over xGMI) -- the only exchange; with the fused network and RCCL
it sits INSIDE the captured update graph, between the gradient reduction and the clip + Adam launch.
"""
import ctypes
import math
import os
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.nn as nn

from simple_distributed_rl_amd import _native as N


@dataclass
class PPODeviceConfig:
    n_envs: int = 4096
    horizon: int = 32
    epochs: int = 4
    minibatches: int = 4
    episode_len: int = 200
    obs_dim: int = 3
    action_dim: int = 1
    hidden_sizes: Tuple[int, ...] = (64, 64)   # hidden_block (config.py:47)
    value_sizes: Tuple[int, ...] = (64,)       # value_block
    policy_sizes: Tuple[int, ...] = (64,)      # policy_block
    discount: float = 0.9
    gae_discount: float = 0.9
    baseline_type: str = "advantage"
    v_target: str = "gae"                      # "gae" = the reference's target (ppo.py:214), "return" = gae + V
    surrogate_type: str = "clip"
    policy_clip_range: float = 0.2
    enable_value_clip: bool = True
    value_clip_range: float = 0.2
    lr: float = 0.0002
    value_loss_weight: float = 1.0
    entropy_weight: float = 0.01
    global_gradient_clip_norm: float = 0.5
    stable_gradients_scale_range: Tuple[float, float] = (1e-10, 10)
    seed: int = 0


class ActorCritic(nn.Module):
    """in -> hidden_block -> {value_block -> V, policy_block -> (loc, log_scale)} (ppo.py:55-99)."""

    def __init__(self, cfg: PPODeviceConfig):
        super().__init__()

        def mlp(n_in, sizes):
            layers, n = [], n_in
            for s in sizes:
                layers += [nn.Linear(n, s), nn.ReLU()]
                n = s
            return nn.Sequential(*layers), n

        self.hidden_block, n = mlp(cfg.obs_dim, cfg.hidden_sizes)
        self.value_block, nv = mlp(n, cfg.value_sizes)
        self.value_out_layer = nn.Linear(nv, 1)
        self.policy_block, n_pol = mlp(n, cfg.policy_sizes)
        self.loc_layer = nn.Linear(n_pol, cfg.action_dim)
        self.log_scale_layer = nn.Linear(n_pol, cfg.action_dim)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight)  # ppo.py:60-61
                nn.init.zeros_(m.bias)
        nn.init.trunc_normal_(self.loc_layer.bias, std=0.05)  # normal_dist_block.py:101-106

    def forward(self, x):
        h = self.hidden_block(x)
        p = self.policy_block(h)
        return self.value_out_layer(self.value_block(h)).squeeze(-1), self.loc_layer(p), self.log_scale_layer(p)


class PendulumVecEnv:
    """E Pendulum-shaped environments stepped by `srlx_pendulum_step` (state stays on the device)."""

    def __init__(self, n_envs: int, episode_len: int, seed: int, device: torch.device):
        self.E, self.episode_len, self.seed, self.dev, self.lib = n_envs, episode_len, seed, device, N.lib()
        g = torch.Generator(device="cpu").manual_seed(seed)
        th = (torch.rand(n_envs, generator=g) * 2 - 1) * math.pi
        thd = torch.rand(n_envs, generator=g) * 2 - 1
        self.state = torch.stack([th, thd], dim=1).to(device).contiguous()
        self.t = torch.zeros(n_envs, dtype=torch.int32, device=device)
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.obs = torch.stack([torch.cos(self.state[:, 0]), torch.sin(self.state[:, 0]), self.state[:, 1]], dim=1).contiguous()

    def step(self, action: torch.Tensor, obs_out: torch.Tensor, reward_out: torch.Tensor, done_out: torch.Tensor):
        N.check(self.lib.srlx_pendulum_step(self.E, N.tptr(self.state), N.tptr(self.t), N.tptr(action), self.episode_len, self.seed, N.tptr(self.counter),
                                            N.tptr(obs_out), N.tptr(reward_out), N.tptr(done_out), N.torch_stream_ptr()))


class PPOEngine:
    def __init__(self, cfg: PPODeviceConfig, device: int = 0, grad_sync: Optional[Callable[[nn.Module], None]] = None, fused: Optional[bool] = None,
                 flat_grad_sync: Optional[Callable[[torch.Tensor], float]] = None):
        """fused: the network in libsrlx (None: whenever the geometry is the reference's default blocks; True: required; False: torch modules + autograd, the test
        yardstick).  grad_sync(module): the torch path's gradient exchange; flat_grad_sync(flat_grad) -> scale: the fused path's (all-reduces the flat gradient in
        place, returns the factor the optimiser launch applies: 1 / world size)."""
        if not torch.cuda.is_available():
            raise RuntimeError("simple_distributed_rl_amd.device.ppo needs an MI355X: its rollout / GAE / loss arithmetic is libsrlx HIP code (no CPU fallback)")
        if cfg.surrogate_type not in ("clip", ""):
            raise ValueError('surrogate_type must be "clip" or "" (the reference\'s "kl" needs tensorflow_probability, functions.py:95-103)')
        self.cfg, self.lib = cfg, N.lib()
        self.dev = torch.device(f"cuda:{device}")
        torch.manual_seed(cfg.seed)
        self.net = ActorCritic(cfg).to(self.dev)
        can_fuse = (tuple(cfg.hidden_sizes) == (64, 64) and tuple(cfg.value_sizes) == (64,) and tuple(cfg.policy_sizes) == (64,) and 1 <= cfg.obs_dim <= 8
                    and 1 <= cfg.action_dim <= 4)
        if fused and not can_fuse:
            raise ValueError("PPOEngine(fused=True): the libsrlx network covers hidden (64, 64), value (64,), policy (64,), obs_dim <= 8, action_dim <= 4")
        self.fused = can_fuse if fused is None else bool(fused)
        self.grad_sync, self.flat_grad_sync = grad_sync, flat_grad_sync
        if self.fused:
            # one flat parameter vector in `parameters()` order; the module's tensors become views of it (state_dict / export keep working, always current)
            P = self.lib.srlx_ppo_net_param_count(cfg.obs_dim, cfg.action_dim)
            ps = list(self.net.parameters())
            assert sum(p.numel() for p in ps) == P
            self.flat = torch.cat([p.detach().reshape(-1) for p in ps]).contiguous()
            off = 0
            for p in ps:
                p.data = self.flat[off : off + p.numel()].view_as(p)
                off += p.numel()
            self.flat_grad = torch.zeros(P, dtype=torch.float32, device=self.dev)
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
            self.opt_step = torch.zeros(2, dtype=torch.int64, device=self.dev)  # [steps taken, the optimiser launch's arrival counter]
            self.partials = torch.zeros(self.lib.srlx_ppo_net_partials_floats(cfg.obs_dim, cfg.action_dim), dtype=torch.float32, device=self.dev)
            self.opt = None
        else:
            self.opt = torch.optim.Adam(self.net.parameters(), lr=cfg.lr, capturable=True)
        self.env = PendulumVecEnv(cfg.n_envs, cfg.episode_len, cfg.seed, self.dev)
        self.ls_range = (math.log(cfg.stable_gradients_scale_range[0]), math.log(cfg.stable_gradients_scale_range[1]))
        E, T, A, d = cfg.n_envs, cfg.horizon, cfg.action_dim, self.dev
        f32 = dict(dtype=torch.float32, device=d)
        self.b_obs = torch.zeros((T + 1, E, cfg.obs_dim), **f32)
        self.b_act = torch.zeros((T, E, A), **f32)
        self.b_logp = torch.zeros((T, E, A), **f32)
        self.b_val = torch.zeros((T, E), **f32)
        self.b_rew = torch.zeros((T, E), **f32)
        self.b_done = torch.zeros((T, E), dtype=torch.uint8, device=d)
        self.b_adv = torch.zeros((T, E), **f32)
        self.act_counter = torch.zeros(1, dtype=torch.int64, device=d)
        self.losses = torch.zeros(3, **f32)
        self.b_obs[0].copy_(self.env.obs)
        self.iterations = 0
        self._rollout_graph = None
        self._update_graph = None
        self._last_v = torch.zeros(E, **f32)
        self.episode_return = torch.zeros(E, **f32)
        self.finished_returns = torch.zeros(2, **f32)  # sum, count of finished episodes since the last read
        # One minibatch permutation per epoch from libsrlx's keyed permutation kernel (srlx_rng_permutation: device state only), INSIDE the captured
        # update.  What round 2 hid behind a hipStreamSynchronize per iteration (tools/ppo_replay_bisect.py, ROCm 7.2 / torch 2.10, E = 4096): with NO
        # eager launch between replays of the two large graphs (~700 / ~1000 nodes) -- fixed permutations, or this kernel -- 40 unsynchronised iterations
        # equal the synchronised run to 1e-7; with torch.randperm as a graph node (torch refreshes the generator's offset tensors with eager launches before
        # every replay) they turn non-finite, and with torch.randperm drawn eagerly between the replays they stay finite but train differently: eager
        # kernels enqueued behind a large graph launch do not reliably wait for the graph's tail (an event recorded there does not either:
        # event.synchronize() per iteration does not help, hipStreamSynchronize does).  A graph whose only node is randperm replays fine
        # (tools/randperm_graph_repro.py).  SRLX_PPO_PERM = in_graph (torch.randperm as a node) / eager / fixed: the bisect's other arms.
        self._perm_mode = os.environ.get("SRLX_PPO_PERM", "srlx")
        self._perms = torch.stack([torch.randperm(T * E, device=d) for _ in range(cfg.epochs)])  # ("fixed": these stay)
        self.perm_counter = torch.zeros(1, dtype=torch.int64, device=d)

    # --- rollout ---------------------------------------------------------------------------------------------------
    def act(self, obs: torch.Tensor, action_out: torch.Tensor, logp_out: torch.Tensor, deterministic: bool = False):
        v, loc, ls = self.forward(obs)
        self._keep = (loc, ls)
        N.check(self.lib.srlx_ppo_normal_act(loc.numel(), N.tptr(loc), N.tptr(ls), self.ls_range[0], self.ls_range[1], self.cfg.seed ^ 0x61637400,
                                             N.tptr(self.act_counter), int(deterministic), N.tptr(action_out), N.tptr(logp_out), N.torch_stream_ptr()))
        return v

    def forward(self, obs: torch.Tensor):
        """(v [n], loc [n][A], log_scale [n][A]) of obs [n][obs_dim]: the libsrlx network when fused, the torch modules otherwise."""
        if not self.fused:
            with torch.no_grad():
                return self.net(obs)
        n, A = obs.shape[0], self.cfg.action_dim
        v, loc, ls = (torch.empty(n, dtype=torch.float32, device=self.dev), torch.empty((n, A), dtype=torch.float32, device=self.dev),
                      torch.empty((n, A), dtype=torch.float32, device=self.dev))
        N.check(self.lib.srlx_ppo_net_forward(n, self.cfg.obs_dim, A, N.tptr(self.flat), N.tptr(obs.contiguous()), N.tptr(v), N.tptr(loc), N.tptr(ls), N.torch_stream_ptr()))
        return v, loc, ls

    def _fused_rollout_ok(self) -> bool:
        return (self.fused and isinstance(self.env, PendulumVecEnv) and self.cfg.obs_dim == 3 and self.cfg.n_envs % 16 == 0
                and self.cfg.horizon <= self.lib.srlx_ppo_net_rollout_max_horizon(self.cfg.action_dim))  # (longer horizons: the step-wise kernels)

    def rollout(self):
        cfg = self.cfg
        if self._fused_rollout_ok():  # T steps of everything in ONE launch (csrc/srlx_ppo_net.hip: k_ppo_rollout)
            env = self.env
            N.check(self.lib.srlx_ppo_net_rollout(cfg.n_envs, cfg.horizon, cfg.action_dim, N.tptr(self.flat), N.tptr(env.state), N.tptr(env.t), N.tptr(env.obs), env.episode_len,
                                                  env.seed, N.tptr(env.counter), cfg.seed ^ 0x61637400, N.tptr(self.act_counter), self.ls_range[0], self.ls_range[1],
                                                  cfg.discount, cfg.gae_discount, N.tptr(self.b_obs), N.tptr(self.b_act), N.tptr(self.b_logp), N.tptr(self.b_val),
                                                  N.tptr(self.b_rew), N.tptr(self.b_done), N.tptr(self.b_adv), N.tptr(self._last_v), N.tptr(self.episode_return),
                                                  N.tptr(self.finished_returns), N.torch_stream_ptr()))
            return
        for t in range(cfg.horizon):
            self.b_val[t].copy_(self.act(self.b_obs[t], self.b_act[t], self.b_logp[t]))
            self.env.step(self.b_act[t, :, 0].contiguous() if cfg.action_dim > 1 else self.b_act[t].view(-1), self.b_obs[t + 1], self.b_rew[t], self.b_done[t])
            self.episode_return += self.b_rew[t]
            d = self.b_done[t].bool()
            self.finished_returns[0] += (self.episode_return * d).sum()
            self.finished_returns[1] += d.sum()
            self.episode_return.masked_fill_(d, 0.0)
        last_v, _, _ = self.forward(self.b_obs[cfg.horizon])
        # episode ends are never bootstrapped (ppo.py:396-397); a horizon cut inside an episode bootstraps from V(s_T)
        N.check(self.lib.srlx_gae_scan(cfg.n_envs, cfg.horizon, N.tptr(self.b_rew), N.tptr(self.b_val), N.tptr(self.b_done), N.tptr(last_v.contiguous()),
                                       cfg.discount, cfg.gae_discount, N.tptr(self.b_adv), N.torch_stream_ptr()))
        self._last_v = last_v

    # --- update ----------------------------------------------------------------------------------------------------
    def loss_and_seeds(self, obs, action, old_logp, adv, v_target, old_v):
        """forward + the fused loss kernel; returns (v, loc, log_scale) with their gradient seeds."""
        cfg = self.cfg
        v, loc, ls = self.net(obs)
        B, A = loc.shape
        g_loc, g_ls, g_v = torch.empty_like(loc), torch.empty_like(ls), torch.empty_like(v)
        N.check(self.lib.srlx_ppo_loss_normal(
            B, A, N.tptr(loc.detach()), N.tptr(ls.detach()), self.ls_range[0], self.ls_range[1], N.tptr(action), N.tptr(old_logp), N.tptr(adv), N.tptr(v.detach()),
            N.tptr(v_target), N.tptr(old_v), int(cfg.baseline_type == "advantage"), int(cfg.surrogate_type == "clip"), cfg.policy_clip_range,
            int(cfg.enable_value_clip), cfg.value_clip_range, cfg.value_loss_weight, cfg.entropy_weight, N.tptr(self.losses), N.tptr(g_loc), N.tptr(g_ls),
            N.tptr(g_v), N.torch_stream_ptr()))
        return (v, loc, ls), (g_v, g_loc, g_ls)

    def update(self):
        cfg = self.cfg
        T, E = cfg.horizon, cfg.n_envs
        n = T * E
        obs = self.b_obs[:T].reshape(n, cfg.obs_dim)
        act = self.b_act.reshape(n, cfg.action_dim)
        logp = self.b_logp.reshape(n, cfg.action_dim)
        adv = self.b_adv.reshape(n)
        val = self.b_val.reshape(n)
        v_target = adv if cfg.v_target == "gae" else adv + val
        mb = n // cfg.minibatches
        if self.fused:
            return self._update_fused(n, mb, obs, act, logp, adv, val, v_target)
        for ep in range(cfg.epochs):
            if self._perm_mode == "srlx":
                N.check(self.lib.srlx_rng_permutation(cfg.seed ^ 0x7065726D, N.tptr(self.perm_counter), n, N.tptr(self._perms[ep]), N.torch_stream_ptr()))
            perm = self._perms[ep] if self._perm_mode != "in_graph" else torch.randperm(n, device=self.dev)
            for k in range(cfg.minibatches):
                idx = perm[k * mb : (k + 1) * mb]
                outs, seeds = self.loss_and_seeds(obs[idx], act[idx].contiguous(), logp[idx].contiguous(), adv[idx].contiguous(), v_target[idx].contiguous(),
                                                  val[idx].contiguous())
                self.opt.zero_grad(set_to_none=False)
                torch.autograd.backward(outs, seeds)
                if self.grad_sync is not None:
                    self.grad_sync(self.net)
                if cfg.global_gradient_clip_norm != 0:
                    torch.nn.utils.clip_grad_norm_(self.net.parameters(), cfg.global_gradient_clip_norm)
                self.opt.step()

    def _update_fused(self, n, mb, obs, act, logp, adv, val, v_target):
        """epochs x minibatches of (k_ppo_minibatch + k_ppo_reduce) -> [all-reduce of the flat gradient] -> k_ppo_adam; the buffers are read in place through the
        permutation's rows."""
        cfg = self.cfg
        st = N.torch_stream_ptr()
        if self._perm_mode == "srlx":  # the epochs' shuffles in one launch (the values `epochs` successive srlx_rng_permutation calls would write)
            N.check(self.lib.srlx_rng_permutations(cfg.seed ^ 0x7065726D, N.tptr(self.perm_counter), n, cfg.epochs, N.tptr(self._perms), st))
        for ep in range(cfg.epochs):
            for k in range(cfg.minibatches):
                rows = self._perms[ep][k * mb : (k + 1) * mb]
                N.check(self.lib.srlx_ppo_net_minibatch(mb, N.tptr(rows), cfg.obs_dim, cfg.action_dim, N.tptr(self.flat), N.tptr(obs), N.tptr(act), N.tptr(logp), N.tptr(adv),
                                                        N.tptr(v_target), N.tptr(val), self.ls_range[0], self.ls_range[1], int(cfg.baseline_type == "advantage"),
                                                        int(cfg.surrogate_type == "clip"), cfg.policy_clip_range, int(cfg.enable_value_clip), cfg.value_clip_range,
                                                        cfg.value_loss_weight, cfg.entropy_weight, N.tptr(self.partials), N.tptr(self.flat_grad), N.tptr(self.losses), st))
                scale = self.flat_grad_sync(self.flat_grad) if self.flat_grad_sync is not None else 1.0
                N.check(self.lib.srlx_ppo_net_adam(cfg.obs_dim, cfg.action_dim, N.tptr(self.flat), N.tptr(self.flat_grad), N.tptr(self.exp_avg), N.tptr(self.exp_avg_sq),
                                                   N.tptr(self.opt_step), cfg.lr, 0.9, 0.999, 1e-8, cfg.global_gradient_clip_norm, scale, st))

    def capture_graphs(self):
        """Captures the T-step rollout (+ GAE) and the whole update phase into two HIP graphs: an iteration becomes two
        graph launches instead of ~T*20 + epochs*minibatches*60 eager ones.  Call after a few eager iterations (Adam
        state and every scratch buffer must exist)."""
        torch.cuda.synchronize(self.dev)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # warm-up on a capture-style stream
            self.rollout()
            self.update()
            self.b_obs[0].copy_(self.b_obs[self.cfg.horizon])
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            self.rollout()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            self.update()
            self.b_obs[0].copy_(self.b_obs[self.cfg.horizon])
        self._rollout_graph, self._update_graph = g1, g2
        torch.cuda.synchronize(self.dev)

    def _draw_permutations(self):
        if self._perm_mode == "eager":
            for ep in range(self.cfg.epochs):
                torch.randperm(self._perms.shape[1], device=self.dev, out=self._perms[ep])

    def step(self):
        """one PPO iteration: T x E environment steps + epochs x minibatches updates"""
        self._draw_permutations()
        if self._rollout_graph is not None:
            # two graph launches per iteration, nothing waits on the host and nothing is launched eagerly between them (see __init__)
            self._rollout_graph.replay()
            self._update_graph.replay()
            mode = os.environ.get("SRLX_PPO_SYNC", "none")  # tools/ppo_replay_bisect.py: host-side waits as an experiment
            if mode == "stream":
                torch.cuda.current_stream(self.dev).synchronize()
            elif mode == "event":
                ev = torch.cuda.Event()
                ev.record()
                ev.synchronize()
            elif mode.startswith("every"):
                if (self.iterations + 1) % int(mode[5:]) == 0:
                    torch.cuda.current_stream(self.dev).synchronize()
        else:
            self.rollout()
            self.update()
            self.b_obs[0].copy_(self.b_obs[self.cfg.horizon])
        self.iterations += 1

    def pop_mean_episode_return(self) -> float:
        s, c = self.finished_returns.tolist()
        self.finished_returns.zero_()
        return s / c if c else float("nan")

    def info(self) -> dict:
        pl, vl, el = self.losses.tolist()
        return dict(policy_loss=pl, value_loss=vl, entropy_loss=el)


def flat_grad_all_reduce(net: nn.Module, group=None):
    """Average the gradients of every rank: ONE all-reduce of a flat buffer (about 40 KB for the config-5 network)."""
    import torch.distributed as dist

    grads = [p.grad for p in net.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    staged = dist.get_backend(group) == "gloo" and flat.is_cuda  # test rigs: ranks sharing one GPU
    buf = flat.cpu() if staged else flat
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    buf = buf.to(flat.device) / dist.get_world_size(group)
    off = 0
    for g in grads:
        g.copy_(buf[off : off + g.numel()].view_as(g))
        off += g.numel()


def flat_vector_all_reduce(flat: torch.Tensor, group=None) -> float:
    """The fused network's exchange: ONE in-place all-reduce (sum) of the flat gradient vector; returns 1 / world size, which the clip + Adam launch applies.  With
    RCCL the collective is captured into the update graph like any other node; gloo (test rigs: ranks sharing one GPU) stages through the host and runs eagerly."""
    import torch.distributed as dist

    if dist.get_backend(group) == "gloo" and flat.is_cuda:
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(h)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


class DistributedPPO:
    """Data-parallel PPO (BASELINE config 5): identical networks, disjoint environments, averaged gradients."""

    def __init__(self, cfg: PPODeviceConfig, device: int, fused: Optional[bool] = None):
        import dataclasses

        import torch.distributed as dist

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        local = dataclasses.replace(cfg, seed=cfg.seed)  # same seed -> same initial network on every rank
        self.engine = PPOEngine(local, device, grad_sync=flat_grad_all_reduce, flat_grad_sync=flat_vector_all_reduce, fused=fused)
        # decorrelate environments and sampling noise across ranks
        self.engine.env = PendulumVecEnv(cfg.n_envs, cfg.episode_len, cfg.seed + 7919 * (self.rank + 1), self.engine.dev)
        self.engine.b_obs[0].copy_(self.engine.env.obs)
        self.engine.act_counter.fill_(self.rank << 40)
        tensors = [self.engine.flat] if self.engine.fused else [p.data for p in self.engine.net.parameters()]
        for t in tensors:  # belt and braces: one broadcast of the initial parameters
            if dist.get_backend() == "gloo" and t.is_cuda:
                h = t.cpu()
                dist.broadcast(h, src=0)
                t.copy_(h)
            else:
                dist.broadcast(t, src=0)

    def capture_graphs(self):
        """The rollout and the update as HIP graphs; with the fused network over RCCL the update's graph holds its 16 all-reduces of the flat gradient (every rank
        replays the same graph, so the collectives stay matched).  Other set-ups keep the update eager (a host-staged gloo all-reduce cannot be captured)."""
        import torch.distributed as dist

        if self.engine.fused and dist.get_backend() == "nccl":
            self.engine.capture_graphs()

    def step(self):
        self.engine.step()
