"""PPO plugin for discrete and continuous action spaces (srl/algorithms/ppo/config.py:31-128,
srl/algorithms/ppo/ppo.py:28-404), registered as "PPO:torch".

The reference's PPO is a TensorFlow/Keras model (`get_framework() == "tensorflow"`, ppo.py:6-7) and cannot be imported
in the build container, so this module is a restatement of the cited lines on the torch/ROCm stack, NOT pinned against
recorded reference outputs (parity UNPINNED; the loss arithmetic is checked against oracle/hot_path_oracle.py:ppo_loss,
which restates ppo.py:102-169, and the GAE scan against the oracle's restatement of ppo.py:389-404).

Worker: the reference's host logic -- one environment, per-step dict items with the taken action's log-probability
floored at log(1e-6) (:307), and at the end of the episode the GAE reverse scan whose result is stored as BOTH the
value target and the advantage (:389-404, :214-215), here one `srlx_gae_scan` launch over the episode.
Trainer: `train_num` minibatch updates per call (:191-201); the clipped surrogate, clipped value loss and entropy
bonus with their gradient seeds come from one `srlx_ppo_loss_logpi` launch, torch only back-propagates the seeds
through the small actor-critic MLP; global-norm clipping and Adam with the staircase schedule as configured.
Continuous (NpArraySpace) actions: a Normal policy head (loc, log-scale clipped to the stable-gradient range,
srl/rl/tf/distributions/normal_dist_block.py:76-155), the worker samples loc + scale * N(0,1), rescales [-1, 1] onto the
environment's bounds and sanitises (:332-339), the trainer's losses and seeds come from `srlx_ppo_loss_normal`.  The
vectorised engine for the same policy (E environments on the device) is device/ppo.py."""
import math
from dataclasses import dataclass, field
from typing import Any, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.base.exception import UndefinedError
from simple_distributed_rl_amd.base.rl.algorithms.base_ppo import RLConfig, RLWorker
from simple_distributed_rl_amd.base.rl.memory import RLMemory
from simple_distributed_rl_amd.base.rl.parameter import RLParameter
from simple_distributed_rl_amd.base.rl.registration import register
from simple_distributed_rl_amd.base.rl.trainer import RLTrainer
from simple_distributed_rl_amd.base.spaces.discrete import DiscreteSpace
from simple_distributed_rl_amd.base.spaces.np_array import NpArraySpace
from simple_distributed_rl_amd.rl.memories.replay_buffer import ReplayBuffer
from simple_distributed_rl_amd.rl.models.config import HiddenBlockConfig, InputBlockConfig
from simple_distributed_rl_amd.rl.schedulers.lr_scheduler import LRSchedulerConfig

from ._device_ops import require_gpu


@dataclass
class MemoryConfig:
    warmup_size: int = 500
    compress: bool = False
    compress_level: int = -1


@dataclass
class Config(RLConfig):
    batch_size: int = 32
    memory: MemoryConfig = field(default_factory=lambda: MemoryConfig())
    train_num: int = 50
    input_block: InputBlockConfig = field(default_factory=lambda: InputBlockConfig())
    hidden_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((64, 64)))
    value_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((64,)))
    policy_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((64,)))
    experience_collection_method: str = "GAE"  # "MC" | "GAE"
    discount: float = 0.9
    gae_discount: float = 0.9
    baseline_type: str = "advantage"  # "" "none" | "ave" | "std" | "normal" | "advantage" "v"
    surrogate_type: str = "clip"  # "" | "clip"  ("kl" is not offered: see module docstring of device/ppo.py)
    policy_clip_range: float = 0.2
    adaptive_kl_target: float = 0.01
    enable_value_clip: float = True
    value_clip_range: float = 0.2
    lr: float = 0.0002
    lr_scheduler: LRSchedulerConfig = field(default_factory=lambda: LRSchedulerConfig().set_step(2000, 0.01))
    value_loss_weight: float = 1.0
    entropy_weight: float = 0.01
    enable_state_normalized: bool = False
    global_gradient_clip_norm: float = 0.5
    state_clip: Optional[Tuple[float, float]] = None
    reward_clip: Optional[Tuple[float, float]] = None
    enable_stable_gradients: bool = True
    stable_gradients_scale_range: tuple = (1e-10, 10)
    #: NOT a reference field.  The reference's `Trainer.train` (ppo.py:203-205) loops `for _ in range(train_num): f = f or
    #: self._train()`, and `or` short-circuits: after the first successful minibatch update the remaining `train_num - 1`
    #: iterations do nothing, so ONE gradient step is taken per collected buffer before it is cleared.  False (default) keeps
    #: exactly that; True applies all `train_num` minibatch updates per buffer (what the field name suggests).
    train_every_epoch: bool = False

    def get_name(self) -> str:
        return "PPO"

    def get_framework(self) -> str:
        return "torch"


register(Config(), __name__ + ":Memory", __name__ + ":Parameter", __name__ + ":Trainer", __name__ + ":Worker", check_duplicate=False)


class Memory(RLMemory):
    """ppo.py:28-52: a ReplayBuffer that holds one on-policy generation (capacity = warm-up + 100), cleared by the trainer."""

    def setup(self):
        m = self.config.memory
        self.memory = ReplayBuffer(self.config.batch_size, m.warmup_size + 100, m.warmup_size, m.compress, m.compress_level)
        self.register_worker_func_custom(self.add, self.memory.serialize)
        self.register_trainer_recv_func(self.sample)
        self.register_trainer_send_func(self.clear)

    def length(self) -> int:
        return self.memory.length()

    def add(self, batch: Any, serialized: bool = False) -> None:
        self.memory.add(batch, serialized)

    def sample(self):
        return self.memory.sample()

    def clear(self):
        self.memory.clear()

    def call_backup(self, **kwargs):
        return self.memory.call_backup()

    def call_restore(self, data, **kwargs):
        self.memory.call_restore(data)


class ActorCriticNetwork(nn.Module):
    """ppo.py:55-99: input block -> shared hidden block -> (value block -> Dense(1), policy block -> logits);
    orthogonal initialisation of the value head (:61, :71)."""

    def __init__(self, config: Config):
        super().__init__()
        self.in_block = config.input_block.create_torch_block(config)
        self.hidden_block = config.hidden_block.create_torch_block(self.in_block.out_size)
        h = self.hidden_block.out_size
        self.value_block = config.value_block.create_torch_block(h)
        self.value_out = nn.Linear(self.value_block.out_size, 1)
        nn.init.orthogonal_(self.value_out.weight)
        nn.init.zeros_(self.value_out.bias)
        self.policy_block = config.policy_block.create_torch_block(h)
        ph = self.policy_block.out_size
        self.continuous = isinstance(config.action_space, NpArraySpace)
        if isinstance(config.action_space, DiscreteSpace):
            self.policy_out = nn.Linear(ph, config.action_space.n)  # CategoricalDistBlock: logits
        elif self.continuous:  # NormalDistBlock: a loc layer (truncated-normal bias) and a log-scale layer (zero bias)
            self.policy_out = nn.Linear(ph, config.action_space.size)
            nn.init.trunc_normal_(self.policy_out.bias, std=0.05)
            self.log_scale_out = nn.Linear(ph, config.action_space.size)
            nn.init.zeros_(self.log_scale_out.bias)
            lo, hi = config.stable_gradients_scale_range if config.enable_stable_gradients else (1e-30, 1e30)
            self.log_scale_range = (math.log(lo), math.log(hi))
        else:
            raise UndefinedError(config.action_space)

    def forward(self, x):
        """(v, logits) for discrete actions, (v, loc, log_scale) for continuous ones (log_scale NOT yet clipped: the loss kernel
        clips it itself so that the clip's zero gradient is part of its seeds)."""
        x = self.hidden_block(self.in_block(x))
        v = self.value_out(self.value_block(x))
        p = self.policy_block(x)
        if self.continuous:
            return v, self.policy_out(p), self.log_scale_out(p)
        return v, self.policy_out(p)


class Parameter(RLParameter):
    def setup(self):
        self.np_dtype = self.config.get_dtype("np")
        self.device = torch.device(self.config.used_device_torch)
        self.model = ActorCriticNetwork(self.config).to(self.device)
        self.adaptive_kl_beta = 0.5  # ppo.py:176 (kept in the backup layout; the "kl" surrogate is not offered)

    def call_restore(self, data: Any, **kwargs) -> None:
        self.model.load_state_dict(data[0])
        self.adaptive_kl_beta = data[1]

    def call_backup(self, serialized: bool = False, **kwargs) -> Any:
        sd = self.model.state_dict()
        if serialized:
            sd = {k: v.detach().to("cpu").clone() for k, v in sd.items()}
        return [sd, self.adaptive_kl_beta]

    def to_device(self, device):
        self.device = torch.device(device)
        self.model.to(self.device)

    def pred(self, state: np.ndarray):
        with torch.no_grad():
            return self.model(torch.as_tensor(np.asarray(state, dtype=self.np_dtype), device=self.device))


class Trainer(RLTrainer):
    def on_setup(self) -> None:
        self.device = require_gpu(self.config.used_device_torch)
        self.parameter.to_device(self.device)
        self.lib = N.lib()
        self.np_dtype = self.config.get_dtype("np")
        self.optimizer = torch.optim.Adam(self.parameter.model.parameters(), lr=self.config.lr)
        self.lr_sch = self.config.lr_scheduler.apply_torch_scheduler(self.optimizer)
        if self.config.surrogate_type not in ("clip", ""):
            raise UndefinedError(self.config.surrogate_type)
        self.parameter.model.train()

    def train(self) -> None:  # ppo.py:191-201
        if self.memory.sample() is None:
            return
        trained = False
        for _ in range(self.config.train_num):
            if self.config.train_every_epoch:
                trained = self._train() or trained
            else:
                trained = trained or self._train()  # the reference's short-circuit (ppo.py:203-205): one update per buffer
        if trained:
            self.memory.clear()

    def losses_and_seeds(self, new_logpi, old_logpi, advantage, v, v_target, old_v):
        """compute_train_loss (ppo.py:102-169) on device tensors: (losses[3], d loss/d new_logpi, d loss/d v)."""
        cfg, d = self.config, self.device
        B = new_logpi.shape[0]
        losses = torch.empty(3, dtype=torch.float32, device=d)
        g_lp = torch.empty((B, 1), dtype=torch.float32, device=d)
        g_v = torch.empty(B, dtype=torch.float32, device=d)
        keep = [t.detach().contiguous().float() for t in (new_logpi, old_logpi, advantage, v, v_target, old_v)]
        N.check(self.lib.srlx_ppo_loss_logpi(
            B, 1, N.tptr(keep[0]), N.tptr(keep[1]), N.tptr(keep[2]), N.tptr(keep[3]), N.tptr(keep[4]), N.tptr(keep[5]),
            int(cfg.baseline_type in ("advantage", "v")), int(cfg.surrogate_type == "clip"), float(cfg.policy_clip_range), int(bool(cfg.enable_value_clip)),
            float(cfg.value_clip_range), float(cfg.value_loss_weight), float(cfg.entropy_weight), N.tptr(losses), N.tptr(g_lp), N.tptr(g_v), N.torch_stream_ptr()))
        self._keep = keep
        return losses, g_lp, g_v

    def losses_and_seeds_normal(self, loc, log_scale, action, old_logpi, advantage, v, v_target, old_v):
        """The same for a Normal policy (`srlx_ppo_loss_normal`): (losses[3], d/d loc, d/d log_scale, d/d v)."""
        cfg, d = self.config, self.device
        B, D = loc.shape
        losses = torch.empty(3, dtype=torch.float32, device=d)
        g_loc = torch.empty((B, D), dtype=torch.float32, device=d)
        g_ls = torch.empty((B, D), dtype=torch.float32, device=d)
        g_v = torch.empty(B, dtype=torch.float32, device=d)
        keep = [t.detach().contiguous().float() for t in (loc, log_scale, action, old_logpi, advantage, v, v_target, old_v)]
        lo, hi = self.parameter.model.log_scale_range
        N.check(self.lib.srlx_ppo_loss_normal(
            B, D, N.tptr(keep[0]), N.tptr(keep[1]), float(lo), float(hi), N.tptr(keep[2]), N.tptr(keep[3]), N.tptr(keep[4]), N.tptr(keep[5]), N.tptr(keep[6]),
            N.tptr(keep[7]), int(cfg.baseline_type in ("advantage", "v")), int(cfg.surrogate_type == "clip"), float(cfg.policy_clip_range),
            int(bool(cfg.enable_value_clip)), float(cfg.value_clip_range), float(cfg.value_loss_weight), float(cfg.entropy_weight), N.tptr(losses),
            N.tptr(g_loc), N.tptr(g_ls), N.tptr(g_v), N.torch_stream_ptr()))
        self._keep = keep
        return losses, g_loc, g_ls, g_v

    def _train(self) -> bool:
        batches = self.memory.sample()
        if batches is None:
            return False
        cfg, d = self.config, self.device
        states = np.asarray([e["state"] for e in batches], dtype=self.np_dtype)
        adv = np.asarray([e["discounted_reward"] for e in batches], dtype=np.float32)
        v_target = adv.copy()  # :214-215: the same numbers serve as value target and advantage
        if cfg.enable_state_normalized:  # :218-219
            states = (states - np.mean(states, axis=0, keepdims=True)) / (np.std(states, axis=0, keepdims=True) + 1e-8)
        bt = cfg.baseline_type  # :222-233
        if bt == "ave":
            adv = adv - np.mean(adv)
        elif bt == "std":
            adv = adv / (np.std(adv) + 1e-8)
        elif bt == "normal":
            adv = (adv - np.mean(adv)) / (np.std(adv) + 1e-8)
        elif bt not in ("", "none", "advantage", "v"):
            raise UndefinedError(bt)
        actions = torch.as_tensor(np.asarray([e["action"] for e in batches], dtype=np.float32), device=d)  # one-hot rows / action vectors
        old_logpi = torch.as_tensor(np.asarray([e["log_prob"] for e in batches], dtype=np.float32), device=d).view(len(batches), -1)
        old_v = torch.as_tensor(np.asarray([e["v"] for e in batches], dtype=np.float32), device=d)
        adv_t, vt_t = torch.as_tensor(adv, device=d), torch.as_tensor(v_target, device=d)
        out = self.parameter.model(torch.as_tensor(states.astype(self.np_dtype), device=d))
        v1 = out[0].view(-1)
        if self.parameter.model.continuous:
            loc, log_scale = out[1], out[2]
            losses, g_loc, g_ls, g_v = self.losses_and_seeds_normal(loc, log_scale, actions.view(loc.shape), old_logpi, adv_t, v1, vt_t, old_v)
            heads, seeds = [loc, log_scale, v1], [g_loc, g_ls, g_v]
        else:
            new_logpi = (torch.log_softmax(out[1], dim=-1) * actions).sum(-1, keepdim=True)  # CategoricalDist.log_prob(onehot)
            losses, g_lp, g_v = self.losses_and_seeds(new_logpi, old_logpi, adv_t, v1, vt_t, old_v)
            heads, seeds = [new_logpi, v1], [g_lp, g_v]
        self.optimizer.zero_grad()
        torch.autograd.backward(heads, seeds)  # the fused kernel's seeds through the network
        if cfg.global_gradient_clip_norm != 0:  # :269-270
            torch.nn.utils.clip_grad_norm_(self.parameter.model.parameters(), cfg.global_gradient_clip_norm)
        self.optimizer.step()
        if self.lr_sch is not None:
            self.lr_sch.step()
        pl, vl, el = losses.tolist()
        self.info["policy_loss"], self.info["value_loss"], self.info["entropy_loss"] = pl, vl, el
        self.train_count += 1
        return True


class Worker(RLWorker):
    def on_setup(self, worker, context) -> None:
        if self.distributed:
            raise NotImplementedError("NotSupported")  # ppo.py:295-297
        if self.training and self.config.experience_collection_method == "GAE":
            require_gpu(str(self.parameter.device))  # the episode GAE is srlx_gae_scan: it takes device pointers, there is no CPU path
        self.lib = N.lib()

    def on_reset(self, worker):
        self.recent_batch = []
        self.recent_rewards = []
        self.recent_next_states = []

    def _clip_state(self, state):
        c = self.config.state_clip
        return state if c is None else np.clip(state, c[0], c[1])

    def policy(self, worker):
        state = self._clip_state(worker.state)
        out = self.parameter.pred(state[np.newaxis, ...])
        v = out[0]
        if self.parameter.model.continuous:
            loc, log_scale = out[1][0], out[2][0]
            lo, hi = self.parameter.model.log_scale_range
            log_scale = torch.clamp(log_scale, lo, hi)
            scale = torch.exp(log_scale)
            action = loc + scale * torch.randn_like(loc) if self.training else loc  # NormalDist.sample / mean (:316-319)
            logp = -0.5 * math.log(2 * math.pi) - log_scale - 0.5 * ((action - loc) / scale) ** 2  # normal_dist_block.py:13-20
            a_np = action.cpu().numpy().astype(np.float32)
            self.recent_batch.append({
                "state": state,
                "action": a_np,
                "v": float(v.item()),
                "log_prob": np.maximum(logp.cpu().numpy().astype(np.float32), math.log(1e-6)),  # :322
            })
            if np.isnan(a_np).any():  # :333-335
                return self.config.action_space.sample()
            env_action = self.config.action_space.rescale_from(a_np)  # the policy's [-1, 1] onto the environment's bounds (:336)
            return self.config.action_space.sanitize(env_action)
        logp = torch.log_softmax(out[1], dim=-1)[0]
        if self.training:
            a = int(torch.multinomial(torch.exp(logp), 1).item())  # CategoricalDist.sample (:301)
        else:
            a = int(torch.argmax(logp).item())
        onehot = np.zeros(self.config.action_space.n, np.float32)
        onehot[a] = 1.0
        self.recent_batch.append({
            "state": state,
            "action": onehot,
            "v": float(v.item()),
            "log_prob": max(float(logp[a].item()), math.log(1e-6)),  # :307
        })
        return a

    def on_step(self, worker):
        if not self.training:
            return
        reward = worker.reward
        rc = self.config.reward_clip
        if rc is not None:  # :374-379
            reward = min(max(reward, rc[0]), rc[1])
        cfg = self.config
        if cfg.experience_collection_method == "GAE":
            self.recent_next_states.append(self._clip_state(worker.next_state))
        self.recent_rewards.append(reward)
        if not worker.done:
            return
        T = len(self.recent_batch)
        if cfg.experience_collection_method == "MC":  # :385-393
            mc_r = 0.0
            for i in reversed(range(T)):
                mc_r = self.recent_rewards[i] + cfg.discount * mc_r
                self.recent_batch[i]["discounted_reward"] = np.asarray(mc_r, dtype=np.float32)
                self.memory.add(self.recent_batch[i])
        elif cfg.experience_collection_method == "GAE":  # :395-410
            d = self.parameter.device
            states = np.asarray([e["state"] for e in self.recent_batch], dtype=self.parameter.np_dtype)
            v = self.parameter.pred(states)[0]
            # the scan bootstraps step t with V(s_t+1) = v[t+1] inside an episode; the reference evaluates the network
            # on the stored next states, which ARE the following states (:396-397), and drops the bootstrap at the end
            rew = torch.as_tensor(np.asarray(self.recent_rewards, dtype=np.float32), device=d).view(T, 1)
            val = v.view(T, 1).float().contiguous()
            done = torch.zeros((T, 1), dtype=torch.uint8, device=d)
            done[T - 1, 0] = 1
            gae = torch.empty((T, 1), dtype=torch.float32, device=d)
            N.check(self.lib.srlx_gae_scan(1, T, N.tptr(rew), N.tptr(val), N.tptr(done), None, float(cfg.discount), float(cfg.gae_discount), N.tptr(gae),
                                           N.torch_stream_ptr()))
            g = gae.view(-1).cpu().numpy()
            for i in reversed(range(T)):  # :400 (the reference adds in reverse order)
                self.recent_batch[i]["discounted_reward"] = np.asarray(g[i], dtype=np.float32)
                self.memory.add(self.recent_batch[i])
        else:
            raise UndefinedError(cfg.experience_collection_method)
