"""Agent57_light plugin (srl/algorithms/agent57_light/agent57_light.py:56-545, model_torch.py:18-443),
registered as "Agent57_light:torch".

Worker: the reference's host logic (sliding-window UCB meta-controller, UVFA inputs, epsilon-greedy on
q_ext + beta*q_int) with the intrinsic reward evaluated on the GPU: the embedding stays on the device, the
episodic memory is a device-resident structure-of-arrays buffer and `srlx_ngu_episodic_reward` replaces the
per-entry `np.linalg.norm` list comprehension (agent57_light.py:488); the RND error is reduced by
`srlx_ngu_lifelong_reward`.
Trainer: batch -> GPU once; for each of the two Q-networks the per-actor-discount double-DQN target
(`srlx_dqn_target` with a per-sample gamma), Huber loss + gradient seed (fused libsrlx kernel) and the mixed
priorities |td_ext + beta*td_int| (`srlx_agent57_priority`) run in HIP; forwards/backwards/Adam through torch."""
import random
from dataclasses import dataclass, field
from typing import Any, List

import numpy as np
import torch
import torch.nn as nn

from simple_distributed_rl_amd.base.rl.algorithms.base_dqn import RLConfig, RLWorker
from simple_distributed_rl_amd.base.rl.parameter import RLParameter
from simple_distributed_rl_amd.base.rl.registration import register
from simple_distributed_rl_amd.base.rl.trainer import RLTrainer
from simple_distributed_rl_amd.rl import functions as funcs
from simple_distributed_rl_amd.rl.memories.priority_replay_buffer import PriorityReplayBufferConfig, RLPriorityReplayBuffer
from simple_distributed_rl_amd.rl.models.config import DuelingNetworkConfig, HiddenBlockConfig, InputBlockConfig, RLConfigComponentFramework

from ._device_ops import NguOps, TdOps, invalid_mask, require_gpu


@dataclass
class Config(RLConfig, RLConfigComponentFramework):
    test_epsilon: float = 0
    test_beta: float = 0
    batch_size: int = 32
    memory: PriorityReplayBufferConfig = field(default_factory=lambda: PriorityReplayBufferConfig().set_proportional())
    lr_ext: float = 0.0001
    lr_int: float = 0.0001
    target_model_update_interval: int = 1500
    enable_double_dqn: bool = True
    enable_rescale: bool = False
    input_block: InputBlockConfig = field(default_factory=lambda: InputBlockConfig())
    hidden_block: DuelingNetworkConfig = field(default_factory=lambda: DuelingNetworkConfig().set_dueling_network((512,)))
    # meta-controller (agent57_light.py:93-100)
    actor_num: int = 32
    ucb_window_size: int = 3600
    ucb_epsilon: float = 0.01
    ucb_beta: float = 1
    enable_intrinsic_reward: bool = True
    # episodic novelty (:105-122)
    episodic_lr: float = 0.0005
    episodic_count_max: int = 10
    episodic_epsilon: float = 0.001
    episodic_cluster_distance: float = 0.008
    episodic_memory_capacity: int = 30000
    episodic_pseudo_counts: float = 0.1
    episodic_emb_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((32,)))
    episodic_out_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((128,)))
    # lifelong novelty (:124-131)
    lifelong_lr: float = 0.0005
    lifelong_max: float = 5.0
    lifelong_hidden_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((128,)))
    # UVFA (:133-138)
    input_ext_reward: bool = True
    input_int_reward: bool = False
    input_action: bool = False
    disable_int_priority: bool = False
    dummy_state_val: float = 0.0

    def get_name(self) -> str:
        return "Agent57_light"

    def get_framework(self) -> str:
        return RLConfigComponentFramework.get_framework(self)


register(Config(), __name__ + ":Memory", __name__ + ":Parameter", __name__ + ":Trainer", __name__ + ":Worker", check_duplicate=False)


class Memory(RLPriorityReplayBuffer):
    pass


# ---------------------------------------------------------------------------------------------------
# networks (model_torch.py:18-117); attribute names give the reference's state_dict keys
# ---------------------------------------------------------------------------------------------------
class QNetwork(nn.Module):
    def __init__(self, config: Config):
        super().__init__()
        self.input_ext_reward = config.input_ext_reward
        self.input_int_reward = config.input_int_reward and config.enable_intrinsic_reward
        self.input_action = config.input_action
        self.in_block = config.input_block.create_torch_block(config)
        in_size = self.in_block.out_size + int(self.input_ext_reward) + int(self.input_int_reward) + config.actor_num
        if self.input_action:
            in_size += config.action_space.n
        self.hidden_block = config.hidden_block.create_torch_block(in_size, config.action_space.n)

    def forward(self, inputs):
        state, reward_ext, reward_int, onehot_action, onehot_actor = inputs
        parts = [self.in_block(state)]
        if self.input_ext_reward:
            parts.append(reward_ext)
        if self.input_int_reward:
            parts.append(reward_int)
        if self.input_action:
            parts.append(onehot_action)
        parts.append(onehot_actor)
        return self.hidden_block(torch.cat(parts, dim=1))


class EmbeddingNetwork(nn.Module):
    """Inverse-dynamics embedding: f(s), f(s') -> action probabilities (model_torch.py:70-99)."""

    def __init__(self, config: Config):
        super().__init__()
        self.in_block = config.input_block.create_torch_block(config)
        self.emb_block = config.episodic_emb_block.create_torch_block(self.in_block.out_size)
        self.out_block = config.episodic_out_block.create_torch_block(self.emb_block.out_size * 2)
        self.out_block_normalize = nn.LayerNorm(self.out_block.out_size)
        self.out_block_out1 = nn.Linear(self.out_block.out_size, config.action_space.n)

    def predict(self, state):
        return self.emb_block(self.in_block(state))

    def forward(self, x):
        h = torch.cat([self.predict(x[0]), self.predict(x[1])], dim=1)
        return torch.softmax(self.out_block_out1(self.out_block_normalize(self.out_block(h))), dim=1)


class LifelongNetwork(nn.Module):
    """RND trunk (model_torch.py:105-117)."""

    def __init__(self, config: Config):
        super().__init__()
        self.in_block = config.input_block.create_torch_block(config)
        self.hidden_block = config.lifelong_hidden_block.create_torch_block(self.in_block.out_size)
        self.hidden_normalize = nn.LayerNorm(self.hidden_block.out_size)

    def forward(self, x):
        return self.hidden_normalize(self.hidden_block(self.in_block(x)))


def _backup(model: nn.Module, serialized: bool):
    sd = model.state_dict()
    return {k: v.detach().to("cpu").clone() for k, v in sd.items()} if serialized else sd


class Parameter(RLParameter):
    def setup(self):
        self.np_dtype = self.config.get_dtype("np")
        self.device = torch.device(self.config.used_device_torch)
        c = self.config
        self.q_ext_online, self.q_ext_target = QNetwork(c).to(self.device), QNetwork(c).to(self.device)
        self.q_int_online, self.q_int_target = QNetwork(c).to(self.device), QNetwork(c).to(self.device)
        self.q_ext_target.eval()
        self.q_int_target.eval()
        self.q_ext_target.load_state_dict(self.q_ext_online.state_dict())
        self.q_int_target.load_state_dict(self.q_int_online.state_dict())
        self.emb_network = EmbeddingNetwork(c).to(self.device)
        self.lifelong_target = LifelongNetwork(c).to(self.device)
        self.lifelong_train = LifelongNetwork(c).to(self.device)
        self.lifelong_target.eval()

    def _nets(self):
        return [self.q_ext_online, self.q_ext_target, self.q_int_online, self.q_int_target, self.emb_network, self.lifelong_target, self.lifelong_train]

    def to_device(self, device):
        self.device = torch.device(device)
        for m in self._nets():
            m.to(self.device)

    def call_restore(self, data: Any, from_serialized: bool = False, **kwargs) -> None:  # model_torch.py:139-146
        self.q_ext_online.load_state_dict(data[0])
        self.q_ext_target.load_state_dict(data[0])
        self.q_int_online.load_state_dict(data[1])
        self.q_int_target.load_state_dict(data[1])
        self.emb_network.load_state_dict(data[2])
        self.lifelong_target.load_state_dict(data[3])
        self.lifelong_train.load_state_dict(data[4])

    def call_backup(self, serialized: bool = False, **kwargs):  # :148-156
        return [_backup(m, serialized) for m in (self.q_ext_online, self.q_int_online, self.emb_network, self.lifelong_target, self.lifelong_train)]

    # device-side forwards (tensors in, tensors out; the worker and the trainer never leave the GPU between them)
    def q_inputs(self, state, reward_ext, reward_int, onehot_action, onehot_actor):
        d = self.device
        as_t = lambda x, dt=torch.float32: x.to(d) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float32), device=d)  # noqa: E731
        return [as_t(state), as_t(reward_ext), as_t(reward_int), as_t(onehot_action), as_t(onehot_actor)]

    def predict_q_ext_online(self, x) -> np.ndarray:
        with torch.no_grad():
            return self.q_ext_online(self.q_inputs(*x)).cpu().numpy()

    def predict_q_int_online(self, x) -> np.ndarray:
        with torch.no_grad():
            return self.q_int_online(self.q_inputs(*x)).cpu().numpy()

    def predict_q_ext_target(self, x) -> np.ndarray:
        with torch.no_grad():
            return self.q_ext_target(self.q_inputs(*x)).cpu().numpy()

    def predict_q_int_target(self, x) -> np.ndarray:
        with torch.no_grad():
            return self.q_int_target(self.q_inputs(*x)).cpu().numpy()


class Trainer(RLTrainer):
    """Host batch -> GPU once, then `Agent57LightLearner` (device/agent57_light.py): the update shared with the E-environment engine."""

    def on_setup(self) -> None:
        from simple_distributed_rl_amd.device.agent57_light import Agent57LightLearner

        self.device = require_gpu(self.config.used_device_torch)
        self.parameter.to_device(self.device)
        self.core = Agent57LightLearner(self.config, self.parameter, self.device, channels_first=False)
        self.core.train_count = self.train_count
        self.np_dtype = self.config.get_dtype("np")

    # the optimisers / lists of the reference trainer, for callers that reach into it
    q_ext_optimizer = property(lambda self: self.core.q_ext_optimizer)
    q_int_optimizer = property(lambda self: self.core.q_int_optimizer)
    emb_optimizer = property(lambda self: self.core.emb_optimizer)
    lifelong_optimizer = property(lambda self: self.core.lifelong_optimizer)
    beta_list = property(lambda self: self.core.beta_list)
    discount_list = property(lambda self: self.core.discount_list)
    sync_count = property(lambda self: self.core.sync_count)
    td_ext = property(lambda self: self.core.td_ext)
    td_int = property(lambda self: self.core.td_int)

    def train(self) -> None:
        sampled = self.memory.sample()
        if sampled is None:
            return
        batches, weights, update_args = sampled
        cfg, d = self.config, self.device
        (states, n_states, onehot_actions, next_invalid, rewards_ext, rewards_int, undone, prev_onehot_actions, prev_rewards_ext, prev_rewards_int,
         actor_idx) = zip(*batches)
        B, A = len(batches), cfg.action_space.n
        f32 = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32), device=d)  # noqa: E731
        action = torch.as_tensor(np.argmax(np.asarray(onehot_actions), axis=1).astype(np.int64), device=d)
        prev_action = torch.as_tensor(np.argmax(np.asarray(prev_onehot_actions), axis=1).astype(np.int64), device=d)
        actor = torch.as_tensor(np.asarray(actor_idx, dtype=np.int64), device=d)
        self.core.train_count = self.train_count
        priorities = self.core.update(f32(states), f32(n_states), action, f32(rewards_ext), f32(rewards_int), f32(undone), prev_action, f32(prev_rewards_ext),
                                      f32(prev_rewards_int), actor, f32(weights), invalid_mask(next_invalid, (B, A), d))
        self.info.update(self.core.losses())
        self.memory.update(update_args, priorities.cpu().numpy(), self.train_count)
        self.train_count += 1


class UcbMetaController:
    """Sliding-window UCB over the actor (beta, epsilon, gamma) family, agent57_light.py:317-353."""

    def __init__(self, actor_num: int, window_size: int, epsilon: float, beta: float, tie_break=funcs.get_random_max_index):
        self.actor_num, self.window_size, self.epsilon, self.beta, self.tie_break = actor_num, window_size, epsilon, beta, tie_break
        self.actor_index = -1
        self.recent: List[tuple] = []
        self.count = [1] * actor_num  # every arm counts as tried once
        self.reward = [0.0] * actor_num

    def next_actor(self, last_episode_reward: float) -> int:
        if self.actor_index != -1:
            self.recent.append((self.actor_index, last_episode_reward))
            self.count[self.actor_index] += 1
            self.reward[self.actor_index] += last_episode_reward
            if len(self.recent) >= self.window_size:
                old_actor, old_reward = self.recent.pop(0)
                self.count[old_actor] -= 1
                self.reward[old_actor] -= old_reward
        n_recent = len(self.recent)
        if n_recent < self.actor_num:  # round-robin first
            self.actor_index = n_recent
        elif random.random() < self.epsilon:
            self.actor_index = random.randint(0, self.actor_num - 1)
        else:
            ucbs = [self.reward[i] / self.count[i] + self.beta * np.sqrt(np.log(n_recent) / self.count[i]) for i in range(self.actor_num)]
            self.actor_index = self.tie_break(ucbs)
        return self.actor_index


class Worker(RLWorker):
    def on_setup(self, worker, context) -> None:
        c = self.config
        self.beta_list = funcs.create_beta_list(c.actor_num)
        self.epsilon_list = funcs.create_epsilon_list(c.actor_num)
        self.discount_list = funcs.create_discount_list(c.actor_num)
        self.ucb = UcbMetaController(c.actor_num, c.ucb_window_size, c.ucb_epsilon, c.ucb_beta)
        self.discount = 0
        self.episode_reward = 0.0
        self.action_eye = np.identity(c.action_space.n, dtype=np.float32)
        self.actor_eye = np.identity(c.actor_num, dtype=np.float32)
        self.ngu = None
        self.ops = None
        if c.enable_intrinsic_reward:
            dev = require_gpu(str(self.parameter.device))
            emb_dim = self.parameter.emb_network.emb_block.out_size
            self.ngu = NguOps(dev, 1, emb_dim, c.episodic_memory_capacity, c.episodic_count_max, c.episodic_epsilon, c.episodic_cluster_distance, c.episodic_pseudo_counts)
        if self.distributed and c.memory.requires_priority():
            self.ops = TdOps(require_gpu(str(self.parameter.device)))

    def on_reset(self, worker):
        c = self.config
        if self.training:  # one actor of the family per episode (:288-293)
            self.actor_index = self.ucb.next_actor(self.episode_reward)
            self.beta, self.epsilon, self.discount = self.beta_list[self.actor_index], self.epsilon_list[self.actor_index], self.discount_list[self.actor_index]
        else:
            self.actor_index, self.epsilon, self.beta = 0, c.test_epsilon, c.test_beta
        self.prev_onehot_action = self.action_eye[random.randint(0, c.action_space.n - 1)]
        self.prev_reward_ext = 0
        self.prev_reward_int = 0
        self.onehot_actor_idx = self.actor_eye[self.actor_index][np.newaxis, ...]
        self.episode_reward = 0.0
        if self.ngu is not None:
            self.ngu.reset()  # a fresh episodic memory per episode (:310-311)
        self.info["epsilon"], self.info["beta"], self.info["discount"] = self.epsilon, self.beta, self.discount

    def policy(self, worker) -> int:
        in_ = [worker.state[np.newaxis, ...], np.array([[self.prev_reward_ext]], np.float32), np.array([[self.prev_reward_int]], np.float32),
               self.prev_onehot_action[np.newaxis, ...], self.onehot_actor_idx]
        self.q_ext = self.parameter.predict_q_ext_online(in_)[0]
        self.q_int = self.parameter.predict_q_int_online(in_)[0]
        self.q = self.q_ext + self.beta * self.q_int
        invalid_actions = worker.invalid_actions
        if random.random() < self.epsilon:
            action = random.choice([a for a in range(self.config.action_space.n) if a not in invalid_actions])
        else:
            self.q[invalid_actions] = -np.inf
            action = int(np.argmax(self.q))
        self.onehot_action = self.action_eye[action]
        return action

    def intrinsic_reward(self, next_state: np.ndarray):
        """agent57_light.py:383-391, 473-529 with both novelty terms evaluated on the device."""
        p = self.parameter
        with torch.no_grad():
            s = torch.as_tensor(np.asarray(next_state, np.float32)[np.newaxis, ...], device=p.device)
            p.emb_network.eval()
            p.lifelong_train.eval()
            episodic = self.ngu.episodic(p.emb_network.predict(s))
            lifelong = self.ngu.lifelong(p.lifelong_target(s), p.lifelong_train(s), self.config.lifelong_max)
            both = torch.stack([episodic[0], lifelong[0]]).cpu().numpy()
        return both[0], both[1], both[0] * both[1]

    def on_step(self, worker):
        c = self.config
        next_state, reward_ext, next_invalid_actions = worker.next_state, worker.reward, worker.next_invalid_actions
        self.episode_reward += reward_ext
        if c.enable_intrinsic_reward:
            self.info["episodic"], self.info["lifelong"], reward_int = self.intrinsic_reward(next_state)
            self.info["reward_int"] = reward_int
        else:
            reward_int = 0.0
        prev_onehot_action, prev_reward_ext, prev_reward_int = self.prev_onehot_action, self.prev_reward_ext, self.prev_reward_int
        self.prev_onehot_action, self.prev_reward_ext, self.prev_reward_int = self.onehot_action, reward_ext, reward_int
        if not self.training:
            return
        undone = int(not worker.terminated)
        batch = [worker.state, next_state, self.onehot_action, next_invalid_actions, reward_ext, reward_int, undone, prev_onehot_action, prev_reward_ext,
                 prev_reward_int, self.actor_index]  # :420-432
        priority = None
        if self.ops is not None:  # distributed initial priority (:439-468)
            priority = self._initial_priority(worker, next_state, reward_ext, reward_int, prev_reward_ext, prev_reward_int, next_invalid_actions, undone)
        self.memory.add(batch, priority)

    def _initial_priority(self, worker, next_state, reward_ext, reward_int, prev_reward_ext, prev_reward_int, next_invalid_actions, undone) -> float:
        c, p, d = self.config, self.parameter, self.parameter.device
        A = c.action_space.n
        x = p.q_inputs(next_state[np.newaxis, ...], np.array([[prev_reward_ext]], np.float32), np.array([[prev_reward_int]], np.float32),
                       self.prev_onehot_action[np.newaxis, ...], self.onehot_actor_idx)
        inv = invalid_mask([next_invalid_actions], (1, A), d)
        und = torch.tensor([undone], dtype=torch.float32, device=d)
        disc = torch.tensor([self.discount], dtype=torch.float32, device=d)

        def one(online, target_net, reward):
            with torch.no_grad():
                q_tg = target_net(x)
                q_on = online(x) if c.enable_double_dqn else None
            return self.ops.dqn_target(q_on, q_tg, torch.tensor([reward], dtype=torch.float32, device=d), und, inv, 0.0, c.enable_double_dqn, c.enable_rescale,
                                       False, discount_per_sample=disc)

        target_ext = float(one(p.q_ext_online, p.q_ext_target, reward_ext).item())
        if c.disable_int_priority or not c.enable_intrinsic_reward or self.beta == 0:
            return abs(target_ext - float(self.q_ext[worker.action]))
        target_int = float(one(p.q_int_online, p.q_int_target, reward_int).item())
        return abs((target_ext + self.beta * target_int) - float(self.q[worker.action]))
