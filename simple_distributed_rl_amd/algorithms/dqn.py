"""DQN plugin (srl/algorithms/dqn/dqn.py:50-246, srl/algorithms/dqn/model_torch.py:17-132), registered as
"DQN:torch".  The worker is the reference's host logic (one env, Python `random` stream); the trainer keeps
the batch on the GPU from the first tensor on: forwards through torch (MIOpen/hipBLASLt), the 1-step
(double-)DQN target in `srlx_dqn_target`, Huber loss + d loss/d q + priorities in the fused libsrlx kernel --
two host<->device hops per train() (batch in, priorities out) instead of the reference's four."""
import random
from dataclasses import dataclass, field
from typing import Any

import numpy as np
import torch

from simple_distributed_rl_amd.base.rl.algorithms.base_dqn import RLConfig, RLWorker
from simple_distributed_rl_amd.base.rl.parameter import RLParameter
from simple_distributed_rl_amd.base.rl.registration import register
from simple_distributed_rl_amd.base.rl.trainer import RLTrainer
from simple_distributed_rl_amd.rl.memories.priority_replay_buffer import PriorityReplayBufferConfig, RLPriorityReplayBuffer
from simple_distributed_rl_amd.rl.models.config import HiddenBlockConfig, InputBlockConfig, RLConfigComponentFramework
from simple_distributed_rl_amd.rl.schedulers.scheduler import SchedulerConfig
from simple_distributed_rl_amd.rl.torch_.networks import QNetwork

from ._device_ops import TdOps, invalid_mask, require_gpu


@dataclass
class Config(RLConfig, RLConfigComponentFramework):
    test_epsilon: float = 0
    batch_size: int = 32
    memory: PriorityReplayBufferConfig = field(default_factory=lambda: PriorityReplayBufferConfig())
    epsilon: float = 0.1
    epsilon_scheduler: SchedulerConfig = field(default_factory=lambda: SchedulerConfig())
    lr: float = 0.001
    discount: float = 0.99
    target_model_update_interval: int = 1000
    enable_reward_clip: bool = False
    enable_double_dqn: bool = True
    enable_rescale: bool = False
    input_block: InputBlockConfig = field(default_factory=lambda: InputBlockConfig())
    hidden_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig())

    def set_atari_config(self):
        """dqn.py:88-101"""
        self.batch_size = 32
        self.memory.capacity = 1_000_000
        self.memory.warmup_size = 50_000
        self.input_block.image.set_dqn_block()
        self.hidden_block.set((512,))
        self.target_model_update_interval = 10000
        self.discount = 0.99
        self.lr = 0.00025
        self.epsilon_scheduler.set_linear(1.0, 0.1, 1_000_000)
        self.enable_reward_clip = True
        self.enable_double_dqn = False
        self.enable_rescale = False

    def get_name(self) -> str:
        return "DQN"

    def get_framework(self) -> str:
        return RLConfigComponentFramework.get_framework(self)


register(Config(), __name__ + ":Memory", __name__ + ":Parameter", __name__ + ":Trainer", __name__ + ":Worker", check_duplicate=False)


class Memory(RLPriorityReplayBuffer):
    pass


def build_qnetwork(config) -> QNetwork:
    """Module trees (hence state_dict keys, hence parameter files) of the reference: DQN is in_block -> hidden_block (MLP) ->
    `out_layer` (dqn/model_torch.py:17-29); Rainbow's dueling hidden block carries its own head (rainbow/model_torch.py:15-29)."""
    from simple_distributed_rl_amd.rl.models.config import DuelingNetworkConfig

    in_block = config.input_block.create_torch_block(config)
    if isinstance(config.hidden_block, DuelingNetworkConfig):
        hidden = config.hidden_block.create_torch_block(in_block.out_size, config.action_space.n, enable_noisy_dense=getattr(config, "enable_noisy_dense", False))
        return QNetwork(in_block, hidden)
    hidden = config.hidden_block.create_torch_block(in_block.out_size)
    return QNetwork(in_block, hidden, torch.nn.Linear(hidden.out_size, config.action_space.n))


class Parameter(RLParameter):
    def setup(self):
        self.np_dtype = self.config.get_dtype("np")
        self.device = torch.device(self.config.used_device_torch)
        self.q_online = build_qnetwork(self.config).to(self.device)
        self.q_target = build_qnetwork(self.config).to(self.device)
        self.q_target.eval()
        self.q_target.load_state_dict(self.q_online.state_dict())

    def call_restore(self, data: Any, from_serialized: bool = False, **kwargs) -> None:
        self.q_online.load_state_dict(data)
        self.q_target.load_state_dict(data)

    def call_backup(self, serialized: bool = False, **kwargs) -> Any:
        sd = self.q_online.state_dict()
        if serialized:  # torch_/helper.py:76-93: a CPU copy that can cross a process boundary
            return {k: v.detach().to("cpu").clone() for k, v in sd.items()}
        return sd

    def to_device(self, device):
        self.device = torch.device(device)
        self.q_online.to(self.device)
        self.q_target.to(self.device)

    def pred_q(self, state: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            return self.q_online(torch.as_tensor(np.asarray(state, dtype=self.np_dtype), device=self.device)).cpu().numpy()

    def pred_target_q(self, state: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            return self.q_target(torch.as_tensor(np.asarray(state, dtype=self.np_dtype), device=self.device)).cpu().numpy()


class Trainer(RLTrainer):
    def on_setup(self) -> None:
        self.device = require_gpu(self.config.used_device_torch)
        self.parameter.to_device(self.device)
        self.ops = TdOps(self.device)
        self.optimizer = torch.optim.Adam(self.parameter.q_online.parameters(), lr=self.config.lr)
        self.sync_count = 0
        self.np_dtype = self.config.get_dtype("np")
        self.parameter.q_online.train()

    def train(self) -> None:
        sampled = self.memory.sample()
        if sampled is None:
            return
        batches, weights, update_args = sampled
        cfg, d = self.config, self.device
        state, n_state, onehot_action, reward, undone, next_invalid = zip(*batches)
        B, A = len(batches), cfg.action_space.n
        state = torch.as_tensor(np.asarray(state, dtype=self.np_dtype), device=d)
        n_state = torch.as_tensor(np.asarray(n_state, dtype=self.np_dtype), device=d)
        action = torch.as_tensor(np.argmax(np.asarray(onehot_action), axis=1).astype(np.int32), device=d)
        reward_t = torch.as_tensor(np.asarray(reward, dtype=np.float32), device=d)
        undone_t = torch.as_tensor(np.asarray(undone, dtype=np.float32), device=d)
        w = torch.as_tensor(np.asarray(weights, dtype=np.float32), device=d)
        inv = invalid_mask(next_invalid, (B, A), d)

        with torch.no_grad():  # dqn.py:154-165
            q_tg_next = self.parameter.q_target(n_state)
            q_on_next = self.parameter.q_online(n_state) if cfg.enable_double_dqn else None
        target = self.ops.dqn_target(q_on_next, q_tg_next, reward_t, undone_t, inv, cfg.discount, cfg.enable_double_dqn, cfg.enable_rescale, True)

        q = self.parameter.q_online(state)  # model_torch.py:111-113
        _, loss, grad, priorities = self.ops.huber(target, q, action, w)
        self.optimizer.zero_grad()
        q.backward(grad)
        self.optimizer.step()
        self.info["loss"] = float(loss.item())
        self.memory.update(update_args, priorities.cpu().numpy(), self.train_count)  # :121-122

        if self.train_count % cfg.target_model_update_interval == 0:  # :125-127 (fires at 0 too)
            self.parameter.q_target.load_state_dict(self.parameter.q_online.state_dict())
            self.sync_count += 1
        self.info["sync"] = self.sync_count
        self.train_count += 1


class Worker(RLWorker):
    def on_setup(self, worker, context) -> None:
        self.epsilon_sch = self.config.epsilon_scheduler.create(self.config.epsilon)

    def policy(self, worker) -> int:
        invalid_actions = worker.invalid_actions
        epsilon = self.epsilon_sch.update(self.step_in_training).to_float() if self.training else self.config.test_epsilon
        if random.random() < epsilon:
            action = random.choice([a for a in range(self.config.action_space.n) if a not in invalid_actions])
        else:
            q = self.parameter.pred_q(worker.state[np.newaxis, ...])[0]
            q[invalid_actions] = -np.inf
            action = int(np.argmax(q))
        self.info["epsilon"] = epsilon
        return action

    def on_step(self, worker):
        if not self.training:
            return
        reward = worker.reward
        if self.config.enable_reward_clip:
            reward = -1 if reward < 0 else (1 if reward > 0 else 0)
        self.memory.add([worker.state, worker.next_state, worker.get_onehot_action(), reward, int(not worker.terminated), worker.next_invalid_actions])
