"""Agent57 plugin (srl/algorithms/agent57/agent57.py:55-753, model_torch.py:18-493), registered as "Agent57:torch":
LSTM Q-networks (extrinsic + intrinsic) with UVFA inputs, burn-in + sequence replay with stored recurrent states,
greedy-policy retrace targets, NGU intrinsic reward, sliding-window UCB over the actor family.

Worker: the reference's host logic (window of burnin + sequence_length + 1 steps shifted every step, dummy-state
padding after the episode end, recurrent state captured at the window head); the intrinsic reward runs on the GPU
(`srlx_ngu_*`, shared with Agent57_light).  Trainer: batch -> GPU once, burn-in and target passes without grad,
one online pass with grad, then ONE libsrlx kernel per Q-network for the sequence target + Huber loss + gradient
seed + mean TD error (`srlx_agent57_seq_td`), priorities in `srlx_agent57_priority`."""
import random
from dataclasses import dataclass, field
from typing import Any

import numpy as np
import torch
import torch.nn as nn

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.base.rl.algorithms.base_dqn import RLConfig, RLWorker
from simple_distributed_rl_amd.base.rl.parameter import RLParameter
from simple_distributed_rl_amd.base.rl.registration import register
from simple_distributed_rl_amd.base.rl.trainer import RLTrainer
from simple_distributed_rl_amd.rl import functions as funcs
from simple_distributed_rl_amd.rl.memories.priority_replay_buffer import PriorityReplayBufferConfig, RLPriorityReplayBuffer
from simple_distributed_rl_amd.rl.models.config import DuelingNetworkConfig, HiddenBlockConfig, InputBlockConfig, RLConfigComponentFramework

from . import agent57_light as _light
from ._device_ops import NguOps, TdOps, require_gpu


@dataclass
class Config(RLConfig, RLConfigComponentFramework):
    test_epsilon: float = 0
    test_beta: float = 0
    batch_size: int = 32
    memory: PriorityReplayBufferConfig = field(default_factory=lambda: PriorityReplayBufferConfig().set_proportional())
    input_block: InputBlockConfig = field(default_factory=lambda: InputBlockConfig())
    lstm_units: int = 512
    hidden_block: DuelingNetworkConfig = field(default_factory=lambda: DuelingNetworkConfig().set_dueling_network((512,)))
    lr_ext: float = 0.0001
    lr_int: float = 0.0001
    target_model_update_interval: int = 1500
    burnin: int = 5
    sequence_length: int = 5
    retrace_h: float = 1.0
    enable_double_dqn: bool = True
    enable_rescale: bool = False
    actor_num: int = 32
    ucb_window_size: int = 3600
    ucb_epsilon: float = 0.01
    ucb_beta: float = 1
    enable_intrinsic_reward: bool = True
    episodic_lr: float = 0.0005
    episodic_count_max: int = 10
    episodic_epsilon: float = 0.001
    episodic_cluster_distance: float = 0.008
    episodic_memory_capacity: int = 30000
    episodic_pseudo_counts: float = 0.1
    episodic_emb_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((32,)))
    episodic_out_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((128,)))
    lifelong_lr: float = 0.0005
    lifelong_max: float = 5.0
    lifelong_hidden_block: HiddenBlockConfig = field(default_factory=lambda: HiddenBlockConfig().set((128,)))
    input_ext_reward: bool = True
    input_int_reward: bool = False
    input_action: bool = False
    disable_int_priority: bool = False

    def set_atari_config(self):
        """agent57.py:151-172"""
        self.lr_ext = self.lr_int = 0.0001
        self.lifelong_lr = self.episodic_lr = 0.0005
        self.batch_size = 64
        self.lstm_units = 512
        self.input_block.image.set_dqn_block()
        self.hidden_block.set_dueling_network((512,))
        self.discount = 0.99
        self.burnin, self.sequence_length, self.retrace_h = 40, 80, 0.95
        self.episodic_memory_capacity = 30_000
        self.memory.set_proportional()
        self.memory.capacity, self.memory.warmup_size = 100_000, 6250
        self.target_model_update_interval = 1500

    def get_name(self) -> str:
        return "Agent57"

    def get_framework(self) -> str:
        return RLConfigComponentFramework.get_framework(self)

    def validate_params(self) -> None:
        super().validate_params()
        if not (self.burnin >= 0):
            raise ValueError(f"assert {self.burnin} >= 0")
        if not (self.sequence_length >= 1):
            raise ValueError(f"assert {self.sequence_length} >= 1")


register(Config(), __name__ + ":Memory", __name__ + ":Parameter", __name__ + ":Trainer", __name__ + ":Worker", check_duplicate=False)


class Memory(RLPriorityReplayBuffer):
    pass


class QNetwork(nn.Module):
    """in_block per step -> UVFA concat -> LSTM -> dueling head per step (model_torch.py:18-87)."""

    def __init__(self, config: Config):
        super().__init__()
        self.input_ext_reward = config.input_ext_reward
        self.input_int_reward = config.input_int_reward and config.enable_intrinsic_reward
        self.input_action = config.input_action
        self.in_block = config.input_block.create_torch_block(config)
        in_size = self.in_block.out_size + int(self.input_ext_reward) + int(self.input_int_reward) + config.actor_num
        if self.input_action:
            in_size += config.action_space.n
        self.hidden_size = config.lstm_units
        self.lstm_layer = nn.LSTM(in_size, config.lstm_units, batch_first=True)
        self.hidden_block = config.hidden_block.create_torch_block(config.lstm_units, config.action_space.n)

    def forward(self, inputs, hidden_states):
        state, reward_ext, reward_int, onehot_action, onehot_actor = inputs
        B, S = state.shape[:2]
        parts = [self.in_block(state.reshape((B * S,) + tuple(state.shape[2:]))).view(B, S, -1)]
        if self.input_ext_reward:
            parts.append(reward_ext)
        if self.input_int_reward:
            parts.append(reward_int)
        if self.input_action:
            parts.append(onehot_action)
        parts.append(onehot_actor)
        x, hidden_states = self.lstm_layer(torch.cat(parts, dim=2), hidden_states)
        return self.hidden_block(x.reshape(B * S, -1)).view(B, S, -1), hidden_states

    def get_initial_state(self, batch_size, device):
        return (torch.zeros(1, batch_size, self.hidden_size, device=device), torch.zeros(1, batch_size, self.hidden_size, device=device))


class Parameter(RLParameter):
    def setup(self):
        c = self.config
        self.np_dtype = c.get_dtype("np")
        self.device = torch.device(c.used_device_torch)
        self.beta_list = funcs.create_beta_list(c.actor_num)
        self.discount_list = funcs.create_discount_list(c.actor_num)
        self.epsilon_list = funcs.create_epsilon_list(c.actor_num)
        self.q_ext_online, self.q_ext_target = QNetwork(c).to(self.device), QNetwork(c).to(self.device)
        self.q_int_online, self.q_int_target = QNetwork(c).to(self.device), QNetwork(c).to(self.device)
        self.q_ext_target.eval()
        self.q_int_target.eval()
        self.q_ext_target.load_state_dict(self.q_ext_online.state_dict())
        self.q_int_target.load_state_dict(self.q_int_online.state_dict())
        self.emb_network = _light.EmbeddingNetwork(c).to(self.device)
        self.lifelong_target = _light.LifelongNetwork(c).to(self.device)
        self.lifelong_train = _light.LifelongNetwork(c).to(self.device)
        self.lifelong_target.eval()

    def to_device(self, device):
        self.device = torch.device(device)
        for m in (self.q_ext_online, self.q_ext_target, self.q_int_online, self.q_int_target, self.emb_network, self.lifelong_target, self.lifelong_train):
            m.to(self.device)

    def call_restore(self, data: Any, from_serialized: bool = False, **kwargs) -> None:  # model_torch.py:159-166
        self.q_ext_online.load_state_dict(data[0])
        self.q_ext_target.load_state_dict(data[0])
        self.q_int_online.load_state_dict(data[1])
        self.q_int_target.load_state_dict(data[1])
        self.emb_network.load_state_dict(data[2])
        self.lifelong_target.load_state_dict(data[3])
        self.lifelong_train.load_state_dict(data[4])

    def call_backup(self, serialized: bool = False, **kwargs):
        return [_light._backup(m, serialized) for m in (self.q_ext_online, self.q_int_online, self.emb_network, self.lifelong_target, self.lifelong_train)]

    def get_initial_hidden_state_q_ext(self):
        return self.q_ext_online.get_initial_state(1, self.device)

    def get_initial_hidden_state_q_int(self):
        return self.q_int_online.get_initial_state(1, self.device)

    def _predict(self, net, x, hidden_state):
        net.eval()
        with torch.no_grad():
            q, h = net([torch.as_tensor(np.asarray(v, np.float32), device=self.device) for v in x], hidden_state)
        return q.cpu().numpy(), h

    def predict_q_ext_online(self, x, hidden_state):
        return self._predict(self.q_ext_online, x, hidden_state)

    def predict_q_int_online(self, x, hidden_state):
        return self._predict(self.q_int_online, x, hidden_state)

    def convert_numpy_from_hidden_state(self, h):  # model_torch.py:258-262: [(1, units), (1, units)]
        return [h[0][0].cpu().numpy(), h[1][0].cpu().numpy()]


class Trainer(RLTrainer):
    def on_setup(self) -> None:
        self.device = require_gpu(self.config.used_device_torch)
        self.parameter.to_device(self.device)
        self.ops = TdOps(self.device)
        self.lib = N.lib()
        c, p = self.config, self.parameter
        self.q_ext_optimizer = torch.optim.Adam(p.q_ext_online.parameters(), lr=c.lr_ext)
        self.q_int_optimizer = torch.optim.Adam(p.q_int_online.parameters(), lr=c.lr_int)
        self.emb_optimizer = torch.optim.Adam(p.emb_network.parameters(), lr=c.episodic_lr)
        self.lifelong_optimizer = torch.optim.Adam(p.lifelong_train.parameters(), lr=c.lifelong_lr)
        self.beta_list = torch.tensor(np.array(p.beta_list, np.float32), device=self.device)
        self.discount_list = torch.tensor(np.array(p.discount_list, np.float32), device=self.device)
        self.actor_eye = torch.eye(c.actor_num, dtype=torch.float32, device=self.device)
        self.action_eye = torch.eye(c.action_space.n, dtype=torch.float32, device=self.device)
        self.sync_count = 0

    def seq_td(self, q, q_target, actions, rewards, dones, invalid, discounts, weights):
        """srlx_agent57_seq_td: returns (target [S][B], loss [1], grad_q, td_mean [B])."""
        c, d = self.config, self.device
        B, S1, A = q.shape
        S = S1 - 1
        target = torch.empty((S, B), dtype=torch.float32, device=d)
        loss = torch.empty(1, dtype=torch.float32, device=d)
        grad = torch.empty((B, S1, A), dtype=torch.float32, device=d)
        td = torch.empty(B, dtype=torch.float32, device=d)
        scratch = torch.empty(2 * B * S, dtype=torch.float32, device=d)
        keep = [t.detach().contiguous() if t is not None else None for t in (q, q_target, actions, rewards, dones, invalid, discounts, weights)]
        N.check(self.lib.srlx_agent57_seq_td(B, S, A, *[N.tptr(t) for t in keep], float(c.retrace_h), int(c.enable_double_dqn), int(c.enable_rescale), N.tptr(target),
                                             N.tptr(loss), N.tptr(grad), N.tptr(td), N.tptr(scratch), N.torch_stream_ptr()))
        self._keep = keep + [scratch]
        return target, loss, grad, td

    def _train_q(self, online, target_net, optimizer, step_rewards, hidden, in_burnin, in_steps, actions, dones, invalid, discounts, weights):
        c = self.config
        hidden_t = hidden
        with torch.no_grad():  # model_torch.py:455-463
            if c.burnin > 0:
                _, hidden = online(in_burnin, hidden)
                _, hidden_t = target_net(in_burnin, hidden_t)
            q_target, _ = target_net(in_steps, hidden_t)
        online.train()
        q, _ = online(in_steps, hidden)
        _, loss, grad, td = self.seq_td(q, q_target, actions, step_rewards, dones, invalid, discounts, weights)
        optimizer.zero_grad()
        q.backward(grad)
        optimizer.step()
        return td, loss

    def train(self) -> None:
        sampled = self.memory.sample()
        if sampled is None:
            return
        batches, weights, update_args = sampled
        c, d, p = self.config, self.device, self.parameter
        states, onehot_actions, rewards_ext, rewards_int, dones, actors, invalid_lists, hidden_ext, hidden_int = zip(*batches)
        B, A, bi, S = len(batches), c.action_space.n, c.burnin, c.sequence_length
        f32 = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32), device=d)  # noqa: E731
        states = f32(states)  # (B, burnin + S + 1, ...)
        r_ext, r_int = f32(rewards_ext).unsqueeze(-1), f32(rewards_int).unsqueeze(-1)
        act_idx = torch.as_tensor(np.argmax(np.asarray(onehot_actions), axis=2).astype(np.int64), device=d)  # (B, burnin + S + 1)
        onehot = self.action_eye[act_idx]
        actor = torch.as_tensor(np.asarray(actors, dtype=np.int64), device=d)
        actor_onehot = self.actor_eye[actor].unsqueeze(1).expand(B, bi + S + 1, c.actor_num)
        in_burnin = [states[:, :bi], r_ext[:, :bi], r_int[:, :bi], onehot[:, :bi], actor_onehot[:, :bi]]
        in_steps = [states[:, bi:], r_ext[:, bi:], r_int[:, bi:], onehot[:, bi:], actor_onehot[:, bi:]]
        step_actions = act_idx[:, bi + 1 :].to(torch.int32).contiguous()  # agent57.py:237: instep actions shifted by one
        step_r_ext, step_r_int = r_ext[:, bi + 1 :, 0].contiguous(), r_int[:, bi + 1 :, 0].contiguous()
        step_dones = f32(dones)
        inv = np.zeros((B, S, A), np.uint8)
        any_inv = False
        for b, per_step in enumerate(invalid_lists):
            for t, lst in enumerate(per_step):
                for a in lst:
                    inv[b, t, a] = 1
                    any_inv = True
        invalid = torch.from_numpy(inv).to(d) if any_inv else None
        discounts = self.discount_list[actor]
        w = f32(weights)
        hid = lambda hs: (f32([h[0] for h in hs]).permute(1, 0, 2).contiguous(), f32([h[1] for h in hs]).permute(1, 0, 2).contiguous())  # noqa: E731

        self.td_ext, ext_loss = self._train_q(p.q_ext_online, p.q_ext_target, self.q_ext_optimizer, step_r_ext, hid(hidden_ext), in_burnin, in_steps, step_actions,
                                              step_dones, invalid, discounts, w)
        self.info["ext_loss"] = float(ext_loss.item())
        self.td_int = None
        if c.enable_intrinsic_reward:
            self.td_int, int_loss = self._train_q(p.q_int_online, p.q_int_target, self.q_int_optimizer, step_r_int, hid(hidden_int), in_burnin, in_steps,
                                                  step_actions, step_dones, invalid, discounts, w)
            self.info["int_loss"] = float(int_loss.item())
            one_states, one_n_states, one_actions = states[:, bi], states[:, bi + 1], onehot[:, bi]  # model_torch.py:348-351
            p.emb_network.train()
            emb_loss = torch.nn.functional.mse_loss(p.emb_network([one_states, one_n_states]), one_actions)
            self.emb_optimizer.zero_grad()
            emb_loss.backward()
            self.emb_optimizer.step()
            self.info["emb_loss"] = float(emb_loss.item())
            with torch.no_grad():
                lifelong_target_val = p.lifelong_target(one_states)
            p.lifelong_train.train()
            lifelong_loss = torch.nn.functional.mse_loss(lifelong_target_val, p.lifelong_train(one_states))
            self.lifelong_optimizer.zero_grad()
            lifelong_loss.backward()
            self.lifelong_optimizer.step()
            self.info["lifelong_loss"] = float(lifelong_loss.item())

        use_int = c.enable_intrinsic_reward and not c.disable_int_priority  # :385-391
        _, _, priorities = self.ops.agent57_priority(self.td_ext, None, self.td_int if use_int else None, None, None, actor.to(torch.int32), self.beta_list, n_actions=A)
        self.memory.update(update_args, priorities.cpu().numpy(), self.train_count)
        if self.train_count % c.target_model_update_interval == 0:
            p.q_ext_target.load_state_dict(p.q_ext_online.state_dict())
            p.q_int_target.load_state_dict(p.q_int_online.state_dict())
            self.sync_count += 1
        self.info["sync"] = self.sync_count
        self.train_count += 1


def _ucb_tie_break_numpy(ucbs):
    return int(np.random.choice(np.where(ucbs == np.max(ucbs))[0]))  # agent57.py:510 draws from numpy's global generator


class Worker(RLWorker):
    def on_setup(self, worker, context) -> None:
        c = self.config
        self.dummy_state = np.zeros(c.observation_space.shape, dtype=np.float32)
        self.act_onehot_arr = np.identity(c.action_space.n, dtype=int)
        self.beta_list, self.epsilon_list, self.discount_list = self.parameter.beta_list, self.parameter.epsilon_list, self.parameter.discount_list
        self.ucb = _light.UcbMetaController(c.actor_num, c.ucb_window_size, c.ucb_epsilon, c.ucb_beta, tie_break=_ucb_tie_break_numpy)
        self.episode_reward = 0.0
        self.window = c.burnin + c.sequence_length + 1
        self.ngu = None
        if c.enable_intrinsic_reward:
            dev = require_gpu(str(self.parameter.device))
            self.ngu = NguOps(dev, 1, self.parameter.emb_network.emb_block.out_size, c.episodic_memory_capacity, c.episodic_count_max, c.episodic_epsilon,
                              c.episodic_cluster_distance, c.episodic_pseudo_counts)

    def on_reset(self, worker):
        c, n, L = self.config, self.config.action_space.n, self.window
        self.q_ext, self.q_int, self.q = [0] * n, [0] * n, [0] * n
        self.recent_states = [self.dummy_state for _ in range(L)]
        self.recent_actions = [self.act_onehot_arr[random.randint(0, n - 1)] for _ in range(L)]
        self.recent_rewards_ext = [0.0 for _ in range(L)]
        self.recent_rewards_int = [0.0 for _ in range(L)]
        self.recent_done = [1 for _ in range(c.sequence_length)]
        self.recent_next_invalid_actions = [[] for _ in range(c.sequence_length)]
        self.hidden_state_ext = self.parameter.get_initial_hidden_state_q_ext()
        self.hidden_state_int = self.parameter.get_initial_hidden_state_q_int()
        conv = self.parameter.convert_numpy_from_hidden_state
        self.recent_hidden_states_ext = [conv(self.hidden_state_ext) for _ in range(L)]
        self.recent_hidden_states_int = [conv(self.hidden_state_int) for _ in range(L)]
        self.recent_states.pop(0)
        self.recent_states.append(worker.state.astype(np.float32))
        self._calc_td_error = bool(self.distributed and c.memory.requires_priority())  # agent57.py:441-447
        self._history_batch = []
        if self.training:
            self.actor_index = self.ucb.next_actor(self.episode_reward)
            self.beta, self.epsilon, self.discount = self.beta_list[self.actor_index], self.epsilon_list[self.actor_index], self.discount_list[self.actor_index]
        else:
            self.actor_index, self.epsilon, self.beta = 0, c.test_epsilon, c.test_beta
        self.action = random.randint(0, n - 1)
        self.reward_ext = 0
        self.reward_int = 0
        self.onehot_actor_idx = np.identity(c.actor_num, dtype=np.float32)[self.actor_index][np.newaxis, np.newaxis, ...]
        self.episode_reward = 0.0
        if self.ngu is not None:
            self.ngu.reset()

    def policy(self, worker) -> int:
        n = self.config.action_space.n
        prev_onehot_action = np.identity(n, dtype=np.float32)[self.action][np.newaxis, np.newaxis, ...]
        in_ = [self.recent_states[-1][np.newaxis, np.newaxis, ...], np.array([[[self.reward_ext]]], np.float32), np.array([[[self.reward_int]]], np.float32),
               prev_onehot_action, self.onehot_actor_idx]
        q_ext, self.hidden_state_ext = self.parameter.predict_q_ext_online(in_, self.hidden_state_ext)
        q_int, self.hidden_state_int = self.parameter.predict_q_int_online(in_, self.hidden_state_int)
        self.q_ext, self.q_int = q_ext[0][0], q_int[0][0]
        self.q = self.q_ext + self.beta * self.q_int
        probs = funcs.calc_epsilon_greedy_probs(self.q, worker.invalid_actions, self.epsilon, n)
        self.action = funcs.random_choice_by_probs(probs)
        return self.action

    def _shift(self, state, action_onehot, r_ext, r_int, undone, next_invalid, hidden=True):
        for lst, v in ((self.recent_states, state), (self.recent_actions, action_onehot), (self.recent_rewards_ext, r_ext), (self.recent_rewards_int, r_int),
                       (self.recent_done, undone), (self.recent_next_invalid_actions, next_invalid)):
            lst.pop(0)
            lst.append(v)
        self.recent_hidden_states_ext.pop(0)
        self.recent_hidden_states_int.pop(0)
        if hidden:
            conv = self.parameter.convert_numpy_from_hidden_state
            self.recent_hidden_states_ext.append(conv(self.hidden_state_ext))
            self.recent_hidden_states_int.append(conv(self.hidden_state_int))

    def on_step(self, worker):
        c = self.config
        next_state, reward_ext = worker.next_state, worker.reward
        self.episode_reward += reward_ext
        self.reward_ext = reward_ext
        if c.enable_intrinsic_reward:
            self.info["episodic"], self.info["lifelong"], self.reward_int = _light.Worker.intrinsic_reward(self, next_state)
            self.info["reward_int"] = self.reward_int
        else:
            self.reward_int = 0.0
        self._shift(next_state, self.act_onehot_arr[self.action], reward_ext, self.reward_int, 0 if worker.terminated else 1, worker.next_invalid_actions)
        if not self.training:
            return
        self._add_memory(dict(q=self.q[self.action], reward_ext=reward_ext, reward_int=self.reward_int) if self._calc_td_error else None)
        if worker.done:  # flush the window: the remaining steps are padded with dummy states (agent57.py:583-610)
            n = c.action_space.n
            for _ in range(len(self.recent_rewards_ext) - 1):
                self._shift(self.dummy_state, self.act_onehot_arr[random.randint(0, n - 1)], 0.0, 0.0, 0, [], hidden=False)
                self._add_memory(dict(q=self.q[self.action], reward_ext=0.0, reward_int=0.0))
            if self._calc_td_error:  # Monte-Carlo initial priorities, newest first (:612-634)
                r_e = r_i = 0
                for batch, info in reversed(self._history_batch):
                    pe, pi = (funcs.inverse_rescaling(r_e), funcs.inverse_rescaling(r_i)) if c.enable_rescale else (r_e, r_i)
                    r_e, r_i = info["reward_ext"] + self.discount * pe, info["reward_int"] + self.discount * pi
                    if c.enable_rescale:
                        r_e, r_i = funcs.rescaling(r_e), funcs.rescaling(r_i)
                    priority = abs(r_e - info["q"]) if c.disable_int_priority else abs((r_e + self.beta * r_i) - info["q"])
                    self.memory.add(batch, priority)

    def _add_memory(self, calc_info):
        batch = [self.recent_states[:], self.recent_actions[:], self.recent_rewards_ext[:], self.recent_rewards_int[:], self.recent_done[:], self.actor_index,
                 self.recent_next_invalid_actions[:], self.recent_hidden_states_ext[0], self.recent_hidden_states_int[0]]  # :652-663
        if self._calc_td_error:
            self._history_batch.append([batch, calc_info])
        else:
            self.memory.add(batch, None)
