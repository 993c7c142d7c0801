"""GPU tests: `srl.Runner(...).train()` on the device engine.

The same loop that plays the plugin classes (base/run/sequence.py) is driven by the device drivers
(device/vector_runner.py): E environments per iteration, RunCallback hooks in the reference's order
(srl/base/run/callback.py:11-78), RunState counters, stop rules, weights written back into `runner.parameter`."""
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

import simple_distributed_rl_amd as srl  # noqa: E402
from simple_distributed_rl_amd.algorithms import rainbow  # noqa: E402
from simple_distributed_rl_amd.base.run.callback import RunCallback  # noqa: E402


def _atari_like(capacity=200_000, warmup=20_000, hidden=512, noisy=False):
    cfg = rainbow.Config()
    cfg.set_atari_config()  # rainbow.py:116-148
    cfg.enable_noisy_dense = noisy
    cfg.window_length = 4
    cfg.memory.capacity, cfg.memory.warmup_size = capacity, warmup
    cfg.hidden_block.set_dueling_network((hidden,))
    return cfg


def test_episode_ledger_matches_a_host_model():
    """srlx_episode_account vs a numpy restatement of env_run.py:334-352 + core_play.py:200-214 for E lanes, incl. lanes that
    skip a lock-step (reset frame only), several ring wraps and the mailbox."""
    from simple_distributed_rl_amd.device.vector_runner import EpisodeLedger

    E, cap = 1500, 64
    dev = torch.device("cuda:0")
    led = EpisodeLedger(E, dev, ring_cap=cap)
    rng = np.random.default_rng(0)
    ret, ln = np.zeros(E, np.float32), np.zeros(E, np.int64)
    book, steps = [], 0
    got = []
    for it in range(40):
        r = rng.integers(-1, 2, E).astype(np.float32)
        d = (rng.random(E) < 0.01).astype(np.uint8)
        skip = (rng.random(E) < 0.05).astype(np.uint8)
        ts = torch.as_tensor(skip, device=dev)
        led.account(torch.as_tensor(r, device=dev), torch.as_tensor(d, device=dev), ctypes_ptr(ts))
        for e in range(E):
            if skip[e]:
                continue
            steps += 1
            ret[e] = np.float32(ret[e] + r[e])
            ln[e] += 1
            if d[e]:
                book.append((float(ret[e]), int(ln[e])))
                ret[e], ln[e] = 0, 0
        led.post()
        if it % 3 == 2:
            got += led.drain()
            assert len(got) == len(book) or len(book) - len(got) <= 0
    got += led.drain()
    eps, st, rsum, lsum = led.peek(wait=True)
    assert eps == len(book) and st == steps
    assert lsum == sum(b[1] for b in book)
    np.testing.assert_allclose(rsum, sum(b[0] for b in book), rtol=1e-12)
    assert got == book  # drained every 3 lock-steps: fewer finished episodes than the ring holds, nothing lost, environment order


def ctypes_ptr(t):
    from simple_distributed_rl_amd import _native as N

    return N.tptr(t)


class _Recorder(RunCallback):
    def __init__(self, stop_at=None):
        self.seq, self.stop_at = [], stop_at
        self.steps_seen = []

    def _hit(self, name):
        self.seq.append(name)

    def on_start(self, context, **kw):
        self._hit("on_start")

    def on_end(self, context, **kw):
        self._hit("on_end")

    def on_episodes_begin(self, context, state, **kw):
        self._hit("on_episodes_begin")

    def on_episodes_end(self, context, state, **kw):
        self._hit("on_episodes_end")

    def on_episode_begin(self, context, state, **kw):
        self._hit("on_episode_begin")

    def on_episode_end(self, context, state, **kw):
        self._hit("on_episode_end")
        assert len(state.last_episode_rewards) == 1 and state.last_episode_step > 0

    def on_step_begin(self, context, state, **kw):
        self._hit("on_step_begin")

    def on_step_action_before(self, context, state, **kw):
        self._hit("on_step_action_before")

    def on_step_action_after(self, context, state, **kw):
        self._hit("on_step_action_after")
        assert state.action.shape[0] > 1  # all lanes' actions

    def on_step_end(self, context, state, **kw):
        self._hit("on_step_end")
        self.steps_seen.append(state.total_step)
        return self.stop_at is not None and state.total_step >= self.stop_at


def test_callback_protocol_on_the_device_engine():
    """The hook sequence of tests/test_plugin_surface.py::test_callback_hooks_fire, with E = 16 lanes per iteration."""
    cfg = _atari_like(capacity=16 * 64, warmup=64, hidden=64)
    cfg.batch_size = 8
    runner = srl.Runner(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=9)), cfg)
    runner.set_vector_envs(16)
    runner.set_seed(3)
    rec = _Recorder(stop_at=16 * 30)
    st = runner.train(max_steps=10**9, callbacks=[rec], train_interval=16)
    assert runner.vector_reason == "" and st.end_reason == "callback.intermediate_stop"
    assert st.total_step == 16 * 30 and rec.steps_seen == [16 * (i + 1) for i in range(30)]
    seq = rec.seq
    assert seq[:3] == ["on_start", "on_episodes_begin", "on_episode_begin"] and seq[-2:] == ["on_episodes_end", "on_end"]
    per_iter = ["on_step_begin", "on_step_action_before", "on_step_action_after", "on_step_end"]
    body = [s for s in seq if s in per_iter]
    assert body == per_iter * 30
    # episodes of 9 steps + 1 reset lock-step: lanes finish at iterations 9, 19, 29 -> 3 x 16 episodes
    assert seq.count("on_episode_end") == 48 == st.episode_count == len(st.episode_rewards_list)
    assert seq.count("on_episode_begin") in (1 + 32, 1 + 48)  # the begins of the last batch fire at the next iteration's start
    i_end = seq.index("on_episode_end")
    assert seq[i_end - 1] == "on_step_end"  # episode ends are reported after the lock-step's on_step_end
    assert st.train_count > 0 and st.trainer.train_count == st.train_count
    assert st.memory.length() > 0
    for r in st.episode_rewards_list:
        assert -9 <= r[0] <= 9


def test_runner_train_on_engine_throughput_and_writeback():
    """`srl.Runner(<84x84 env>, rainbow.Config(...)).train(max_steps=...)`: >= 1e5 env-steps/s through the hooks, learner
    updates at the configured ratio, trained weights in runner.parameter afterwards, evaluate() on the plugin path works."""
    cfg = _atari_like()
    runner = srl.Runner("SyntheticAtari-v0", cfg)
    runner.set_seed(1)
    before = {k: v.detach().clone() for k, v in runner.parameter.q_online.state_dict().items()}
    hits = []

    class Tick(RunCallback):
        def on_step_end(self, context, state, **kw):
            hits.append(state.total_step)
            return False

    runner.train(max_steps=1024 * 30, train_interval=1024, callbacks=[Tick()])  # builds the engine, fills past the warm-up, captures graphs
    assert runner.vector_reason == ""
    t0 = time.time()
    st = runner.train(max_steps=1024 * 300, train_interval=1024, callbacks=[Tick()])
    dt = time.time() - t0
    assert st.end_reason == "max_steps over." and st.total_step == 1024 * 300
    rate = st.total_step / dt
    print(f"Runner.train on the engine: {rate:,.0f} env-steps/s, {st.train_count / dt:,.0f} updates/s ")
    assert rate >= 1e5, rate
    assert st.train_count >= 250  # one update per 1024 steps once the replay is warm (20 lock-steps)
    assert st.episode_count > 1000 and len(st.episode_rewards_list) > 0
    assert st.shared_vars["env_steps_exact"] <= st.total_step
    after = runner.parameter.q_online.state_dict()
    changed = [k for k in before if not torch.equal(before[k].cpu(), after[k].cpu())]
    assert len(changed) == len(before), set(before) - set(changed)
    rewards = runner.evaluate(max_episodes=2)
    assert len(rewards) == 2


def test_stop_rules_and_train_ratio_on_the_engine():
    cfg = _atari_like(capacity=32 * 64, warmup=256, hidden=64)
    cfg.batch_size = 16
    runner = srl.Runner(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=20)), cfg)
    runner.set_vector_envs(32)
    st = runner.train(max_train_count=40, train_interval=8, train_repeat=1)
    assert st.end_reason == "max_train_count over." and 40 <= st.train_count < 40 + 4  # 4 updates are owed per 32-step iteration
    st = runner.train(max_episodes=50)
    assert st.end_reason == "episode_count over." and 50 <= st.episode_count <= 50 + 32
    st = runner.train(timeout=1.0, train_interval=32)
    assert st.end_reason == "timeout."
    st = runner.train(max_memory=1024, train_interval=32)
    assert st.end_reason == "max_memory over." and st.memory.length() >= 1024


def test_host_environments_behind_the_engine():
    """Any registered single-channel image environment runs on the engine through HostVecEnv (E host copies, frames uploaded):
    what the ring received is what the environments produced, lane by lane."""
    from simple_distributed_rl_amd.base.env import registration

    registration.register("HostFrames84", "test_runner_vector_gpu:HostFrames", check_duplicate=False)
    cfg = _atari_like(capacity=8 * 64, warmup=64, hidden=64)
    cfg.batch_size = 8
    runner = srl.Runner("HostFrames84", cfg)
    runner.set_vector_envs(8)
    st = runner.train(max_steps=8 * 40, train_interval=8)
    assert runner.vector_reason == "" and st.total_step == 320 and st.train_count > 0
    assert st.episode_count == len(st.episode_rewards_list) > 0
    for r, in st.episode_rewards_list:
        assert r == 7.0  # HostFrames pays 1 per step, episodes of 7 steps


class HostFrames:
    pass


def _define_host_frames():
    from simple_distributed_rl_amd.base.define import SpaceTypes
    from simple_distributed_rl_amd.base.env.base import EnvBase
    from simple_distributed_rl_amd.base.spaces.box import BoxSpace
    from simple_distributed_rl_amd.base.spaces.discrete import DiscreteSpace

    class _HostFrames(EnvBase):
        def __init__(self):
            super().__init__()
            self.rng = np.random.default_rng(0)

        action_space = property(lambda self: DiscreteSpace(5))
        observation_space = property(lambda self: BoxSpace((84, 84, 1), 0, 1, np.float32, SpaceTypes.GRAY_HW1))
        max_episode_steps = property(lambda self: 100)
        player_num = property(lambda self: 1)

        def _frame(self):
            return self.rng.integers(0, 256, (84, 84, 1), dtype=np.uint8).astype(np.float32) / np.float32(255)

        def reset(self, **kw):
            self.t = 0
            return self._frame()

        def step(self, action):
            self.t += 1
            return self._frame(), 1.0, self.t >= 7, False

        def backup(self, **kw):
            return None

        def restore(self, d, **kw):
            pass

    return _HostFrames


HostFrames = _define_host_frames()


def test_train_mp_on_the_engine_two_ranks_sharing_the_gpu():
    """`Runner.train_mp(actor_num, actor_devices=[...])` with the Rainbow family on GPU devices: one process per rank on
    DistributedRainbow (here 2 ranks time-sharing the test GPU over gloo; distinct GPUs would rendezvous over RCCL).  The calling
    process is the learner rank: trainer-side hooks fire, the stop rule is max_train_count, the weights come back."""
    cfg = _atari_like(capacity=2 * 16 * 40, warmup=64, hidden=32)
    cfg.batch_size = 8
    runner = srl.Runner(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(20, 20), n_actions=4, episode_len=7)), cfg)
    runner.set_vector_envs(16)
    before = {k: v.detach().clone().cpu() for k, v in runner.parameter.q_online.state_dict().items()}
    seen = []

    class TCB(RunCallback):
        def on_trainer_start(self, context, state, **kw):
            seen.append("start")

        def on_train_after(self, context, state, **kw):
            seen.append(state.train_count)

        def on_trainer_end(self, context, state, **kw):
            seen.append("end")

    st = runner.train_mp(actor_num=2, actor_devices=["cuda:0", "cuda:0"], max_train_count=30, timeout=300, callbacks=[TCB()], sync_interval_steps=4)
    assert runner.vector_reason == ""
    assert st.end_reason == "max_train_count over." and 30 <= st.train_count < 30 + 16  # the stop flag is agreed every 16 lock-steps
    assert seen[0] == "start" and seen[-1] == "end" and seen[-2] == st.train_count
    assert st.trainer_recv_q > 0 and st.sync_trainer > 0 and st.memory.length() > 64
    after = runner.parameter.q_online.state_dict()
    assert any(not torch.equal(before[k], after[k].cpu()) for k in before)


def test_uniform_replay_buffer_on_the_device():
    """a6: `memory.set_replay_buffer()` (srl/rl/memories/priority_memories/replay_buffer.py:10-55: uniform draws, all weights 1, update a no-op)
    on the engine is the alpha = 0 corner of the HBM sum-tree: every stored item weighs 1, so draws are uniform over the stored items, every
    importance weight is exactly 1 and priority updates leave the distribution unchanged."""
    cfg = _atari_like(capacity=16 * 64, warmup=128, hidden=64)
    cfg.batch_size = 32
    cfg.memory.set_replay_buffer()
    runner = srl.Runner(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=50)), cfg)
    runner.set_vector_envs(16)
    st = runner.train(max_train_count=60, train_interval=16)
    assert runner.vector_reason == "" and st.train_count >= 60
    eng = runner._vector_actor.engine
    rep = eng.replay
    assert rep.per_state()["max_priority"] == 1.0  # (|p| + eps)^0
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    hits = np.zeros(rep.capacity, np.int64)
    for _ in range(400):
        b = rep.sample_items(step, all_states=True)
        torch.cuda.synchronize()
        assert float(b.weights.min()) == 1.0 == float(b.weights.max())
        np.add.at(hits, b.indices.cpu().numpy() - (rep.capacity - 1), 1)
    live = hits[: rep.length()] if rep.length() < rep.capacity else hits
    filled = (live > 0).mean()
    assert filled > 0.95 and live.max() < 12 * live.mean()  # 12 800 draws over <= 1024 items: every item is hit (the 2 % of ring positions that hold
    # an episode's terminal frame are no items and weigh 0), nothing dominates
