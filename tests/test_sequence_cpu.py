"""CPU tests of the loop machinery shared by the plugin drivers and the device drivers (base/run/sequence.py,
base/run/hooks.py), of the engine-eligibility rules (device/vector_runner.py) and of the rank plan of train_mp
(device/mp_runner.py)."""
import time

import numpy as np
import pytest

import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import ql, rainbow
from simple_distributed_rl_amd.base.context import RunContext, RunStateActor
from simple_distributed_rl_amd.base.run.callback import RunCallback
from simple_distributed_rl_amd.base.run.hooks import HookTable
from simple_distributed_rl_amd.base.run.sequence import ActorDriver, LearnerDriver, StopRules, _TrainingDebt, run_sequence


def test_hook_table_resolves_once_and_polls_every_listener():
    log = []

    class A(RunCallback):
        def on_step_end(self, context, state, **kw):
            log.append("A")
            return True

    class B(RunCallback):
        def on_step_end(self, context, state, **kw):
            log.append("B")

        def on_episode_end(self, context, state, **kw):
            log.append("ep")

    h = HookTable([A(), B()], context=None, state=None)
    assert h.wants("on_step_end") and h.wants("on_episode_end") and not h.wants("on_step_begin", "on_train_after")
    assert h.poll("on_step_end") is True and log == ["A", "B"]  # every listener runs even after one asked to stop
    h.fire("on_episode_end")
    h.fire("on_step_begin")  # nobody listens: a no-op
    assert log == ["A", "B", "ep"]


def test_training_debt_counts_crossed_multiples():
    d = _TrainingDebt(interval=1, repeat=1)
    assert d.owed(5, 6) == 1 and d.owed(0, 1024) == 1024
    d = _TrainingDebt(interval=4, repeat=2)
    assert [d.owed(s, s + 1) for s in range(8)] == [0, 0, 0, 2, 0, 0, 0, 2]  # total_step % 4 == 0 after the increment
    assert d.owed(0, 1024) == 512 and d.owed(1023, 1024 + 1023) == 2 * 256
    d = _TrainingDebt(interval=1024, repeat=1)
    assert d.owed(0, 1024) == 1 and d.owed(1024, 2048) == 1 and d.owed(0, 512) == 0


def test_stop_rules_order_and_switches():
    c = RunContext(max_steps=10, max_train_count=5, timeout=0, max_memory=3)
    s = RunStateActor()
    s.elapsed_t0 = time.time()

    class M:
        n = 0

        def length(self):
            return self.n

    m = M()
    r = StopRules(c, counts_training=True, memory=m)
    assert r.reason(s) == ""
    m.n = 3
    assert r.reason(s) == "max_memory over."
    s.train_count = 5
    assert r.reason(s) == "max_train_count over."
    s.total_step = 10
    assert r.reason(s) == "max_steps over."
    assert StopRules(c, counts_training=False, memory=None).reason(RunStateActor(total_step=0, train_count=99)) == ""
    c2 = RunContext(timeout=0.01)
    s2 = RunStateActor()
    s2.elapsed_t0 = time.time() - 1
    assert StopRules(c2, True, None).reason(s2) == "timeout."


class _Lanes(ActorDriver):
    """A stand-in for the device actor: `lanes` counters, every lane's episode ends after `ep_len` lock-steps."""

    def __init__(self, lanes, ep_len):
        self.lanes, self.ep_len, self.t, self.log = lanes, ep_len, 0, []

    def open(self, context, state):
        state.memory = None
        state.episode_count = 0
        self.log.append("open")

    def roll_episodes(self, context, state, hooks):
        if self.t == 0:
            hooks.fire("on_episode_begin")
        return not (context.max_episodes > 0 and state.episode_count >= context.max_episodes)

    def act(self, context, state, hooks):
        state.action = list(range(self.lanes))
        hooks.fire("on_step_action_after")
        self.t += 1
        state.total_step += self.lanes

    def settle(self, context, state, hooks):
        if self.t % self.ep_len == 0:
            for _ in range(self.lanes):
                state.episode_count += 1
                state.episode_rewards_list.append([1.0])
                hooks.fire("on_episode_end")

    def close(self, context, state):
        self.log.append("close")


class _CountingLearner(LearnerDriver):
    def __init__(self, warm_after):
        self.calls, self.warm_after, self.done = [], warm_after, 0

    def open(self, context, state):
        state.trainer = self

    def update(self, count, state):
        self.calls.append(count)
        if state.total_step < self.warm_after:
            return 0
        self.done += count
        return count

    def close(self, context, state):
        pass


def test_run_sequence_with_many_lanes_per_iteration():
    """The loop itself, on stand-in drivers: step accounting by `lanes`, owed updates, warm-up, stop rules, hook order."""
    seq = []

    class CB(RunCallback):
        def on_start(self, context, **kw):
            seq.append("on_start")

        def on_end(self, context, **kw):
            seq.append("on_end")

        def on_episodes_begin(self, context, state, **kw):
            seq.append("on_episodes_begin")

        def on_episodes_end(self, context, state, **kw):
            seq.append("on_episodes_end")

        def on_step_begin(self, context, state, **kw):
            seq.append("b")

        def on_step_end(self, context, state, **kw):
            seq.append("e")

        def on_episode_end(self, context, state, **kw):
            seq.append("E")

    c = RunContext(rl_config=ql.Config(), callbacks=[CB()], max_steps=64, train_interval=4, train_repeat=1, training=True, device="CPU")
    actor, learner = _Lanes(8, 4), _CountingLearner(warm_after=24)
    st = run_sequence(c, actor, learner)
    assert st.end_reason == "max_steps over." and st.total_step == 64 and actor.log == ["open", "close"]
    assert learner.calls == [2] * 8  # 8 steps per iteration / train_interval 4
    assert st.train_count == learner.done == 2 * 6  # the first two iterations are below the warm-up
    assert st.episode_count == 16 and seq.count("E") == 16
    assert seq[:2] == ["on_start", "on_episodes_begin"] and seq[-2:] == ["on_episodes_end", "on_end"]
    body = "".join(x for x in seq if x in "beE")
    assert body == ("be" * 3 + "be" + "E" * 8) * 2
    c.max_steps, c.max_episodes = 0, 8
    st = run_sequence(c, _Lanes(8, 4), None)
    assert st.end_reason == "episode_count over." and st.episode_count == 8 and st.train_count == 0


def test_engine_eligibility_reasons():
    from simple_distributed_rl_amd.device import vector_runner as vr

    def reason(env="SyntheticAtari-v0", cfg=None, device="cuda:0", **env_kw):
        r = srl.Runner(srl.EnvConfig(env, kwargs=env_kw), cfg or atari())
        r.setup_rl_config()
        c = RunContext(rl_config=r.rl_config)
        c.used_device_torch = device
        return vr.why_not_vector(c, r.env, r.rl_config)

    def atari():
        cfg = rainbow.Config()
        cfg.set_atari_config()
        cfg.enable_noisy_dense = False
        cfg.window_length = 4
        return cfg

    assert reason() == ""
    assert "not on a GPU" in reason(device="cpu")
    cfg = atari()
    cfg.window_length = 1
    assert "window of 4" in reason(cfg=cfg)
    cfg = atari()
    cfg.hidden_block.set((512,))
    assert "dueling" in reason(cfg=cfg)
    cfg = atari()
    cfg.hidden_block.set_dueling_network((512, 512))
    assert "dueling" in reason(cfg=cfg)
    cfg = atari()
    cfg.memory.set_rankbased()
    assert "RankBased" in reason(cfg=cfg)
    cfg = atari()
    cfg.memory.set_replay_buffer()
    assert reason(cfg=cfg) == ""  # the uniform buffer is the alpha = 0 corner of the device tree
    assert "image" in reason(env="Grid")
    # rainbow.Config -> the engine's configuration
    r = srl.Runner("SyntheticAtari-v0", atari())
    r.setup_rl_config()
    d = vr.device_config_from(r.rl_config, r.env, 256, 7)
    assert (d.batch_size, d.multisteps, d.memory_capacity, d.memory_warmup_size, d.memory_alpha, d.memory_beta_initial) == (32, 3, 1_000_000, 80_000, 0.5, 0.4)
    assert (d.hidden_units, d.dueling_type, d.obs_hw, d.n_actions, d.n_envs, d.seed, d.lr) == (512, "average", (84, 84), 6, 256, 7, 0.0000625)
    on_cpu = srl.Runner("SyntheticAtari-v0", atari())
    on_cpu.set_device("CPU")
    on_cpu.rl_config.memory.warmup_size, on_cpu.rl_config.memory.capacity = 32, 1000
    on_cpu.rl_config.memory.set_replay_buffer()  # the proportional memory is device-backed: there is no CPU tree in this build
    st = on_cpu.rollout(max_steps=5)  # the same environment id is an ordinary host environment on the plugin path
    assert st.total_step == 5


def test_rank_plan_of_train_mp():
    from simple_distributed_rl_amd.device.mp_runner import plan_ranks

    p = plan_ranks("cuda:0", 2, ["cuda:0", "cuda:0"])  # a 1-GPU box: the learner hosts one actor, the other shares its GPU
    assert p == dict(devices=[0, 0], learner_acts=True, backend="gloo")
    p = plan_ranks("cuda:0", 7, [f"cuda:{i}" for i in range(1, 8)])  # BASELINE.json configs[3]: 7 actor GPUs + 1 learner GPU
    assert p == dict(devices=[0, 1, 2, 3, 4, 5, 6, 7], learner_acts=False, backend="nccl")
    p = plan_ranks("cuda:0", 2, ["cuda:0", "cuda:1"])  # 2 GPUs: rank 0 acts and learns
    assert p == dict(devices=[0, 1], learner_acts=True, backend="nccl")
