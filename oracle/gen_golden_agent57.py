"""oracle/gen_golden_agent57.py -- TEST INFRASTRUCTURE ONLY.  Golden vectors for the NGU / Agent57_light rows
 recorded from the imported reference (run through oracle/gen_golden.py --only agent57).
Everything saved is data: inputs and the reference's outputs."""
import collections
import os
import random
import types

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _cfg(**kw):
    from srl.algorithms import agent57_light

    c = agent57_light.Config()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# ----------------------------------------------------------------------------------------
# episodic novelty (agent57_light.py:473-513): scripted embedding sequences -> rewards
# ----------------------------------------------------------------------------------------
def gen_episodic():
    from srl.algorithms.agent57_light.agent57_light import Worker

    rng = np.random.default_rng(31)
    cases = {}
    for name, D, T, cap, k, dup in [
        ("d32_t300", 32, 300, 30000, 10, 0.1),
        ("d32_cap64", 32, 200, 64, 10, 0.05),  # the deque drops its oldest entries
        ("d8_k3", 8, 40, 1000, 3, 0.3),
        ("d32_alldup", 32, 12, 100, 10, 1.0),  # every distance 0 -> the mean==0 branch
    ]:
        emb = np.maximum(rng.standard_normal((T, D)), 0).astype(np.float32)  # relu-like, as the emb block emits
        for t in range(1, T):  # revisits: exact duplicates and near-duplicates (inside the cluster distance)
            u = rng.random()
            if u < dup:
                emb[t] = emb[rng.integers(0, t)]
            elif u < 1.5 * dup and dup < 1:
                emb[t] = emb[rng.integers(0, t)] + (rng.standard_normal(D) * 1e-3).astype(np.float32)
        cfg = _cfg(episodic_count_max=k, episodic_memory_capacity=cap)
        holder = {}
        fake = types.SimpleNamespace(
            config=cfg,
            parameter=types.SimpleNamespace(predict_emb=lambda s, h=holder: h["e"][np.newaxis, ...]),
            episodic_memory=collections.deque(maxlen=cap),
        )
        out = np.zeros(T, np.float64)
        for t in range(T):
            holder["e"] = emb[t]
            out[t] = float(Worker._calc_episodic_reward(fake, None))
        cases[name] = (emb, out, cap, k)
        print(f"ngu_episodic {name}: reward range [{out.min():.5f}, {out.max():.5f}]")
    save = dict(epsilon=np.float64(cfg.episodic_epsilon), cluster_distance=np.float64(cfg.episodic_cluster_distance),
                pseudo_counts=np.float64(cfg.episodic_pseudo_counts), names=np.array(list(cases)))
    for name, (emb, out, cap, k) in cases.items():
        save[name + ".emb"], save[name + ".reward"] = emb, out
        save[name + ".capacity"], save[name + ".k"] = np.int64(cap), np.int64(k)
    np.savez_compressed(os.path.join(OUT, "ngu_episodic.npz"), **save)


# ----------------------------------------------------------------------------------------
# lifelong novelty (agent57_light.py:515-529)
# ----------------------------------------------------------------------------------------
def gen_lifelong():
    from srl.algorithms.agent57_light.agent57_light import Worker

    rng = np.random.default_rng(32)
    n, D = 64, 128
    tgt = rng.standard_normal((n, D)).astype(np.float32)
    scale = np.concatenate([np.zeros(4), rng.random(n - 4) * 3]).astype(np.float32)  # a few exact matches, many > L
    trn = tgt + scale[:, None] * rng.standard_normal((n, D)).astype(np.float32)
    cfg = _cfg()
    out = np.zeros(n, np.float64)
    for i in range(n):
        fake = types.SimpleNamespace(
            config=cfg,
            parameter=types.SimpleNamespace(predict_lifelong_target=lambda s, i=i: tgt[i : i + 1], predict_lifelong_train=lambda s, i=i: trn[i : i + 1]),
        )
        out[i] = float(Worker._calc_lifelong_reward(fake, None))
    np.savez_compressed(os.path.join(OUT, "ngu_lifelong.npz"), target=tgt, train=trn, reward=out, lifelong_max=np.float64(cfg.lifelong_max))
    print(f"ngu_lifelong: range [{out.min():.4f}, {out.max():.4f}], clipped {int((out == cfg.lifelong_max).sum())}")


# ----------------------------------------------------------------------------------------
# sliding-window UCB meta-controller (agent57_light.py:317-353)
# ----------------------------------------------------------------------------------------
def gen_ucb():
    from srl.algorithms.agent57_light.agent57_light import Worker

    cfg = _cfg(actor_num=8, ucb_window_size=20, ucb_epsilon=0.2, ucb_beta=1.0)
    rng = np.random.default_rng(33)
    episodes = 120
    ep_rewards = rng.integers(-3, 6, episodes).astype(np.float64)
    random.seed(77)
    state0 = random.getstate()
    fake = types.SimpleNamespace(config=cfg, actor_index=-1, ucb_recent=[], ucb_actors_count=[1] * cfg.actor_num, ucb_actors_reward=[0.0] * cfg.actor_num,
                                 episode_reward=0.0)
    idx = np.zeros(episodes, np.int64)
    for e in range(episodes):
        fake.actor_index = Worker._calc_actor_index(fake)
        idx[e] = fake.actor_index
        fake.episode_reward = float(ep_rewards[e])
    np.savez_compressed(os.path.join(OUT, "agent57_ucb.npz"), actor_num=np.int64(8), window=np.int64(20), ucb_epsilon=np.float64(0.2), ucb_beta=np.float64(1.0),
                        seed=np.int64(77), episode_rewards=ep_rewards, actor_index=idx)
    print(f"agent57_ucb: {episodes} episodes, histogram {np.bincount(idx, minlength=8).tolist()}")


# ----------------------------------------------------------------------------------------
# calc_target_q (agent57_light.py:218-268) with scripted Q tables
# ----------------------------------------------------------------------------------------
def gen_target():
    from srl.algorithms.agent57_light.agent57_light import CommonInterfaceParameter
    from srl.rl import functions as F

    rng = np.random.default_rng(34)
    B, A = 40, 5
    discount_list = np.array(F.create_discount_list(32), np.float64)
    for name, double_dqn, rescale, with_inv in [("double", True, False, False), ("single_inv", False, False, True), ("double_rescale_inv", True, True, True)]:
        q_on = (rng.standard_normal((B, A)) * 2).astype(np.float32)
        q_tg = (q_on + 0.3 * rng.standard_normal((B, A))).astype(np.float32)
        rewards = (rng.standard_normal(B)).astype(np.float32)
        dones = rng.integers(0, 2, B).astype(np.float32)  # the reference's "done" is int(not terminated)
        actor = rng.integers(0, 32, B)
        disc = np.array([discount_list[a] for a in actor], np.float32)
        inv_idx, inv_act = [], []
        invalid = np.zeros((B, A), bool)
        if with_inv:
            for i in range(B):
                for a in rng.choice(A, size=int(rng.integers(0, 3)), replace=False):
                    inv_idx.append(i)
                    inv_act.append(int(a))
                    invalid[i, a] = True
        cfg = _cfg(enable_double_dqn=double_dqn, enable_rescale=rescale)
        fake = types.SimpleNamespace(config=cfg, predict_q_ext_target=lambda x: q_tg.copy(), predict_q_ext_online=lambda x: q_on.copy(),
                                     predict_q_int_target=lambda x: q_tg.copy(), predict_q_int_online=lambda x: q_on.copy())
        tgt = CommonInterfaceParameter.calc_target_q(fake, True, rewards, None, None, None, None, None, inv_idx, inv_act, dones, disc)
        np.savez_compressed(os.path.join(OUT, f"agent57_light_target_{name}.npz"), double_dqn=np.int64(double_dqn), rescale=np.int64(rescale), q_online=q_on,
                            q_target=q_tg, rewards=rewards, dones=dones, discount=disc, invalid=invalid, target=np.asarray(tgt))
        print(f"agent57_light_target_{name}: dtype {np.asarray(tgt).dtype}, range [{tgt.min():.4f}, {tgt.max():.4f}]")


# ----------------------------------------------------------------------------------------
# one full Trainer.train() (agent57_light/model_torch.py:263-443)
# ----------------------------------------------------------------------------------------
def gen_train_step():
    import torch

    import srl
    from srl.algorithms import agent57_light
    from srl.base.context import RunContext

    from gen_golden_algo import _register_env

    _register_env()
    env_config = srl.EnvConfig("TinyImageEnvGolden", kwargs=dict(hw=8, actions=4))
    cfg = agent57_light.Config(batch_size=16, actor_num=8, target_model_update_interval=5, lr_ext=0.001, lr_int=0.002)
    cfg.window_length = 4
    cfg.memory.warmup_size = 16
    cfg.memory.capacity = 1000
    cfg.memory.compress = False
    cfg.memory.set_proportional(alpha=0.5, beta_initial=0.4, beta_steps=1000)
    cfg.hidden_block.set_dueling_network((32,))
    cfg.set_torch()
    env = env_config.make()
    cfg.setup(env)
    torch.manual_seed(5)
    random.seed(5)
    parameter = cfg.make_parameter()
    with torch.no_grad():
        for net in (parameter.q_ext_target, parameter.q_int_target):
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
    memory = cfg.make_memory()
    trainer = cfg.make_trainer(parameter, memory)
    trainer.setup(RunContext())

    rng = np.random.default_rng(6)
    obs_shape = tuple(cfg.observation_space.shape)
    A, n_items = 4, 48
    eye = np.identity(A, dtype=np.float32)
    for _ in range(n_items):
        a, pa = int(rng.integers(0, A)), int(rng.integers(0, A))
        memory.add([
            rng.random(obs_shape, dtype=np.float32), rng.random(obs_shape, dtype=np.float32), eye[a], [], float(rng.integers(-1, 2)),
            np.float32(rng.random() * 3), int(rng.random() < 0.8), eye[pa], float(rng.integers(-1, 2)), np.float32(rng.random() * 3), int(rng.integers(0, 8)),
        ], None)
    mem = memory.memory
    mem.update([i + mem.capacity - 1 for i in range(n_items)], rng.random(n_items).astype(np.float32))

    nets = dict(q_ext=parameter.q_ext_online, q_int=parameter.q_int_online, q_ext_target=parameter.q_ext_target, q_int_target=parameter.q_int_target,
                emb=parameter.emb_network, lifelong_target=parameter.lifelong_target, lifelong_train=parameter.lifelong_train)
    before = {n: {k: v.detach().clone().numpy() for k, v in m.state_dict().items()} for n, m in nets.items()}
    rec = {}
    _sample, _update = memory.sample, memory.update

    def sample(*a, **k):
        out = _sample(*a, **k)
        rec["batches"], rec["weights"], rec["update_args"] = out
        return out

    def update(update_args, priorities, step):
        rec["priorities"] = np.asarray(priorities).copy()
        return _update(update_args, priorities, step)

    memory.sample, memory.update = sample, update
    _uq = trainer._update_q
    tds = []

    def uq(*a, **k):
        td, loss = _uq(*a, **k)
        tds.append(np.asarray(td).copy())
        return td, loss

    trainer._update_q = uq
    trainer.train_count = 1  # not a target-sync step
    trainer.train()
    after = {n: {k: v.detach().clone().numpy() for k, v in m.state_dict().items()} for n, m in nets.items()}

    b = rec["batches"]
    save = dict(
        states=np.array([x[0] for x in b], np.float32), n_states=np.array([x[1] for x in b], np.float32),
        actions=np.array([int(np.argmax(x[2])) for x in b], np.int32), rewards_ext=np.array([x[4] for x in b], np.float32),
        rewards_int=np.array([x[5] for x in b], np.float32), dones=np.array([x[6] for x in b], np.float32),
        prev_actions=np.array([int(np.argmax(x[7])) for x in b], np.int32), prev_rewards_ext=np.array([x[8] for x in b], np.float32),
        prev_rewards_int=np.array([x[9] for x in b], np.float32), actor_idx=np.array([x[10] for x in b], np.int32),
        weights=np.asarray(rec["weights"]), td_ext=tds[0], td_int=tds[1], priorities=rec["priorities"],
        ext_loss=np.float32(trainer.info["ext_loss"]), int_loss=np.float32(trainer.info["int_loss"]), emb_loss=np.float32(trainer.info["emb_loss"]),
        lifelong_loss=np.float32(trainer.info["lifelong_loss"]), lr_ext=np.float64(cfg.lr_ext), lr_int=np.float64(cfg.lr_int), episodic_lr=np.float64(cfg.episodic_lr),
        lifelong_lr=np.float64(cfg.lifelong_lr), actor_num=np.int64(8), hw=np.int64(8), n_actions=np.int64(A), hidden=np.int64(32),
    )
    for n in nets:
        for k, v in before[n].items():
            save[f"before.{n}.{k}"] = v
        if not n.endswith("_target") or n == "lifelong_target":
            for k, v in after[n].items():
                save[f"after.{n}.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "train_step_agent57_light.npz"), **save)
    print("train_step_agent57_light:", {k: round(float(v), 6) for k, v in trainer.info.items() if "loss" in k})


AGENT57_GENERATORS = dict(ngu_episodic=gen_episodic, ngu_lifelong=gen_lifelong, agent57_ucb=gen_ucb, agent57_target=gen_target, agent57_train_step=gen_train_step)


# ----------------------------------------------------------------------------------------
# Agent57 (LSTM): calc_target_q (agent57.py:301-379) with scripted Q tensors
# ----------------------------------------------------------------------------------------
def gen_a57_target():
    from srl.algorithms.agent57 import agent57 as a57
    from srl.rl import functions as F

    rng = np.random.default_rng(41)
    dlist = np.array(F.create_discount_list(32), np.float64)
    for name, B, S, A, double_dqn, rescale, h, with_inv in [("s5_double", 16, 5, 4, True, False, 1.0, False), ("s9_single_inv_h095", 12, 9, 5, False, False, 0.95, True),
                                                            ("s4_double_rescale_inv", 10, 4, 3, True, True, 0.9, True), ("s1_double", 8, 1, 4, True, False, 1.0, False)]:
        q = (rng.standard_normal((B, S + 1, A)) * 2).astype(np.float32)
        qt = (q + 0.3 * rng.standard_normal((B, S + 1, A))).astype(np.float32)
        # make the greedy action coincide with the taken action often (otherwise every retrace coefficient is 0)
        actions = np.where(rng.random((B, S)) < 0.6, np.argmax(q[:, 1:, :], axis=2), rng.integers(0, A, (B, S))).astype(np.int32)
        onehot = np.identity(A, dtype=np.float32)[actions]
        action_q = np.take_along_axis(q[:, :-1, :], actions[..., None], axis=2)[..., 0]
        rewards = rng.standard_normal((B, S)).astype(np.float32)
        dones = (rng.random((B, S)) < 0.85).astype(np.float32)
        actor = rng.integers(0, 32, B)
        disc = np.array([dlist[a] for a in actor], np.float32)
        i1, i2, i3 = [], [], []
        invalid = np.zeros((B, S, A), bool)
        if with_inv:
            for b in range(B):
                for t in range(S):
                    for a in rng.choice(A, size=int(rng.integers(0, 2)), replace=False):
                        i1.append(b), i2.append(t), i3.append(int(a))
                        invalid[b, t, a] = True
        cfg = a57.Config(enable_double_dqn=double_dqn, enable_rescale=rescale, retrace_h=h, sequence_length=S, batch_size=B)
        fake = types.SimpleNamespace(config=cfg)
        tgt = a57.CommonInterfaceParameter.calc_target_q(fake, q.copy(), qt.copy(), action_q.copy(), rewards, onehot, dones, i1, i2, i3, disc)
        np.savez_compressed(os.path.join(OUT, f"agent57_target_{name}.npz"), double_dqn=np.int64(double_dqn), rescale=np.int64(rescale), retrace_h=np.float64(h), q=q,
                            q_target=qt, actions=actions, rewards=rewards, dones=dones, discounts=disc, invalid=invalid, target=np.asarray(tgt))
        print(f"agent57_target_{name}: {np.asarray(tgt).dtype} {np.asarray(tgt).shape} range [{np.min(tgt):.3f}, {np.max(tgt):.3f}]")


def _a57_config(**kw):
    import srl
    from srl.algorithms import agent57

    from gen_golden_algo import _register_env

    _register_env()
    cfg = agent57.Config(batch_size=8, actor_num=4, target_model_update_interval=5, lr_ext=0.001, lr_int=0.002, lstm_units=16, burnin=2, sequence_length=3, **kw)
    cfg.window_length = 1
    cfg.memory.warmup_size = 8
    cfg.memory.capacity = 1000
    cfg.memory.compress = False
    cfg.memory.set_proportional(alpha=0.5, beta_initial=0.4, beta_steps=1000)
    cfg.hidden_block.set_dueling_network((16,))
    cfg.set_torch()
    return cfg


# ----------------------------------------------------------------------------------------
# the sequence items the reference Agent57 worker emits (window shifting, dummy-state padding at episode end,
# stored LSTM states, UCB actor choice) for a recorded trajectory
# ----------------------------------------------------------------------------------------
def gen_a57_rollout():
    import srl
    import torch

    cfg = _a57_config(enable_intrinsic_reward=False)
    env_config = srl.EnvConfig("TinyImageEnvGolden", kwargs=dict(hw=8, actions=4, ep_len=4, seed=13))
    runner = srl.Runner(env_config, cfg)
    runner.set_seed(6)
    torch.manual_seed(6)
    runner.rollout(max_steps=14)
    env = runner.env.unwrapped if hasattr(runner.env, "unwrapped") else runner.env.env
    log = env.log
    mem = runner.memory.memory
    items = [mem.tree.data[i] for i in range(mem.size)]
    sd = {k: v.detach().numpy() for k, v in runner.parameter.q_ext_online.state_dict().items()}
    sdi = {k: v.detach().numpy() for k, v in runner.parameter.q_int_online.state_dict().items()}
    save = dict(
        frames=np.array([l[0].reshape(-1) for l in log], np.uint8), env_actions=np.array([l[1] for l in log], np.int32), env_rewards=np.array([l[2] for l in log], np.float32),
        env_terminated=np.array([l[3] for l in log], np.uint8),
        item_states=np.array([np.asarray(it[0], np.float32) for it in items]), item_actions=np.array([np.argmax(np.asarray(it[1]), axis=1) for it in items], np.int32),
        item_rewards_ext=np.array([it[2] for it in items], np.float32), item_rewards_int=np.array([it[3] for it in items], np.float32),
        item_dones=np.array([it[4] for it in items], np.float32), item_actor=np.array([it[5] for it in items], np.int32),
        item_h_ext=np.array([it[7][0] for it in items], np.float32), item_c_ext=np.array([it[7][1] for it in items], np.float32),
        item_h_int=np.array([it[8][0] for it in items], np.float32), item_c_int=np.array([it[8][1] for it in items], np.float32),
        seed=np.int64(6),
    )
    for k, v in sd.items():
        save["q_ext." + k] = v
    for k, v in sdi.items():
        save["q_int." + k] = v
    np.savez_compressed(os.path.join(OUT, "rollout_items_agent57.npz"), **save)
    print(f"rollout_items_agent57: {len(log)} env records, {len(items)} items, actors {sorted(set(save['item_actor'].tolist()))}")


# ----------------------------------------------------------------------------------------
# one full Trainer.train() of Agent57 (agent57/model_torch.py:273-493)
# ----------------------------------------------------------------------------------------
def gen_a57_train_step():
    import torch

    import srl
    from srl.base.context import RunContext

    cfg = _a57_config()
    env = srl.EnvConfig("TinyImageEnvGolden", kwargs=dict(hw=8, actions=4)).make()
    cfg.setup(env)
    torch.manual_seed(8)
    random.seed(8)
    parameter = cfg.make_parameter()
    with torch.no_grad():
        for net in (parameter.q_ext_target, parameter.q_int_target):
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
    memory = cfg.make_memory()
    trainer = cfg.make_trainer(parameter, memory)
    trainer.setup(RunContext())
    rng = np.random.default_rng(9)
    obs_shape = tuple(cfg.observation_space.shape)
    A, n_items, L, U = 4, 24, cfg.burnin + cfg.sequence_length + 1, cfg.lstm_units
    eye = np.identity(A, dtype=int)
    for _ in range(n_items):
        memory.add([
            [rng.random(obs_shape, dtype=np.float32) for _ in range(L)], [eye[int(rng.integers(0, A))] for _ in range(L)],
            [float(rng.integers(-1, 2)) for _ in range(L)], [float(np.float32(rng.random() * 3)) for _ in range(L)],
            [int(rng.random() < 0.85) for _ in range(cfg.sequence_length)], int(rng.integers(0, 4)), [[] for _ in range(cfg.sequence_length)],
            [(rng.standard_normal((1, U)) * 0.3).astype(np.float32), (rng.standard_normal((1, U)) * 0.3).astype(np.float32)],
            [(rng.standard_normal((1, U)) * 0.3).astype(np.float32), (rng.standard_normal((1, U)) * 0.3).astype(np.float32)],
        ], None)
    mem = memory.memory
    mem.update([i + mem.capacity - 1 for i in range(n_items)], rng.random(n_items).astype(np.float32))
    nets = dict(q_ext=parameter.q_ext_online, q_int=parameter.q_int_online, q_ext_target=parameter.q_ext_target, q_int_target=parameter.q_int_target,
                emb=parameter.emb_network, lifelong_target=parameter.lifelong_target, lifelong_train=parameter.lifelong_train)
    before = {n: {k: v.detach().clone().numpy() for k, v in m.state_dict().items()} for n, m in nets.items()}
    rec = {}
    _sample, _update = memory.sample, memory.update

    def sample(*a, **k):
        out = _sample(*a, **k)
        rec["batches"], rec["weights"], rec["update_args"] = out
        return out

    def update(update_args, priorities, step):
        rec["priorities"] = np.asarray(priorities).copy()
        return _update(update_args, priorities, step)

    memory.sample, memory.update = sample, update
    _tq = trainer._train_q
    tds = []

    def tq(*a, **k):
        td, loss = _tq(*a, **k)
        tds.append(np.asarray(td).copy())
        return td, loss

    trainer._train_q = tq
    trainer.train_count = 1
    trainer.train()
    after = {n: {k: v.detach().clone().numpy() for k, v in m.state_dict().items()} for n, m in nets.items()}
    b = rec["batches"]
    save = dict(
        states=np.array([x[0] for x in b], np.float32), actions=np.array([np.argmax(np.asarray(x[1]), axis=1) for x in b], np.int32),
        rewards_ext=np.array([x[2] for x in b], np.float32), rewards_int=np.array([x[3] for x in b], np.float32), dones=np.array([x[4] for x in b], np.float32),
        actor_idx=np.array([x[5] for x in b], np.int32), h_ext=np.array([x[7][0] for x in b], np.float32), c_ext=np.array([x[7][1] for x in b], np.float32),
        h_int=np.array([x[8][0] for x in b], np.float32), c_int=np.array([x[8][1] for x in b], np.float32), weights=np.asarray(rec["weights"]),
        td_ext=tds[0], td_int=tds[1], priorities=rec["priorities"], ext_loss=np.float32(trainer.info["ext_loss"]), int_loss=np.float32(trainer.info["int_loss"]),
        emb_loss=np.float32(trainer.info["emb_loss"]), lifelong_loss=np.float32(trainer.info["lifelong_loss"]), lr_ext=np.float64(cfg.lr_ext), lr_int=np.float64(cfg.lr_int),
        episodic_lr=np.float64(cfg.episodic_lr), lifelong_lr=np.float64(cfg.lifelong_lr), actor_num=np.int64(4), n_actions=np.int64(A), burnin=np.int64(cfg.burnin),
        sequence_length=np.int64(cfg.sequence_length), lstm_units=np.int64(U),
    )
    for n in nets:
        for k, v in before[n].items():
            save[f"before.{n}.{k}"] = v
        if not n.endswith("_target") or n == "lifelong_target":
            for k, v in after[n].items():
                save[f"after.{n}.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "train_step_agent57.npz"), **save)
    print("train_step_agent57:", {k: round(float(v), 6) for k, v in trainer.info.items() if "loss" in k})


AGENT57_GENERATORS.update(agent57_seq_target=gen_a57_target, agent57_rollout=gen_a57_rollout, agent57_seq_train_step=gen_a57_train_step)


# ----------------------------------------------------------------------------------------
# rank-based memory (srl/rl/memories/priority_memories/rankbased_memory.py) scripted trace
# ----------------------------------------------------------------------------------------
def gen_rankbased():
    from srl.rl.memories.priority_memories.rankbased_memory import RankBasedMemory

    rng = np.random.default_rng(51)
    cap, alpha = 300, 0.6
    mem = RankBasedMemory(cap, alpha, 0.4, 1000)
    np.random.seed(123)
    adds = rng.permutation(1000)[:460].astype(np.float32) / 7  # distinct priorities; 440 adds > capacity: the ring wraps
    ops = []
    k = 0
    for rnd in range(12):
        n_add = 60 if rnd < 5 else 20
        for _ in range(n_add):
            mem.add(int(k), float(adds[k]))
            k += 1
        B = 16
        batches, weights, idx = mem.sample(B, 100 * rnd)
        new_p = (rng.permutation(5000)[:B].astype(np.float32) + 2000) / 3  # distinct from everything stored
        mem.update(idx, new_p)
        ops.append((n_add, np.asarray(batches), np.asarray(weights), np.asarray(idx), new_p))
    np.savez_compressed(os.path.join(OUT, "rankbased_trace.npz"), capacity=np.int64(cap), alpha=np.float64(alpha), beta_initial=np.float64(0.4), beta_steps=np.int64(1000),
                        seed=np.int64(123), add_priorities=adds[:k], n_add=np.array([o[0] for o in ops]), batches=np.array([o[1] for o in ops]),
                        weights=np.array([o[2] for o in ops]), indices=np.array([o[3] for o in ops]), new_priorities=np.array([o[4] for o in ops]),
                        final_priorities=mem.priorities.copy())
    print(f"rankbased_trace: {k} adds, 12 samples of 16")


AGENT57_GENERATORS.update(rankbased=gen_rankbased)


# ----------------------------------------------------------------------------------------
# rank-based linear memory (srl/rl/memories/priority_memories/rankbased_memory_linear.py) scripted trace
# ----------------------------------------------------------------------------------------
def gen_rankbased_linear():
    import random

    from srl.rl.memories.priority_memories.rankbased_memory_linear import RankBasedMemoryLinear

    rng = np.random.default_rng(77)
    cap, alpha = 200, 0.8
    mem = RankBasedMemoryLinear(cap, alpha, 0.4, 1000)
    random.seed(321)
    adds = rng.permutation(4000)[:420].astype(np.float64) / 9  # distinct priorities; more adds than capacity: the lowest ones drop out
    ops, k = [], 0
    for rnd in range(10):
        n_add = 90 if rnd < 3 else 15
        for _ in range(n_add):
            mem.add(int(k), float(adds[k]))
            k += 1
        B = 12
        batches, weights, upd = mem.sample(B, 120 * rnd)
        new_p = (rng.permutation(9000)[:B].astype(np.float64) + 5000) / 11  # distinct from everything stored
        mem.update(upd, new_p)
        ops.append((n_add, np.asarray(batches), np.asarray(weights), new_p, mem.length()))
    np.savez_compressed(os.path.join(OUT, "rankbased_linear_trace.npz"), capacity=np.int64(cap), alpha=np.float64(alpha), beta_initial=np.float64(0.4),
                        beta_steps=np.int64(1000), seed=np.int64(321), add_priorities=adds[:k], n_add=np.array([o[0] for o in ops]),
                        batches=np.array([o[1] for o in ops]), weights=np.array([o[2] for o in ops]), new_priorities=np.array([o[3] for o in ops]),
                        lengths=np.array([o[4] for o in ops]), final_keys=np.array([m[0] for m in mem.memory]), final_items=np.array([m[1] for m in mem.memory]),
                        max_priority=np.float64(mem.max_priority))
    print(f"rankbased_linear_trace: {k} adds, 10 samples of 12")


AGENT57_GENERATORS.update(rankbased_linear=gen_rankbased_linear)


# ----------------------------------------------------------------------------------------
# episode replay buffer (srl/rl/memories/episode_replay_buffer.py) scripted trace
# ----------------------------------------------------------------------------------------
def gen_episode_buffer():
    import random

    from srl.rl.memories.episode_replay_buffer import EpisodeReplayBuffer

    rng = np.random.default_rng(12)
    kw = dict(batch_size=4, capacity=120, warmup_size=20, compress=True, prefix_size=2, suffix_size=1, skip_head=1, skip_tail=1, sequential_stride=2)
    mem = EpisodeReplayBuffer(**kw)
    random.seed(99)
    lengths = rng.integers(5, 30, size=14)
    out_sample, out_seq, out_steps, out_len = [], [], [], []
    step_id = 0
    for ep, L in enumerate(lengths):
        steps = [[int(step_id + t), int(ep)] for t in range(int(L))]
        step_id += int(L)
        if ep % 2 == 0:
            mem.add(steps)
        else:
            mem.add(*mem.serialize(steps), serialized=True)
        out_len.append(mem.length())
        if ep >= 3:
            b = mem.sample()
            out_sample.append(np.asarray(b)[..., 0])
            s = mem.sample_sequential(dummy_step=[-1, -1])
            out_seq.append(np.asarray(s)[..., 0])
            out_steps.append(np.asarray(mem.sample_steps())[:, 0][:5])
    np.savez_compressed(os.path.join(OUT, "episode_buffer_trace.npz"), seed=np.int64(99), lengths=lengths, total=np.array(out_len), sample=np.array(out_sample),
                        sequential=np.array(out_seq), steps_head=np.array(out_steps), **{k: np.int64(v) for k, v in kw.items()})
    print(f"episode_buffer_trace: {len(lengths)} episodes")


AGENT57_GENERATORS.update(episode_buffer=gen_episode_buffer)
