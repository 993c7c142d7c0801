"""oracle/gen_golden_algo.py -- TEST INFRASTRUCTURE ONLY.  Algorithm-level golden vectors recorded from the
imported reference (run through oracle/gen_golden.py).  Everything saved is data: inputs and the
reference's outputs."""
import os
import random
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# ----------------------------------------------------------------------------------------
# a tiny image environment registered with the reference from OUTSIDE its tree
# (srl/base/env/registration.py:116-136)
# ----------------------------------------------------------------------------------------
def _register_env():
    from srl.base.env import registration

    import _golden_env

    registration.register("TinyImageEnvGolden", entry_point="_golden_env:TinyImageEnv", check_duplicate=False)
    return _golden_env.TinyImageEnv


# ----------------------------------------------------------------------------------------
# srl.rl.functions
# ----------------------------------------------------------------------------------------
def gen_functions():
    from srl.rl import functions as F

    rng = np.random.default_rng(0)
    x32 = (rng.standard_normal(257) * 5).astype(np.float32)
    x64 = rng.standard_normal(257) * 5
    np.savez_compressed(
        os.path.join(OUT, "functions.npz"),
        x32=x32,
        x64=x64,
        rescaling32=F.rescaling(x32),
        inverse_rescaling32=F.inverse_rescaling(x32),
        rescaling64=F.rescaling(x64),
        inverse_rescaling64=F.inverse_rescaling(x64),
        epsilon_list_8=np.array(F.create_epsilon_list(8, epsilon=0.4, alpha=7.0)),
        epsilon_list_1=np.array(F.create_epsilon_list(1, epsilon=0.4, alpha=7.0)),
        beta_list_32=np.array(F.create_beta_list(32)),
        discount_list_32=np.array(F.create_discount_list(32)),
    )
    print("functions: ok")


# ----------------------------------------------------------------------------------------
# rainbow calc_target_q + Trainer.train arithmetic
# ----------------------------------------------------------------------------------------
def _make_rainbow(multisteps, double_dqn, rescale, hw=8, na=4, invalid=False, retrace_h=1.0, discount=0.99):
    import srl
    from srl.algorithms import rainbow

    _register_env()
    env_config = srl.EnvConfig("TinyImageEnvGolden", kwargs=dict(hw=hw, actions=na, invalid=invalid))
    rl_config = rainbow.Config(
        multisteps=multisteps,
        enable_double_dqn=double_dqn,
        enable_rescale=rescale,
        retrace_h=retrace_h,
        discount=discount,
        batch_size=16,
        lr=0.001,
        target_model_update_interval=5,
    )
    rl_config.window_length = 4
    rl_config.memory.warmup_size = 16
    rl_config.memory.capacity = 1000
    rl_config.memory.compress = False
    rl_config.memory.set_proportional(alpha=0.5, beta_initial=0.4, beta_steps=1000)
    rl_config.hidden_block.set_dueling_network((32,))
    rl_config.set_torch()
    return env_config, rl_config


def _random_batches(rng, B, n, obs_shape, A, with_invalid, end_prob=0.25):
    """Items in the reference's nested-list layout (rainbow.py:377-400): n+1 rows of
    [state, onehot action, reward, terminated, next_invalid_actions]."""
    batches = []
    for _ in range(B):
        rows = []
        ended = False
        for k in range(n + 1):
            state = rng.random(obs_shape, dtype=np.float32)
            if k == 0:
                rows.append([state, None, None, None, None])
                continue
            a = int(rng.integers(0, A))
            onehot = [1.0 if i == a else 0.0 for i in range(A)]
            if ended:
                r, d = 0, 1
            else:
                r = int(rng.integers(-1, 2))
                d = int(rng.random() < end_prob)
                ended = ended or d == 1
            inv = [int(x) for x in rng.choice(A, size=int(rng.integers(0, A - 1)), replace=False)] if with_invalid else []
            rows.append([state, onehot, r, d, inv])
        batches.append(rows)
    return batches


def gen_target_q():
    import torch

    for name, kw in [
        ("n3_double", dict(multisteps=3, double_dqn=True, rescale=False)),
        ("n3_single", dict(multisteps=3, double_dqn=False, rescale=False)),
        ("n3_double_rescale_inv", dict(multisteps=3, double_dqn=True, rescale=True, invalid=True)),
        ("n5_double_h09", dict(multisteps=5, double_dqn=True, rescale=False, retrace_h=0.9, discount=0.997)),
        ("n2_single_inv", dict(multisteps=2, double_dqn=False, rescale=False, invalid=True)),
    ]:
        import srl

        env_config, rl_config = _make_rainbow(**kw)
        env = env_config.make()
        rl_config.setup(env)
        torch.manual_seed(0)
        parameter = rl_config.make_parameter()
        # make target differ from online
        with torch.no_grad():
            for p in parameter.q_target.parameters():
                p.add_(0.05 * torch.randn_like(p))
        rng = np.random.default_rng(11)
        n, A, B = rl_config.multisteps, 4, 24
        obs_shape = tuple(rl_config.observation_space.shape)
        batches = _random_batches(rng, B, n, obs_shape, A, kw.get("invalid", False))

        rec = {}
        _pq, _ptq = parameter.pred_q, parameter.pred_target_q

        def pq(state, _pq=_pq):
            out = _pq(state)
            rec["q_online"] = out.copy()
            return out

        def ptq(state, _ptq=_ptq):
            out = _ptq(state)
            rec["q_target"] = out.copy()
            return out

        parameter.pred_q, parameter.pred_target_q = pq, ptq
        target_q, state, action = parameter.calc_target_q(batches)

        n_on = n if rl_config.enable_double_dqn else n - 1
        actions = np.array([[int(np.argmax(b[1])) for b in steps[1:]] for steps in batches], np.int32)
        reward = np.array([[b[2] for b in steps[1:]] for steps in batches], np.float32)
        done = np.array([[b[3] for b in steps[1:]] for steps in batches], np.float32)
        invalid = np.zeros((B, n, A), bool)
        for i, steps in enumerate(batches):
            for k, b in enumerate(steps[1:]):
                for e in b[4]:
                    invalid[i, k, e] = True
        q_online = rec.get("q_online")
        q_online = q_online.reshape(B, n_on, A) if q_online is not None else np.zeros((B, 0, A), np.float32)
        np.savez_compressed(
            os.path.join(OUT, f"target_q_{name}.npz"),
            multisteps=np.int64(n),
            double_dqn=np.int64(rl_config.enable_double_dqn),
            rescale=np.int64(rl_config.enable_rescale),
            retrace_h=np.float64(rl_config.retrace_h),
            discount=np.float64(rl_config.discount),
            q_online=q_online,
            q_target=rec["q_target"].reshape(B, n, A),
            actions=actions,
            reward=reward,
            done=done,
            invalid=invalid,
            target_q=np.asarray(target_q),
            state0=np.asarray(state),
            action0=np.asarray(action),
        )
        print(f"target_q_{name}: target range [{float(np.min(target_q)):.4f}, {float(np.max(target_q)):.4f}]")


def gen_train_step():
    """One Trainer.train() of the reference (rainbow/model_torch.py:85-122) with the host arithmetic
    around the network recorded: q rows, one-hot, IS weights, target -> loss, d loss/d q, priorities;
    plus the network itself (state_dict before/after, inputs) for the Q-network + Adam parity test."""
    import torch

    env_config, rl_config = _make_rainbow(multisteps=3, double_dqn=True, rescale=False)
    env = env_config.make()
    rl_config.setup(env)
    torch.manual_seed(3)
    random.seed(3)
    parameter = rl_config.make_parameter()
    memory = rl_config.make_memory()
    trainer = rl_config.make_trainer(parameter, memory)
    from srl.base.context import RunContext

    trainer.setup(RunContext())
    rng = np.random.default_rng(5)
    obs_shape = tuple(rl_config.observation_space.shape)
    for b in _random_batches(rng, 64, 3, obs_shape, 4, False):
        memory.add(b, None)
    # give the tree non-trivial priorities so that the IS weights are not all 1
    mem = memory.memory
    mem.update([i + mem.capacity - 1 for i in range(64)], rng.random(64).astype(np.float32))

    sd_before = {k: v.detach().clone().numpy() for k, v in parameter.q_online.state_dict().items()}
    sd_target = {k: v.detach().clone().numpy() for k, v in parameter.q_target.state_dict().items()}
    rec = {}
    _sample = memory.sample

    def sample(*a, **k):
        out = _sample(*a, **k)
        rec["batches"], rec["weights"], rec["update_args"] = out
        return out

    memory.sample = sample
    _update = memory.update

    def update(update_args, priorities, step):
        rec["priorities"] = np.asarray(priorities).copy()
        rec["step"] = step
        return _update(update_args, priorities, step)

    memory.update = update
    _calc = parameter.calc_target_q

    def calc(batches):
        out = _calc(batches)
        rec["target_q"], rec["states"], rec["onehot"] = [np.asarray(o).copy() for o in out]
        return out

    parameter.calc_target_q = calc
    # capture q and its gradient
    orig_forward = parameter.q_online.forward
    holder = {}

    def fwd(x):
        y = orig_forward(x)
        if y.requires_grad:
            y.retain_grad()
            holder["q"] = y
        return y

    parameter.q_online.forward = fwd
    trainer.train_count = 1  # not a sync step (train_count % interval != 0)
    trainer.train()
    parameter.q_online.forward = orig_forward
    q = holder["q"]
    sd_after = {k: v.detach().clone().numpy() for k, v in parameter.q_online.state_dict().items()}

    batches = rec["batches"]
    n, A = 3, 4
    obs = np.array([[b[0] for b in steps] for steps in batches], np.float32)  # (B, n+1, h, w, 4)
    actions = np.array([[int(np.argmax(b[1])) for b in steps[1:]] for steps in batches], np.int32)
    reward = np.array([[b[2] for b in steps[1:]] for steps in batches], np.float32)
    done = np.array([[b[3] for b in steps[1:]] for steps in batches], np.float32)
    save = dict(
        obs=obs,
        actions=actions,
        reward=reward,
        done=done,
        weights=np.asarray(rec["weights"]),
        target_q=rec["target_q"],
        q_all=q.detach().numpy(),
        grad_q=q.grad.detach().numpy(),
        loss=np.float32(trainer.info["loss"]),
        priorities=rec["priorities"],
        lr=np.float64(rl_config.lr),
        discount=np.float64(rl_config.discount),
        hw=np.int64(8),
        n_actions=np.int64(A),
        hidden=np.int64(32),
    )
    for k, v in sd_before.items():
        save["before." + k] = v
    for k, v in sd_target.items():
        save["target." + k] = v
    for k, v in sd_after.items():
        save["after." + k] = v
    np.savez_compressed(os.path.join(OUT, "train_step_rainbow.npz"), **save)
    print(f"train_step_rainbow: loss={float(trainer.info['loss']):.6f}")


def gen_dqn_target():
    """dqn.py:144-176 (int `undone` -> float64 expression) and rainbow_nomultisteps.py:10-43 (float32)."""
    import torch

    import srl
    from srl.algorithms import dqn, rainbow
    from srl.algorithms.rainbow import rainbow_nomultisteps

    _register_env()
    for name, double_dqn, rescale in [("double", True, False), ("single_rescale", False, True), ("double_rescale", True, True)]:
        env_config = srl.EnvConfig("TinyImageEnvGolden", kwargs=dict(hw=8, actions=4))
        env = env_config.make()
        rng = np.random.default_rng(21)
        B, A = 20, 4
        # --- DQN
        cfg = dqn.Config(enable_double_dqn=double_dqn, enable_rescale=rescale, discount=0.99)
        cfg.window_length = 4
        cfg.hidden_block.set((16,))
        cfg.set_torch()
        cfg.setup(env)
        torch.manual_seed(1)
        parameter = cfg.make_parameter()
        with torch.no_grad():
            for p in parameter.q_target.parameters():
                p.add_(0.05 * torch.randn_like(p))
        obs_shape = tuple(cfg.observation_space.shape)
        n_state = rng.random((B,) + obs_shape, dtype=np.float32)
        reward = rng.integers(-1, 2, B).astype(np.float32)
        undone = rng.integers(0, 2, B)
        inv_lists = [[int(x) for x in rng.choice(A, size=int(rng.integers(0, 3)), replace=False)] for _ in range(B)]
        q_on = parameter.pred_q(n_state)
        q_tg = parameter.pred_target_q(n_state)
        target = parameter.calc_target_q(B, n_state, reward, undone, inv_lists)
        invalid = np.zeros((B, A), bool)
        for i, l in enumerate(inv_lists):
            invalid[i, l] = True
        # --- Rainbow 1-step
        rcfg = rainbow.Config(multisteps=1, enable_double_dqn=double_dqn, enable_rescale=rescale, discount=0.99)
        rcfg.window_length = 4
        rcfg.hidden_block.set_dueling_network((16,))
        rcfg.set_torch()
        rcfg.setup(env)
        torch.manual_seed(2)
        rparam = rcfg.make_parameter()
        with torch.no_grad():
            for p in rparam.q_target.parameters():
                p.add_(0.05 * torch.randn_like(p))
        rq_on = rparam.pred_q(n_state)
        rq_tg = rparam.pred_target_q(n_state)
        rbatches = [
            [n_state[i], n_state[i], [0.0] * A, float(reward[i]), int(undone[i]), inv_lists[i]] for i in range(B)
        ]
        rtarget, _, _ = rainbow_nomultisteps.calc_target_q(rparam, rbatches, np_dtype=np.float32)
        np.savez_compressed(
            os.path.join(OUT, f"dqn_target_{name}.npz"),
            double_dqn=np.int64(double_dqn),
            rescale=np.int64(rescale),
            discount=np.float64(0.99),
            reward=reward,
            undone=undone,
            invalid=invalid,
            dqn_q_online=q_on,
            dqn_q_target=q_tg,
            dqn_target=np.asarray(target),
            rb_q_online=rq_on,
            rb_q_target=rq_tg,
            rb_target=np.asarray(rtarget),
        )
        print(f"dqn_target_{name}: ok")


def gen_rollout_items():
    """The items the reference's Rainbow worker really emits (stacking, n-step assembly, terminal
    padding, reward clip) for a recorded single-env trajectory: srl.Runner.rollout on TinyImageEnv,
    items read back from the (uncompressed) memory in insertion order."""
    import srl

    TinyImageEnv = _register_env()
    for name, truncate in [("terminated", False), ("truncated", True)]:
        env_config, rl_config = _make_rainbow(multisteps=3, double_dqn=True, rescale=False)
        env_config.kwargs = dict(hw=8, actions=4, ep_len=6, truncate=truncate, seed=9)
        rl_config.enable_reward_clip = True
        rl_config.memory.capacity = 10_000
        runner = srl.Runner(env_config, rl_config)
        runner.set_seed(4)
        runner.rollout(max_steps=40)
        env = runner.env.unwrapped if hasattr(runner.env, "unwrapped") else runner.env.env
        log = env.log
        mem = runner.memory.memory
        items = [mem.tree.data[i] for i in range(mem.size)]
        frames = np.array([l[0].reshape(-1) for l in log], np.uint8)
        actions = np.array([l[1] for l in log], np.int32)
        rewards = np.array([l[2] for l in log], np.float32)
        term = np.array([l[3] for l in log], np.uint8)
        trunc = np.array([l[4] for l in log], np.uint8)
        it_obs = np.array([[np.transpose(np.asarray(r[0], np.float32), (2, 0, 1)).reshape(4, -1) for r in it] for it in items], np.float32)
        it_act = np.array([[int(np.argmax(r[1])) for r in it[1:]] for it in items], np.int32)
        it_rew = np.array([[r[2] for r in it[1:]] for it in items], np.float32)
        it_term = np.array([[r[3] for r in it[1:]] for it in items], np.float32)
        np.savez_compressed(
            os.path.join(OUT, f"rollout_items_{name}.npz"),
            frames=frames,
            actions=actions,
            rewards=rewards,
            terminated=term,
            truncated=trunc,
            item_obs=it_obs,
            item_actions=it_act,
            item_rewards=it_rew,
            item_terminated=it_term,
        )
        print(f"rollout_items_{name}: {len(log)} env records, {len(items)} items")


ALGO_GENERATORS = dict(
    functions=gen_functions,
    target_q=gen_target_q,
    train_step=gen_train_step,
    dqn_target=gen_dqn_target,
    rollout_items=gen_rollout_items,
)
