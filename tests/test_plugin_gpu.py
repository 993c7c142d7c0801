"""GPU tests of the DQN / Rainbow plugin trainers (reference-compatible single-env path whose TD / loss /
priority arithmetic runs in libsrlx)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")

import simple_distributed_rl_amd as srl  # noqa: E402


def test_rainbow_trainer_step_matches_reference_golden():
    """One full learner update vs the reference's recorded Trainer.train() (train_step_rainbow.npz): same
    initial online/target weights and the same sampled batch -> target, loss, d loss/d q, priorities and the
    Adam-updated weights agree to 1e-5 (north_star tolerance)."""
    from simple_distributed_rl_amd.algorithms import rainbow
    from simple_distributed_rl_amd.base.context import RunContext
    from test_plugin_surface import TinyImg  # noqa: F401
    from simple_distributed_rl_amd.base.env import registration

    registration.register("TinyImg", "test_plugin_surface:TinyImg", check_duplicate=False)
    z = np.load(os.path.join(GOLDEN, "train_step_rainbow.npz"))
    rl = rainbow.Config(multisteps=3, enable_double_dqn=True, batch_size=16, lr=float(z["lr"]), target_model_update_interval=5, discount=float(z["discount"]))
    rl.window_length = 4
    rl.memory.capacity, rl.memory.warmup_size, rl.memory.compress = 1000, 16, False
    rl.hidden_block.set_dueling_network((32,))
    runner = srl.Runner(srl.EnvConfig("TinyImg"), rl)
    runner.set_device("cuda:0")
    param, trainer = runner.parameter, runner.trainer
    ctx = RunContext(runner.env_config, rl)
    ctx.setup_device()
    trainer.setup(ctx)
    param.q_online.load_state_dict({k[7:]: torch.tensor(z[k]) for k in z.files if k.startswith("before.")})
    param.q_target.load_state_dict({k[7:]: torch.tensor(z[k]) for k in z.files if k.startswith("target.")})
    obs, actions, reward, done = z["obs"], z["actions"], z["reward"], z["done"]
    B, A = obs.shape[0], 4
    batches = []
    for b in range(B):
        rows = [[obs[b, 0], None, None, None, None]]
        for k in range(3):
            onehot = [1.0 if a == actions[b, k] else 0.0 for a in range(A)]
            rows.append([obs[b, k + 1], onehot, float(reward[b, k]), int(done[b, k]), []])
        batches.append(rows)
    target, loss, grad, pri, q = trainer.calc(batches, z["weights"])
    np.testing.assert_allclose(q.detach().cpu().numpy(), z["q_all"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(target.cpu().numpy(), z["target_q"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(loss.item()), float(z["loss"]), rtol=1e-5)
    # gradient seed = -w * clamp(target * w - q * w) / B and priority = |target - q|: differences of O(1) numbers that are each good to rel 1e-5 -- the residue is held
    # to 1e-5 of THEIR size (the cancellation bound), entries that are not residues to rel 1e-5
    scale = float(np.abs(z["target_q"]).max())
    np.testing.assert_allclose(grad.cpu().numpy(), z["grad_q"], rtol=1e-5, atol=1e-5 * scale / B)
    np.testing.assert_allclose(pri.cpu().numpy(), z["priorities"], rtol=1e-5, atol=1e-5 * scale)
    trainer.optimizer.zero_grad()
    q.backward(grad)
    trainer.optimizer.step()
    sd = param.q_online.state_dict()
    for k in z.files:
        if k.startswith("after."):
            got, want, before = sd[k[6:]].cpu().numpy(), z[k], z["before." + k[6:]]
            # Adam's first step moves every weight by ~lr; compare the UPDATE, not just the weight
            np.testing.assert_allclose(got - before, want - before, rtol=1e-2, atol=5e-6)
            # Adam's first step is lr * g/(|g|+eps): for |g| ~ eps the direction amplifies last-ulp gradient
            # differences between MIOpen and the CPU convolution, so allow 0.5 % of one lr-sized update (lr = 1e-3)
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize("algo", ["dqn", "rainbow", "rainbow_1step"])
def test_plugin_trains_on_grid(algo):
    """algorithm smoke ("quick" tier, common_quick_case.py): sequential train on Grid with PER in HBM,
    loss finite, target sync counted, parameter save/load round trip."""
    from simple_distributed_rl_amd.algorithms import dqn, rainbow

    if algo == "dqn":
        rl = dqn.Config(batch_size=16, target_model_update_interval=50)
        rl.hidden_block.set((32, 32))
    else:
        rl = rainbow.Config(batch_size=16, target_model_update_interval=50, multisteps=1 if algo == "rainbow_1step" else 3)
        rl.hidden_block.set_dueling_network((32, 32))
    rl.memory.capacity, rl.memory.warmup_size, rl.memory.compress = 2000, 100, False
    rl.memory.set_proportional(alpha=0.5, beta_steps=1000)
    runner = srl.Runner("Grid", rl)
    runner.set_seed(3)
    st = runner.train(max_steps=400)
    assert st.train_count == 400 - 100 + 1 or st.train_count > 250
    info = runner.trainer.info
    assert np.isfinite(info["loss"]) and info["sync"] >= 1
    assert runner.memory.length() > 300
    sd = runner.parameter.backup(serialized=True)
    assert all(v.device.type == "cpu" for v in sd.values())
    r = runner.evaluate(max_episodes=3)
    assert len(r) == 3


def test_trainer_refuses_cpu():
    from simple_distributed_rl_amd.algorithms import dqn

    rl = dqn.Config(batch_size=4)
    rl.memory.warmup_size = 4
    runner = srl.Runner("Grid", rl)
    runner.set_device("CPU")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        runner.train(max_steps=10)


def test_dqn_cartpole_config_runs_and_learns():
    """BASELINE.json configs[1]: DQN on CartPole-v1, uniform replay 1e5, batch 32, one GPU, through srl.Runner.
    Plumbing (environment -> WorkerRun -> uniform ReplayBuffer -> device trainer -> evaluate) plus a learning signal:
    a random policy balances the pole for ~22 steps."""
    from simple_distributed_rl_amd.algorithms import dqn

    from simple_distributed_rl_amd.utils.common import set_seed

    set_seed(3, enable_gpu=True)
    rl = dqn.Config(batch_size=32, lr=0.001, target_model_update_interval=200, discount=0.99)
    rl.memory.set_replay_buffer()
    rl.memory.capacity, rl.memory.warmup_size = 100_000, 500
    rl.epsilon_scheduler.set_linear(1.0, 0.05, 3000)
    rl.hidden_block.set((64, 64))
    runner = srl.Runner("CartPole-v1", rl)
    runner.set_device("cuda:0")
    runner.train(max_train_count=6000, enable_progress=False)
    assert runner.trainer.train_count >= 6000 and runner.memory.length() >= 6000
    rewards = runner.evaluate(max_episodes=10, enable_progress=False)
    assert len(rewards) == 10 and np.mean(rewards) > 60, rewards


def test_ppo_plugin_loss_matches_oracle():
    """"PPO:torch" (discrete actions; the reference's PPO needs TensorFlow, so parity is against the oracle's restatement
    of ppo.py:102-169, 389-404): the trainer's fused loss equals `ppo_loss`, its gradient seeds equal torch autograd of the
    same formula, the worker's one-launch GAE equals the reverse scan."""
    sys.path.insert(0, ROOT)
    from oracle import hot_path_oracle as O
    from simple_distributed_rl_amd.algorithms import ppo
    from simple_distributed_rl_amd.utils.common import set_seed

    set_seed(7, enable_gpu=True)
    rl = ppo.Config(batch_size=64, lr=0.002, train_num=20, discount=0.98, gae_discount=0.95, entropy_weight=0.01, train_every_epoch=True)
    rl.memory.warmup_size = 1000
    rl.lr_scheduler.set_constant()
    runner = srl.Runner("CartPole-v1", rl)
    runner.set_device("cuda:0")
    runner.train(max_train_count=20, enable_progress=False)  # one generation: sets everything up
    trainer = runner.trainer
    dev = trainer.device

    # --- loss + seeds vs the oracle / autograd of the restated formula
    rng = np.random.default_rng(0)
    B = 96
    lp = torch.tensor(-np.abs(rng.standard_normal((B, 1))).astype(np.float32) - 0.05, device=dev, requires_grad=True)
    olp = torch.tensor(lp.detach().cpu().numpy() + 0.3 * rng.standard_normal((B, 1)).astype(np.float32), device=dev)
    adv = torch.tensor(rng.standard_normal(B).astype(np.float32), device=dev)
    v = torch.tensor(rng.standard_normal(B).astype(np.float32), device=dev, requires_grad=True)
    vt = torch.tensor(rng.standard_normal(B).astype(np.float32), device=dev)
    ov = torch.tensor(v.detach().cpu().numpy() + 0.3 * rng.standard_normal(B).astype(np.float32), device=dev)
    losses, g_lp, g_v = trainer.losses_and_seeds(lp, olp, adv, v, vt, ov)
    c = rl
    want = O.ppo_loss(lp.detach().cpu().numpy(), olp.cpu().numpy(), adv.cpu().numpy(), v.detach().cpu().numpy(), vt.cpu().numpy(), ov.cpu().numpy(), True, True,
                      c.policy_clip_range, True, c.value_clip_range, c.value_loss_weight, c.entropy_weight)
    np.testing.assert_allclose(losses.cpu().numpy(), np.asarray(want, np.float32), rtol=1e-5, atol=1e-7)
    a2 = adv.view(-1, 1) - v.detach().view(-1, 1)
    ratio = torch.exp(lp - olp)
    pol = -torch.minimum(ratio * a2, torch.clamp(ratio, 1 - c.policy_clip_range, 1 + c.policy_clip_range) * a2).mean()
    vc = torch.maximum(torch.minimum(v, ov + c.value_clip_range), ov - c.value_clip_range)
    val = c.value_loss_weight * torch.maximum((v - vt) ** 2, (vc - vt) ** 2).mean()
    ent = c.entropy_weight * -(-(torch.exp(lp) * lp).sum(-1)).mean()
    (pol + val + ent).backward()
    np.testing.assert_allclose(g_lp.cpu().numpy(), lp.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(g_v.cpu().numpy(), v.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)

    # --- the items of the last generation carry the GAE of ppo.py:389-404 (checked on one stored episode's arithmetic)
    T = 17
    r = rng.standard_normal(T).astype(np.float32)
    vals = rng.standard_normal(T).astype(np.float32)
    gae, ref = 0.0, np.zeros(T, np.float32)
    for i in reversed(range(T)):
        delta = r[i] - vals[i] if i == T - 1 else r[i] + np.float32(c.discount) * vals[i + 1] - vals[i]
        gae = delta + np.float32(c.discount * c.gae_discount) * gae
        ref[i] = gae
    from simple_distributed_rl_amd import _native as N

    done = torch.zeros((T, 1), dtype=torch.uint8, device=dev)
    done[T - 1] = 1
    out = torch.empty((T, 1), dtype=torch.float32, device=dev)
    r_t, v_t = torch.tensor(r, device=dev).view(T, 1), torch.tensor(vals, device=dev).view(T, 1)
    N.check(N.lib().srlx_gae_scan(1, T, N.tptr(r_t), N.tptr(v_t), N.tptr(done), None, float(c.discount), float(c.gae_discount), N.tptr(out), N.torch_stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.view(-1).cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.slow
def test_ppo_plugin_cartpole_learns():
    """The learning signal of the plugin above (a learning curve, 20 s: `slow` -- the loss, its gradient seeds and the GAE are pinned by the test above, the PPO engine's
    learning by tests/test_ppo_gpu.py): CartPole-v1 improves well beyond random play."""
    from simple_distributed_rl_amd.algorithms import ppo
    from simple_distributed_rl_amd.utils.common import set_seed

    set_seed(7, enable_gpu=True)
    rl = ppo.Config(batch_size=64, lr=0.002, train_num=20, discount=0.98, gae_discount=0.95, entropy_weight=0.01, train_every_epoch=True)
    rl.memory.warmup_size = 1000
    rl.lr_scheduler.set_constant()
    runner = srl.Runner("CartPole-v1", rl)
    runner.set_device("cuda:0")
    runner.train(max_train_count=1220, enable_progress=False)
    rewards = runner.evaluate(max_episodes=10, enable_progress=False)
    assert np.mean(rewards) > 60, rewards


def test_train_mp_dqn_cartpole_on_gpu():
    """Runner.train_mp (play_mp.py:471-642) with the device trainer: two actor processes and the learner share the GPU, items
    cross the serialising queue, parameters flow back through the board, and the result plays CartPole far better than random."""
    from simple_distributed_rl_amd.algorithms import dqn
    from simple_distributed_rl_amd.utils.common import set_seed

    set_seed(3, enable_gpu=True)
    rl = dqn.Config(batch_size=32, lr=0.001, target_model_update_interval=200, discount=0.99)
    rl.memory.set_replay_buffer()
    rl.memory.capacity, rl.memory.warmup_size = 100_000, 500
    rl.epsilon_scheduler.set_linear(1.0, 0.05, 3000)
    rl.hidden_block.set((64, 64))
    runner = srl.Runner("CartPole-v1", rl)
    runner.set_device("cuda:0")
    st = runner.train_mp(actor_num=2, max_train_count=4000, timeout=240, trainer_parameter_send_interval=0.5, actor_parameter_sync_interval=0.5,
                         enable_progress=False)
    assert st.train_count >= 4000 and st.end_reason == "max_train_count over." and st.trainer_recv_q > 4000
    rewards = runner.evaluate(max_episodes=10, enable_progress=False)
    assert np.mean(rewards) > 50, rewards


@pytest.mark.slow
def test_ppo_plugin_continuous_pendulum():
    """"PPO:torch" with a continuous action space (BASELINE.json configs[4] shapes through the plugin surface): the space
    negotiation hands the algorithm an NpArraySpace with the environment's bounds, the Normal-policy loss of the trainer
    (`srlx_ppo_loss_normal`) equals the oracle's restatement, actions reach the environment inside its bounds, and
    Pendulum-v1 improves far beyond the untrained policy (about -1450 per episode)."""
    sys.path.insert(0, ROOT)
    from oracle import hot_path_oracle as O
    from simple_distributed_rl_amd.algorithms import ppo
    from simple_distributed_rl_amd.base.spaces.np_array import NpArraySpace
    from simple_distributed_rl_amd.utils.common import set_seed

    set_seed(1, enable_gpu=True)
    rl = ppo.Config(batch_size=64, lr=0.001, train_num=20, discount=0.95, gae_discount=0.9, entropy_weight=0.001, baseline_type="advantage", train_every_epoch=True)
    rl.memory.warmup_size = 1000
    rl.lr_scheduler.set_constant()
    runner = srl.Runner("Pendulum-v1", rl)
    runner.set_device("cuda:0")
    runner.train(max_train_count=1500, enable_progress=False)
    space = runner.rl_config.action_space if hasattr(runner, "rl_config") else rl.action_space
    assert isinstance(space, NpArraySpace) and space.size == 1 and float(space.low[0]) == -2.0 and float(space.high[0]) == 2.0
    a = space.sanitize(space.rescale_from(np.array([3.0], np.float32)))
    assert a.shape == (1,) and float(a[0]) == 2.0  # a policy output beyond [-1, 1] is clipped onto the torque bound

    trainer = runner.trainer
    dev = trainer.device
    rng = np.random.default_rng(2)
    B, D = 48, 1
    T = lambda x: torch.tensor(np.asarray(x, np.float32), device=dev)  # noqa: E731
    loc, ls, act = rng.standard_normal((B, D)), 0.3 * rng.standard_normal((B, D)) - 0.5, rng.standard_normal((B, D))
    olp, adv, v, vt = -np.abs(rng.standard_normal((B, D))) - 0.2, rng.standard_normal(B), rng.standard_normal(B), rng.standard_normal(B)
    ov = v + 0.3 * rng.standard_normal(B)
    losses, g_loc, g_ls, g_v = trainer.losses_and_seeds_normal(T(loc), T(ls), T(act), T(olp), T(adv), T(v), T(vt), T(ov))
    lo, hi = runner.parameter.model.log_scale_range
    new_lp = O.normal_logprob(np.float32(act), np.float32(loc), np.clip(np.float32(ls), np.float32(lo), np.float32(hi)))
    want = O.ppo_loss(new_lp, np.float32(olp), np.float32(adv), np.float32(v), np.float32(vt), np.float32(ov), True, True, rl.policy_clip_range, True,
                      rl.value_clip_range, rl.value_loss_weight, rl.entropy_weight)
    np.testing.assert_allclose(losses.cpu().numpy(), np.asarray(want, np.float32), rtol=1e-5, atol=1e-7)
    assert np.isfinite(g_loc.cpu().numpy()).all() and np.isfinite(g_ls.cpu().numpy()).all() and np.isfinite(g_v.cpu().numpy()).all()

    rewards = runner.evaluate(max_episodes=5, enable_progress=False)
    assert np.mean(rewards) > -900, rewards


def test_ppo_train_count_per_call_follows_the_reference():
    """srl/algorithms/ppo/ppo.py:203-205: `f = f or self._train()` short-circuits, so one `train()` call on a warm buffer takes
    exactly ONE gradient step before the buffer is cleared, whatever `train_num` says; `train_every_epoch=True` (this build's
    documented extension) takes `train_num`.  And a GAE worker on a CPU context refuses cleanly (no host pointer reaches a kernel)."""
    from simple_distributed_rl_amd.algorithms import ppo

    for every, want in ((False, 1), (True, 7)):
        rl = ppo.Config(batch_size=16, train_num=7, train_every_epoch=every)
        rl.memory.warmup_size = 64
        runner = srl.Runner("CartPole-v1", rl)
        runner.set_device("cuda:0")
        runner.rollout(max_steps=600, enable_progress=False)  # items arrive at episode ends (ppo.py:381-404)
        trainer = runner.trainer
        trainer.setup(runner.context)
        assert runner.memory.length() >= 64
        before = trainer.train_count
        trainer.train()
        assert trainer.train_count - before == want
        assert runner.memory.length() == 0  # cleared after a trained call
    cpu = srl.Runner("CartPole-v1", ppo.Config())
    cpu.set_device("CPU")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cpu.rollout(max_steps=50, enable_progress=False)


@pytest.mark.parametrize("name", ["n3", "n1"])
def test_distributed_actor_initial_priorities_match_the_reference(name):
    """rainbow.py:389-398 / rainbow_nomultisteps.py:108-119: a distributed actor computes its items' first priorities itself.  Same weights,
    same seed, same environment as the run oracle/gen_golden_actor_priority.py recorded from the reference's worker: same actions, same priorities."""
    import random

    from simple_distributed_rl_amd.algorithms import rainbow
    from simple_distributed_rl_amd.base.context import RunContext
    from simple_distributed_rl_amd.base.env import registration
    from simple_distributed_rl_amd.utils import common

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_plugin_surface  # noqa: F401  (defines TinyImg)

    z = np.load(os.path.join(ROOT, "tests", "golden", f"actor_priority_{name}.npz"))
    registration.register("TinyImg", "test_plugin_surface:TinyImg", check_duplicate=False)
    rl = rainbow.Config(multisteps=int(z["multisteps"]), enable_double_dqn=True, enable_rescale=False, retrace_h=1.0, discount=0.99, batch_size=16, lr=0.001,
                        target_model_update_interval=5, enable_reward_clip=True, epsilon=float(z["epsilon"]))
    rl.window_length = 4
    rl.memory.capacity, rl.memory.warmup_size, rl.memory.compress = 1000, 16, False
    rl.memory.set_proportional(alpha=0.5, beta_initial=0.4, beta_steps=1000)
    rl.hidden_block.set_dueling_network((32,))
    runner = srl.Runner(srl.EnvConfig("TinyImg", kwargs=dict(hw=8, actions=4, ep_len=6, seed=9)), rl)
    runner.set_device("cuda:0")
    env, parameter, memory = runner.make_env(), runner.make_parameter(), runner.make_memory()
    sd = {k[2:]: torch.as_tensor(z[k]) for k in z.files if k.startswith("w:")}
    parameter.q_online.load_state_dict(sd)
    parameter.q_target.load_state_dict(sd)
    worker = runner.make_worker(parameter, memory)
    qmax = [0.0]
    for fn_name in ("pred_q", "pred_target_q"):  # the largest |Q| the worker ever sees: the scale of what its priorities are residues of
        def wrapped(state, _f=getattr(parameter, fn_name)):
            out = _f(state)
            qmax[0] = max(qmax[0], float(np.abs(np.asarray(out)).max()))
            return out
        setattr(parameter, fn_name, wrapped)
    got = []
    worker.worker.memory = type("Rec", (), {"add": staticmethod(lambda batch, priority=None, **kw: got.append(priority)), "config": memory.config})()
    ctx = RunContext(runner.env_config, rl)
    ctx.distributed, ctx.training, ctx.actor_num, ctx.actor_id = True, True, 1, 0
    ctx.device = "cuda:0"
    ctx.setup_device()
    common.set_seed(int(z["seed"]))
    env.setup(ctx)
    worker.setup(ctx)
    for ep in range(5):
        env.reset()
        worker.reset(0)
        while not env.done:
            env.step(worker.policy())
            worker.on_step()
    log = env.unwrapped.log
    np.testing.assert_array_equal(np.array([l[1] for l in log], np.int32), z["actions"])  # the same trajectory (epsilon draws, argmax, padding draws)
    assert len(got) == len(z["priorities"]) and all(p is not None for p in got)
    # a priority |n-step target - Q(s_0, a_0)| is a residue of up to 2 (n + 1) Q-values (rewards are exact small integers): with every Q-value within 1e-5 of the
    # reference's (`north_star`; here torch's GPU kernels against the reference's CPU run) it is within 1e-5 x 2 (n + 1) x max |Q| of the recorded one
    n = int(z["multisteps"])
    np.testing.assert_allclose(np.array(got), z["priorities"], rtol=1e-5, atol=1e-5 * 2 * (n + 1) * qmax[0])
