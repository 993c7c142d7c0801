"""GPU tests of the vectorised Rainbow engine end to end (SURVEY 8 a9-a16 fused on the device): the hand-written
training pass against the autograd path on the same replay contents, and the overlapped / HIP-graph step loop."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _engine(torch_backward: bool, fused_adam: bool = True, fused_td: bool = True, **kw):
    from simple_distributed_rl_amd.device.rainbow import EngineSchedule, RainbowDeviceConfig, RainbowEngine

    cfg = RainbowDeviceConfig(n_envs=8, batch_size=8, memory_capacity=8 * 64, memory_warmup_size=32, target_model_update_interval=4, lr=1e-4, seed=3,
                              schedule=EngineSchedule(autograd_yardstick=torch_backward, fused_adam=fused_adam, fused_td=fused_td))
    return RainbowEngine(cfg, 0, episode_len=9, **kw)


def test_training_pass_equals_autograd_path():
    """Two engines on the same seed: one differentiates with the libsrlx backward kernels (one forward over s_0..s_n read
    from the uint8 ring), the other with torch autograd on float32 pixels.  Same sampled items, TD targets, loss,
    priorities and parameter gradients at every learner step.  (The gradient of the first dense layer is only materialised
    with its Adam step left in `srlx_adam_step`: test_fused_first_dense_adam_is_the_same_update covers the fused kernel.)"""
    a, b = _engine(False, fused_adam=False), _engine(True)
    assert a.mfma_train and not b.mfma_train and a.optimizer._fused is None
    b.q_online.load_state_dict(a.q_online.state_dict())
    b.q_target.load_state_dict(a.q_target.state_dict())
    steps = 0
    for it in range(14):
        for e in (a, b):
            e.step(learner_updates=1)
        torch.cuda.synchronize()
        if a.train_count == 0:
            continue
        steps += 1
        np.testing.assert_array_equal(a.replay.batch.indices.cpu().numpy(), b.replay.batch.indices.cpu().numpy())
        if steps == 1:  # identical parameters so far: every intermediate must agree to float32 round-off
            np.testing.assert_allclose(a.target.cpu().numpy(), b.target.cpu().numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(a.loss.item(), b.loss.item(), rtol=1e-5)
            np.testing.assert_allclose(a.priorities.cpu().numpy(), b.priorities.cpu().numpy(), rtol=1e-4, atol=1e-6)
            for (name, pa), pb in zip(a.q_online.named_parameters(), b.q_online.parameters()):
                scale = float(pb.grad.abs().max()) + 1e-12
                np.testing.assert_allclose(pa.grad.cpu().numpy(), pb.grad.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale, err_msg=name)
        else:  # Adam amplifies round-off of tiny gradients: later steps stay close, not identical
            np.testing.assert_allclose(a.loss.item(), b.loss.item(), rtol=2e-2)
    assert steps >= 5
    for pa, pb in zip(a.q_online.parameters(), b.q_online.parameters()):
        assert float((pa - pb).detach().abs().max()) < 5 * 1e-4 * steps  # lr per Adam step bounds the drift


def test_fused_first_dense_adam_is_the_same_update():
    """srlx_qnet_fuse_adam_fc1: Adam inside the first dense layer's weight-gradient kernel (the engine's default) against the same
    engine with that tensor left to srlx_adam_step -- same accumulation order, same arithmetic: parameters and both moment
    estimates of EVERY tensor stay bit-equal over the learner steps, and the fused tensor does move."""
    a, c = _engine(False), _engine(False, fused_adam=False)
    k = a.optimizer._fused
    assert k is not None and a.optimizer.params[k] is a.q_online.fc1.weight and c.optimizer._fused is None
    c.q_online.load_state_dict(a.q_online.state_dict())
    c.q_target.load_state_dict(a.q_target.state_dict())
    w0 = a.q_online.fc1.weight.detach().clone()
    for it in range(12):
        for e in (a, c):
            e.step(learner_updates=1)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(a.priorities.cpu().numpy(), c.priorities.cpu().numpy())
        for i, (pa, pc) in enumerate(zip(a.optimizer.params, c.optimizer.params)):
            assert torch.equal(pa, pc), (it, i)
            assert torch.equal(a.optimizer.exp_avg[i], c.optimizer.exp_avg[i]) and torch.equal(a.optimizer.exp_avg_sq[i], c.optimizer.exp_avg_sq[i]), (it, i)
    assert a.train_count >= 5 and float((a.q_online.fc1.weight - w0).abs().max()) > 0
    assert float(a.optimizer.exp_avg[k].abs().max()) > 0


def test_td_inside_the_backward_head_kernel_is_the_same_update():
    """srlx_qnet_backward_td_u8 (TD target / Huber loss / gradient seed / priorities evaluated in the prologue of the backward's head
    kernel: the engine's default) against srlx_nstep_td_huber_priority_packed + srlx_qnet_backward_u8: everything bit-equal."""
    a, c = _engine(False), _engine(False, fused_td=False)
    assert a._fused_td and not c._fused_td
    c.q_online.load_state_dict(a.q_online.state_dict())
    c.q_target.load_state_dict(a.q_target.state_dict())
    for it in range(10):
        for e in (a, c):
            e.step(learner_updates=1)
        torch.cuda.synchronize()
        for name in ("target", "loss", "grad_q0", "priorities"):
            assert torch.equal(getattr(a, name), getattr(c, name)), (it, name)
        for i, (pa, pc) in enumerate(zip(a.optimizer.params, c.optimizer.params)):
            assert torch.equal(pa, pc), (it, i)
    assert a.train_count >= 5 and float(a.grad_q0.abs().max()) > 0 and float(a.priorities.max()) > 0


def test_engine_overlap_and_graphs():
    """Two-stream overlap + HIP graphs with the hand-written training pass: the captured step loop runs, counts the
    environment steps, keeps a finite loss and moves the parameters."""
    eng = _engine(False, overlap=True)
    before = [p.detach().clone() for p in eng.q_online.parameters()]
    for _ in range(8):
        eng.step(learner_updates=1)
    torch.cuda.synchronize()
    eng.capture_graphs()
    for _ in range(20):
        eng.step(learner_updates=1)
    torch.cuda.synchronize()
    info = eng.info()
    assert eng.total_env_steps >= 28 * 8 and info["train_count"] >= 20 and np.isfinite(info["loss"]), info
    assert any(float((p - q).abs().max()) > 0 for p, q in zip(eng.q_online.parameters(), before))
    for p, q in zip(eng.q_actor.parameters(), eng.q_online.parameters()):
        assert torch.equal(p, q)  # the actor's copy is refreshed after every step


def test_device_adam_matches_torch_adam():
    """`srlx_adam_step` (one launch over all parameter tensors, step count from a device scalar) against
    torch.optim.Adam on the same gradients: parameters and both moment estimates after every one of 6 steps.
    Shapes cover whole 2048-element chunks, ragged tails, a tensor smaller than one chunk and a channels_last weight."""
    from simple_distributed_rl_amd.device.qnet import DeviceAdam

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(32, 4, 8, 8), (32,), (64, 32, 4, 4), (512, 3136), (7,), (6, 512), (4097,)]
    ours, ref = [], []
    for i, s in enumerate(shapes):
        t = torch.randn(s, device=dev, generator=g) * 0.1
        if len(s) == 4 and i == 2:
            t = t.contiguous(memory_format=torch.channels_last)
        a = torch.nn.Parameter(t.clone(memory_format=torch.preserve_format))
        a.grad = torch.zeros_like(a)
        ours.append(a)
        ref.append(torch.nn.Parameter(t.clone(memory_format=torch.preserve_format)))
    opt = DeviceAdam(ours, lr=2.5e-4)
    topt = torch.optim.Adam(ref, lr=2.5e-4)
    steps = torch.zeros(1, dtype=torch.int64, device=dev)
    for it in range(6):
        for a, b in zip(ours, ref):
            gr = torch.randn(a.shape, device=dev, generator=g) * (10.0 ** (-it))  # large and tiny gradients
            a.grad.copy_(gr)
            b.grad = gr.clone(memory_format=torch.preserve_format)
        opt.step(steps)
        steps.add_(1)
        topt.step()
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(ours, ref)):
            st = topt.state[b]
            m_ref = st["exp_avg"].cpu().numpy()  # m + w (g - m) cancels: compare on the scale of the tensor
            np.testing.assert_allclose(opt.exp_avg[k].cpu().numpy(), m_ref, rtol=2e-6, atol=2e-7 * float(np.abs(m_ref).max()), err_msg=f"m {k} step {it}")
            np.testing.assert_allclose(opt.exp_avg_sq[k].cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-20, err_msg=f"v {k} step {it}")
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-6, atol=2e-8, err_msg=f"p {k} step {it}")


class _CueVecEnv:
    """E learnable environments for the engine: every frame is a constant brightness that encodes a cue in {0..A-1};
    the action equal to the cue of the frame being looked at earns +1, any other -1; cues are i.i.d., episodes last
    `episode_len` steps.  Same contract as SyntheticAtariVecEnv (a done env takes one reset step whose action is ignored)."""

    def __init__(self, replay, episode_len, n_actions, seed=0):
        self.replay, self.episode_len, self.A = replay, int(episode_len), int(n_actions)
        d = replay.dev
        E, F = replay.E, replay.F
        self.g = torch.Generator(device=d)
        self.g.manual_seed(seed)
        self.next_obs = torch.zeros((E, F), dtype=torch.uint8, device=d)
        self.rewards = torch.zeros(E, dtype=torch.float32, device=d)
        self.terminated = torch.zeros(E, dtype=torch.uint8, device=d)
        self.done = torch.zeros(E, dtype=torch.uint8, device=d)
        self.cue = torch.zeros(E, dtype=torch.int64, device=d)
        self.t = torch.zeros(E, dtype=torch.int64, device=d)
        self.pending = torch.zeros(E, dtype=torch.bool, device=d)
        self.reward_sum, self.reward_n = 0.0, 0

    def _show(self):
        E, F = self.next_obs.shape
        self.cue = torch.randint(0, self.A, (E,), device=self.cue.device, generator=self.g)
        noise = torch.randint(0, 8, (E, F), device=self.cue.device, generator=self.g)
        self.next_obs.copy_((20 + 60 * self.cue).view(E, 1) + noise)

    def reset(self):
        self._show()
        return self.next_obs

    def step(self, actions):
        acting = ~self.pending
        hit = actions.to(torch.int64) == self.cue
        r = torch.where(hit, 1.0, -1.0)
        self.rewards.copy_(torch.where(acting, r, torch.zeros_like(r)))
        self.t = torch.where(acting, self.t + 1, torch.zeros_like(self.t))
        d = acting & (self.t >= self.episode_len)
        self.done.copy_(d.to(torch.uint8))
        self.terminated.copy_(d.to(torch.uint8))
        self.reward_sum += float(self.rewards[acting].sum().item())
        self.reward_n += int(acting.sum().item())
        self.pending = d
        self._show()
        return self.next_obs, self.rewards, self.terminated, self.done


def test_engine_learns_a_cue_task():
    """End to end on the hand-written path (uint8 ring -> matrix-core forward -> fused TD/Huber -> backward kernels ->
    srlx_adam_step -> PER update): the engine learns to answer a brightness cue.  Random play scores -0.5 per step."""
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    A = 4
    cfg = RainbowDeviceConfig(n_envs=64, batch_size=32, memory_capacity=64 * 64, memory_warmup_size=256, obs_hw=(20, 20), hidden_units=64,
                              n_actions=A, seed=6, target_model_update_interval=25, lr=1e-3, epsilon=0.2, discount=0.9)
    eng = RainbowEngine(cfg, 0, episode_len=6, overlap=False)
    env = _CueVecEnv(eng.replay, 6, A, seed=9)
    eng.env = env
    eng.replay.reset_all(env.reset())
    assert eng.mfma_train
    for _ in range(800):
        eng.step(learner_updates=2)
    torch.cuda.synchronize()
    assert eng.train_count > 1400 and np.isfinite(eng.loss.item())
    eng.eps.fill_(0.0)  # greedy evaluation
    env.reward_sum, env.reward_n = 0.0, 0
    for _ in range(40):
        eng.step(learner_updates=0)
    torch.cuda.synchronize()
    score = env.reward_sum / env.reward_n
    assert score > 0.8, f"greedy score {score:.3f} after {eng.train_count} updates (random play: -0.5)"


def test_graph_replays_are_bit_stable_without_host_synchronisation():
    """The captured actor / learner graphs replayed back to back (as bench.py and the Runner drivers do) against one host synchronisation per
    lock-step: parameters, loss, train count and priorities bit-equal after 300 lock-steps with the two-stream overlap on (the PPO engine's
    large torch-captured graphs were NOT stable that way: device/ppo.py:step)."""
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    def run(sync):
        cfg = RainbowDeviceConfig(n_envs=64, batch_size=32, memory_capacity=20_000, memory_warmup_size=500, seed=0)
        eng = RainbowEngine(cfg, 0, 50, overlap=True)
        eng.prefill()
        for _ in range(5):
            eng.step(1)
        torch.cuda.synchronize()
        eng.capture_graphs()
        for _ in range(300):
            eng.step(1)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [p.detach().clone() for p in eng.q_online.parameters()], float(eng.loss), int(eng.train_count_dev), eng.priorities.clone()

    pa, la, ca, ra = run(True)
    pb, lb, cb, rb = run(False)
    assert ca == cb and ca >= 300 and la == lb and torch.equal(ra, rb)
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "graphs"])
def test_actor_side_initial_priorities(graphs):
    """cfg.actor_initial_priority (the reference's distributed worker, rainbow.py:389-398: a new item enters the memory with |n-step target - Q(s_0, a_0)|
    instead of max_priority).  With the weights standing still the cached Q rows ARE what the reference would re-evaluate, so every leaf added for an item
    inside an episode equals (|td| + eps)^alpha with td from the oracle's n-step target on the network's own Q-values of the stored states (online rows in
    both roles: an actor holds no target network); items whose window touches an episode end keep max_priority; positions without an item weigh 0.
    `graphs`: the same after capture_graphs() in the middle of the run (its warm actor step goes through the deferred-add bookkeeping too: every leaf still
    belongs to the item of its slot).  The Q-values fed to the oracle come from the kernels that produced the cached rows (same network, same uint8 frames):
    what is under test here is the TD arithmetic, at `north_star`'s 1e-5; the forward itself is held to the reference in tests/test_qnet_pinned.py."""
    import ctypes

    sys.path.insert(0, os.path.join(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")), "oracle"))
    import hot_path_oracle as H
    from simple_distributed_rl_amd import _native as N
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    E, n, A = 8, 3, 6
    cfg = RainbowDeviceConfig(n_envs=E, batch_size=8, memory_capacity=E * 64, memory_warmup_size=1 << 40, actor_initial_priority=True, epsilon=0.3, seed=5)
    eng = RainbowEngine(cfg, 0, episode_len=17)
    steps = 50
    for k in range(steps):
        if graphs and k == 20:
            eng.capture_graphs(learner=False)  # (one more lock-step: the warm actor step)
        eng.step(learner_updates=0)
    torch.cuda.synchronize()
    if graphs:
        steps += 1
    r = eng.replay
    cap = r.capacity
    mp, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
    tree = np.empty(2 * cap - 1)
    N.check(r.lib.srlx_per_backup(r.h_per, ctypes.byref(mp), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
    leaves = tree[cap - 1:]
    added = (steps - 1) * E  # the last lock-step's add is still deferred
    assert write.value == added % cap and mp.value == 1.0  # nobody called update(): max_priority is the initial 1.0
    slots = np.arange(added)
    idx = torch.tensor(slots + cap - 1, dtype=torch.int64, device="cuda")
    B = len(slots)
    obs = torch.zeros((B, n + 1, cfg.window_length, 84 * 84), dtype=torch.float32, device="cuda")
    act = torch.zeros((B, n), dtype=torch.int32, device="cuda")
    rew = torch.zeros((B, n), dtype=torch.float32, device="cuda")
    ter = torch.zeros((B, n), dtype=torch.float32, device="cuda")
    N.check(r.lib.srlx_store_gather_nstep(r.h_store, B, N.tptr(idx), N.tptr(obs), N.tptr(act), N.tptr(rew), N.tptr(ter), None))
    from simple_distributed_rl_amd.device.qnet import QNetInference

    off = torch.zeros((B, n + 1, cfg.window_length), dtype=torch.int64, device="cuda")
    N.check(r.lib.srlx_store_gather_items(r.h_store, B, N.tptr(idx), 0, n + 1, N.tptr(off), N.tptr(act), N.tptr(rew), N.tptr(ter), None))
    inf = QNetInference(eng.q_online, E, 0)  # launches of E rows like the acting passes: the same split-K shape, so the rows are the cached ones bit for bit
    rows = off.view(B * (n + 1), cfg.window_length)
    q = torch.cat([inf.forward_u8(r.obs_base, rows[k:k + E].contiguous()).clone() for k in range(0, rows.shape[0], E)]).view(B, n + 1, A).cpu().numpy()
    want_target = H.nstep_target(q[:, 1:], q[:, 1:], act.cpu().numpy(), rew.cpu().numpy(), ter.cpu().numpy(), None, cfg.discount, cfg.retrace_h, True, False)
    td = np.abs(want_target - q[np.arange(B), 0, act[:, 0].cpu().numpy()])
    want = (td.astype(np.float64) + cfg.memory_epsilon) ** cfg.memory_alpha
    got = leaves[slots]
    estimated = (got != 0.0) & (got != 1.0)
    # every lock-step adds E leaves: items inside an episode (estimated), items touching an episode end (max_priority = 1), positions without an item (0)
    assert estimated.sum() > 0.5 * B and (got == 1.0).sum() > 0 and (got == 0.0).sum() > 0
    np.testing.assert_allclose(got[estimated], want[estimated], rtol=1e-5, atol=1e-7)
    # an estimated item never has an episode end inside its window
    assert float(ter.cpu().numpy()[estimated].sum()) == 0.0
