"""GPU tests of PPO's actor-critic in libsrlx (csrc/srlx_ppo_net.hip; SURVEY 8 a20, BASELINE config 5; srl/algorithms/ppo/ppo.py:55-99,102-169,240-241,316-339,
389-404): the forward against the torch modules in float64, one minibatch's gradients against torch autograd of the same loss in float64, clip + Adam against
torch's, the one-launch rollout against the step-wise kernels it fuses (bit-exact), and the fused engine against the torch-autograd engine it replaces.
float32 work: tolerance 1e-5 relative (to a tensor's largest entry where sums cancel); the reference module needs TensorFlow -- parity UNPINNED, as for the whole row."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_ppo_gpu import LS_RANGE, _torch_loss  # noqa: E402

pytestmark = pytest.mark.gpu


def _env():
    import torch

    from simple_distributed_rl_amd import _native as N

    return N, N.lib(), torch, torch.device("cuda:0")


def _net(torch, dev, obs, A, seed):
    from simple_distributed_rl_amd.device.ppo import ActorCritic, PPODeviceConfig

    torch.manual_seed(seed)
    net = ActorCritic(PPODeviceConfig(obs_dim=obs, action_dim=A)).to(dev)
    with torch.no_grad():
        for p in net.parameters():  # (biases start at zero: give every tensor content)
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).contiguous()
    return net, flat


@pytest.mark.parametrize("obs,A,n", [(3, 1, 1000), (5, 3, 77), (8, 4, 4096)])
def test_forward_against_the_torch_modules_in_float64(obs, A, n):
    N, lib, torch, dev = _env()
    net, flat = _net(torch, dev, obs, A, 1)
    assert lib.srlx_ppo_net_param_count(obs, A) == flat.numel()
    x = torch.randn(n, obs, device=dev)
    v, loc, ls = torch.empty(n, device=dev), torch.empty(n, A, device=dev), torch.empty(n, A, device=dev)
    N.check(lib.srlx_ppo_net_forward(n, obs, A, N.tptr(flat), N.tptr(x), N.tptr(v), N.tptr(loc), N.tptr(ls), None))
    torch.cuda.synchronize()
    with torch.no_grad():
        v64, loc64, ls64 = net.double()(x.double())
    for got, want in ((v, v64), (loc, loc64), (ls, ls64)):
        torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    assert lib.srlx_ppo_net_param_count(9, 1) == -1 and lib.srlx_ppo_net_param_count(3, 5) == -1


@pytest.mark.parametrize("base,clip,vclip", [(1, 1, 1), (0, 0, 0)])
@pytest.mark.parametrize("obs,A,mb", [(3, 1, 8192), (5, 3, 1234)])
def test_minibatch_gradients_against_autograd_in_float64(base, clip, vclip, obs, A, mb):
    """d loss / d every parameter of one minibatch (rows drawn from a larger buffer, a count that is no multiple of the 32-sample tile) against autograd of the same
    loss through the torch modules in float64; the three reported losses; then clip + Adam against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam fed the kernel's
    gradient, two steps."""
    N, lib, torch, dev = _env()
    net, flat = _net(torch, dev, obs, A, 2)
    n = 3 * mb
    g = torch.Generator(device=dev).manual_seed(3)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    b_obs, b_act, b_adv, b_vt = r(n, obs), r(n, A), r(n), r(n)
    with torch.no_grad():
        v0, loc0, ls0 = net(b_obs)
    ls_c = torch.clamp(ls0, LS_RANGE[0], LS_RANGE[1])
    b_logp = (-0.5 * math.log(2 * math.pi) - ls_c - 0.5 * ((b_act - loc0) / torch.exp(ls_c)) ** 2 + 0.3 * r(n, A)).contiguous()
    b_val = (v0 + 0.3 * r(n)).contiguous()
    # rows away from the ReLU kinks: a pre-activation within float32 rounding of zero takes the other branch in float64 (measured: 1e-8 among 8192 x 256
    # pre-activations, which moved the value block's gradient by 5e-4 of its largest entry) -- a property of the yardstick's precision, not of the kernel
    import copy

    with torch.no_grad():
        n64 = copy.deepcopy(net).double()
        x64 = b_obs.double()
        z1 = n64.hidden_block[0](x64)
        z2 = n64.hidden_block[2](torch.relu(z1))
        h64 = torch.relu(z2)
        zmin = torch.stack([z.abs().min(dim=1).values for z in (z1, z2, n64.value_block[0](h64), n64.policy_block[0](h64))]).min(dim=0).values
    cand = torch.nonzero(zmin > 1e-5).reshape(-1)
    assert cand.numel() > 2 * mb
    rows = cand[torch.randperm(cand.numel(), device=dev, generator=g)[:mb]].contiguous()
    pc, vc, vw, ew = 0.2, 0.2, 0.7, 0.01
    P = flat.numel()
    partials = torch.zeros(lib.srlx_ppo_net_partials_floats(obs, A), device=dev)
    grad, losses = torch.zeros(P, device=dev), torch.zeros(3, device=dev)
    N.check(lib.srlx_ppo_net_minibatch(mb, N.tptr(rows), obs, A, N.tptr(flat), N.tptr(b_obs), N.tptr(b_act), N.tptr(b_logp), N.tptr(b_adv), N.tptr(b_vt), N.tptr(b_val),
                                       LS_RANGE[0], LS_RANGE[1], base, clip, pc, vclip, vc, vw, ew, N.tptr(partials), N.tptr(grad), N.tptr(losses), None))
    torch.cuda.synchronize()
    d = torch.float64
    net64 = net.double()
    v, loc, ls = net64(b_obs[rows].to(d))
    lsc = torch.clamp(ls, LS_RANGE[0], LS_RANGE[1])
    lp = -0.5 * math.log(2 * math.pi) - lsc - 0.5 * ((b_act[rows].to(d) - loc) / torch.exp(lsc)) ** 2
    parts = _torch_loss(torch, lp, b_logp[rows].to(d), b_adv[rows].to(d), v, b_vt[rows].to(d), b_val[rows].to(d), base, clip, pc, vclip, vc, vw, ew)
    sum(parts).backward()
    torch.testing.assert_close(losses.double(), torch.stack([p.detach() for p in parts]), rtol=1e-4, atol=1e-6)
    off = 0
    for name, p in net64.named_parameters():
        got, want = grad[off : off + p.numel()].double(), p.grad.reshape(-1)
        off += p.numel()
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-12, msg=lambda m: f"{name}: {m}")
    # ---- clip + Adam ----
    ref = flat.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=3e-4)
    m, v2, step = torch.zeros(P, device=dev), torch.zeros(P, device=dev), torch.zeros(2, dtype=torch.int64, device=dev)
    mine = flat.clone()
    for k in range(2):
        gk = grad.clone() * (1.0 + k)
        ref.grad = gk.clone() * 0.5  # (grad_scale 0.5: the sum over two ranks -> their mean)
        norm = torch.nn.utils.clip_grad_norm_([ref], 0.01)
        opt.step()
        N.check(lib.srlx_ppo_net_adam(obs, A, N.tptr(mine), N.tptr(gk), N.tptr(m), N.tptr(v2), N.tptr(step), 3e-4, 0.9, 0.999, 1e-8, 0.01, 0.5, None))
        torch.cuda.synchronize()
        assert float(norm) > 0.01  # (the clip is active)
        torch.testing.assert_close(mine, ref.detach(), rtol=3e-7, atol=3e-4 * 2e-5)  # (an ulp of the parameter, or 2e-5 of the step)
    assert step.tolist() == [2, 0]


def _engines(torch, E, T, seed, **kw):
    from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine

    cfg = PPODeviceConfig(n_envs=E, horizon=T, seed=seed, **kw)
    return cfg, PPOEngine


def test_one_launch_rollout_equals_the_stepwise_kernels():
    """k_ppo_rollout against the launches it fuses (network forward -> srlx_ppo_normal_act -> srlx_pendulum_step per step, then srlx_gae_scan), same seeds: every
    buffer, the environments' state and the episode bookkeeping -- bit for bit (one definition of the arithmetic, srlx_ppo_math.h); episodes end inside the
    rollout (episode_len 11 < T)."""
    N, lib, torch, dev = _env()
    cfg, PPOEngine = _engines(torch, 272, 24, 4, episode_len=11)
    a = PPOEngine(cfg, 0)
    b = PPOEngine(cfg, 0)
    assert a.fused and b.fused and a._fused_rollout_ok()
    b._fused_rollout_ok = lambda: False  # the step-wise path on the libsrlx network
    for it in range(2):
        a.rollout()
        b.rollout()
        torch.cuda.synchronize()
        for name in ("b_obs", "b_act", "b_logp", "b_val", "b_rew", "b_done", "b_adv", "episode_return"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (it, name)
        assert torch.equal(a._last_v, b._last_v) and torch.equal(a.env.state, b.env.state) and torch.equal(a.env.t, b.env.t)
        assert int(a.act_counter.item()) == int(b.act_counter.item()) == cfg.horizon * (it + 1) and int(a.env.counter.item()) == int(b.env.counter.item())
        torch.testing.assert_close(a.finished_returns, b.finished_returns, rtol=1e-5, atol=1e-3)  # (float atomics: order differs)
        assert float(a.finished_returns[1]) == float(cfg.n_envs * ((it + 1) * cfg.horizon // cfg.episode_len))
        b.b_obs[0].copy_(b.b_obs[cfg.horizon])  # (the step-wise path starts from b_obs[0], the fused one from env.obs)
    assert torch.equal(a.env.obs, a.b_obs[cfg.horizon])


@pytest.mark.parametrize("v_target", ["gae", "return"])
def test_fused_engine_against_the_autograd_engine(v_target):
    """One whole iteration (rollout + 4 epochs x 4 minibatches) of the fused engine against the torch-modules / autograd / torch.optim.Adam engine from the same
    initial parameters and seeds: the rollouts agree to float32 rounding of the networks' sums, the parameters after the 16 optimiser steps to a small fraction of
    the steps' size (lr = 2e-4 per step)."""
    N, lib, torch, dev = _env()
    cfg, PPOEngine = _engines(torch, 512, 16, 6, v_target=v_target)
    a, b = PPOEngine(cfg, 0, fused=True), PPOEngine(cfg, 0, fused=False)
    with torch.no_grad():
        for p, q in zip(a.net.parameters(), b.net.parameters()):
            assert torch.equal(p, q)  # (same seed, same initialisation)
    assert a.flat.data_ptr() == next(a.net.parameters()).data_ptr()  # the module's tensors are views of the flat vector
    a.rollout()
    b.rollout()
    torch.cuda.synchronize()
    for name in ("b_act", "b_logp", "b_val", "b_rew", "b_adv"):
        torch.testing.assert_close(getattr(a, name), getattr(b, name), rtol=2e-4, atol=2e-4, msg=lambda m: f"{name}: {m}")
    # the update on IDENTICAL buffers
    for name in ("b_obs", "b_act", "b_logp", "b_val", "b_rew", "b_done", "b_adv"):
        getattr(b, name).copy_(getattr(a, name))
    before = a.flat.clone()
    a.update()
    b.update()
    torch.cuda.synchronize()
    moved = float((a.flat - before).abs().max())
    assert moved > 1e-3  # 16 steps of about lr each
    flat_b = torch.cat([p.detach().reshape(-1) for p in b.net.parameters()])
    diff = (a.flat - flat_b).abs()  # (Adam divides by sqrt(v): an entry whose gradients are rounding residue may step differently -- bounded by a few % of the movement)
    assert float(diff.max()) < 0.03 * moved and float(diff.mean()) < 2e-4 * moved, (float(diff.max()), float(diff.mean()), moved)
    torch.testing.assert_close(a.losses, b.losses, rtol=1e-3, atol=1e-5)
    assert a.opt_step.tolist() == [cfg.epochs * cfg.minibatches, 0]


def test_fused_engine_graphs_and_geometry_gate():
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine

    with pytest.raises(ValueError):
        PPOEngine(PPODeviceConfig(n_envs=64, hidden_sizes=(32, 32)), 0, fused=True)
    long = PPOEngine(PPODeviceConfig(n_envs=32, horizon=400, epochs=1, minibatches=2), 0)  # a horizon whose per-step records do not fit the rollout kernel's LDS
    assert long.fused and not long._fused_rollout_ok() and lib.srlx_ppo_net_rollout_max_horizon(1) < 400
    long.step()  # (the step-wise kernels on the libsrlx network)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(long.flat).all()) and int(long.act_counter.item()) == 400
    assert not PPOEngine(PPODeviceConfig(n_envs=64, hidden_sizes=(32, 32)), 0).fused  # (other blocks: the torch modules)

    def run(graphs):
        eng = PPOEngine(PPODeviceConfig(n_envs=1024, horizon=16, seed=9), 0)
        for k in range(7):
            if k == 2 and graphs:
                eng.capture_graphs()  # (runs one whole iteration itself, as its warm-up)
                continue
            eng.step()
        torch.cuda.synchronize()
        return eng

    a, b = run(False), run(True)
    assert torch.equal(a.flat, b.flat) and torch.equal(a.b_adv, b.b_adv)  # eager launches == graph replays, bit for bit
    assert bool(torch.isfinite(a.flat).all()) and all(np.isfinite(list(a.info().values())))


_DP_NCCL_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SRLX_ROOT"])
from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, DistributedPPO, PPOEngine
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"], rank=0, world_size=1)
cfg = PPODeviceConfig(n_envs=512, horizon=16, epochs=2, minibatches=2, seed=5)
def run(graphs):
    dp = DistributedPPO(cfg, 0)
    assert dp.engine.fused
    for k in range(5):
        if k == 2 and graphs:
            dp.capture_graphs()  # (one whole iteration as its warm-up)
            assert dp.engine._update_graph is not None
            continue
        dp.step()
    torch.cuda.synchronize()
    return dp.engine.flat.clone()
a, b = run(False), run(True)
assert torch.equal(a, b), float((a - b).abs().max())
assert torch.isfinite(a).all()
dist.destroy_process_group()
print("ok")
"""


def test_data_parallel_update_graph_holds_the_all_reduces(tmp_path):
    """DistributedPPO over RCCL (one rank: the 1-GPU test box): the update graph is captured WITH its all-reduces of the flat gradient inside (between the gradient
    reduction and the clip + Adam launch of every minibatch), and replays equal the eager launches bit for bit."""
    import subprocess

    script = tmp_path / "dp_nccl.py"
    script.write_text(_DP_NCCL_WORKER)
    env = dict(os.environ, SRLX_ROOT=ROOT, PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
