"""GPU tests of the PPO row (BASELINE config 5): Normal-policy act kernel, fused loss + gradient-seed
kernels against the oracle restatement (forward) and torch autograd of the same formula (seeds), the Pendulum-shaped
vector environment, the E-environment engine end to end (it learns), and the data-parallel wrapper on 2 ranks.
The reference module needs TensorFlow: parity UNPINNED (restated from source lines); tolerance 1e-5 relative."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hot_path_oracle as H  # noqa: E402

pytestmark = pytest.mark.gpu
LS_RANGE = (math.log(1e-10), math.log(10))


def _env():
    import torch

    from simple_distributed_rl_amd import _native as N

    return N, N.lib(), torch, torch.device("cuda:0")


def test_normal_act_vs_oracle():
    N, lib, torch, dev = _env()
    rng = np.random.default_rng(0)
    n, seed = 5000, 1234
    loc = rng.standard_normal(n).astype(np.float32)
    ls = (rng.standard_normal(n) * 1.5).astype(np.float32)
    ls[:10] = 5.0  # beyond the stable-gradient range: clipped to log(10)
    loc_t, ls_t = torch.as_tensor(loc, device=dev), torch.as_tensor(ls, device=dev)
    act, lp = torch.empty(n, device=dev), torch.empty(n, device=dev)
    counter = torch.full((1,), 7, dtype=torch.int64, device=dev)
    N.check(lib.srlx_ppo_normal_act(n, N.tptr(loc_t), N.tptr(ls_t), LS_RANGE[0], LS_RANGE[1], seed, N.tptr(counter), 0, N.tptr(act), N.tptr(lp), None))
    torch.cuda.synchronize()
    assert int(counter.item()) == 8
    i = np.arange(n, dtype=np.uint64)
    u1 = 1.0 - H.u53(H.rng_u64(seed, np.uint64(7), 2 * i))
    u2 = H.u53(H.rng_u64(seed, np.uint64(7), 2 * i + np.uint64(1)))
    z = (np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)).astype(np.float32)
    lsc = np.clip(ls, np.float32(LS_RANGE[0]), np.float32(LS_RANGE[1]))
    want_a = loc + np.exp(lsc) * z
    np.testing.assert_allclose(act.cpu().numpy(), want_a, rtol=1e-5, atol=1e-5)
    want_lp = np.maximum(H.normal_logprob(act.cpu().numpy(), loc, lsc), np.float32(math.log(1e-6)))
    np.testing.assert_allclose(lp.cpu().numpy(), want_lp, rtol=1e-5, atol=2e-5)
    assert abs(float(np.mean(z))) < 0.05 and abs(float(np.std(z)) - 1) < 0.05  # it is a standard normal
    N.check(lib.srlx_ppo_normal_act(n, N.tptr(loc_t), N.tptr(ls_t), LS_RANGE[0], LS_RANGE[1], seed, None, 1, N.tptr(act), N.tptr(lp), None))
    np.testing.assert_array_equal(act.cpu().numpy(), loc)  # evaluation: the mean (ppo.py:318-319)


def _torch_loss(torch, lp, olp, adv, v, vt, ov, base, clip, pc, vclip, vc, vw, ew):
    adv = adv[:, None] - v.detach()[:, None] if base else adv[:, None]
    ratio = torch.exp(lp - olp)
    pol = torch.minimum(ratio * adv, torch.clamp(ratio, 1 - pc, 1 + pc) * adv) if clip else ratio * adv
    policy_loss = -pol.mean()
    if vclip:
        v_c = torch.maximum(torch.minimum(v, ov + vc), ov - vc)
        value = torch.maximum((v - vt) ** 2, (v_c - vt) ** 2)
    else:
        value = (v - vt) ** 2
    return policy_loss, vw * value.mean(), ew * -(-(torch.exp(lp) * lp)).sum(-1).mean()


@pytest.mark.parametrize("base,clip,vclip", [(1, 1, 1), (0, 1, 0), (1, 0, 1), (0, 0, 0)])
@pytest.mark.parametrize("B,K", [(4096, 1), (1000, 3)])
def test_ppo_loss_kernels_vs_oracle_and_autograd(base, clip, vclip, B, K):
    N, lib, torch, dev = _env()
    rng = np.random.default_rng(B + K + base)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    loc, ls, act = f(B, K), f(B, K) * 0.5, f(B, K)
    ls[:5] = 4.0  # outside the stable-gradient range: no gradient to log_scale there
    olp = (H.normal_logprob(act, loc, np.clip(ls, *np.float32(LS_RANGE))) + 0.3 * f(B, K)).astype(np.float32)
    adv, v, vt = f(B), f(B), f(B)
    ov = (v + 0.3 * f(B)).astype(np.float32)
    pc, vc, vw, ew = 0.2, 0.2, 0.7, 0.01
    t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
    T = dict(loc=t(loc), ls=t(ls), act=t(act), olp=t(olp), adv=t(adv), v=t(v), vt=t(vt), ov=t(ov))
    losses = torch.zeros(3, device=dev)
    g_loc, g_ls, g_v = torch.empty((B, K), device=dev), torch.empty((B, K), device=dev), torch.empty(B, device=dev)
    N.check(lib.srlx_ppo_loss_normal(B, K, N.tptr(T["loc"]), N.tptr(T["ls"]), LS_RANGE[0], LS_RANGE[1], N.tptr(T["act"]), N.tptr(T["olp"]), N.tptr(T["adv"]),
                                     N.tptr(T["v"]), N.tptr(T["vt"]), N.tptr(T["ov"]), base, clip, pc, vclip, vc, vw, ew, N.tptr(losses), N.tptr(g_loc),
                                     N.tptr(g_ls), N.tptr(g_v), None))
    torch.cuda.synchronize()
    lsc = np.clip(ls, *np.float32(LS_RANGE))
    want = H.ppo_loss(H.normal_logprob(act, loc, lsc), olp, adv, v, vt, ov, base, clip, pc, vclip, vc, vw, ew)
    np.testing.assert_allclose(losses.cpu().numpy(), np.array(want), rtol=1e-4, atol=1e-6)
    # gradient seeds == autograd of the same formula (float64 graph as the yardstick)
    d = torch.float64
    loc_g, ls_g, v_g = (T[k].to(d).requires_grad_() for k in ("loc", "ls", "v"))
    lsc_g = torch.clamp(ls_g, LS_RANGE[0], LS_RANGE[1])
    lp_g = -0.5 * math.log(2 * math.pi) - lsc_g - 0.5 * ((T["act"].to(d) - loc_g) / torch.exp(lsc_g)) ** 2
    parts = _torch_loss(torch, lp_g, T["olp"].to(d), T["adv"].to(d), v_g, T["vt"].to(d), T["ov"].to(d), base, clip, pc, vclip, vc, vw, ew)
    sum(parts).backward()
    scale = 1.0 / B
    np.testing.assert_allclose(g_loc.cpu().numpy(), loc_g.grad.cpu().numpy(), rtol=2e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(g_ls.cpu().numpy(), ls_g.grad.cpu().numpy(), rtol=2e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(g_v.cpu().numpy(), v_g.grad.cpu().numpy(), rtol=2e-4, atol=1e-5 * scale)
    assert torch.all(g_ls[:5] == 0)
    # the generic variant on the same log-probabilities
    lp32 = t(H.normal_logprob(act, loc, lsc))
    losses2, g_lp, g_v2 = torch.zeros(3, device=dev), torch.empty((B, K), device=dev), torch.empty(B, device=dev)
    N.check(lib.srlx_ppo_loss_logpi(B, K, N.tptr(lp32), N.tptr(T["olp"]), N.tptr(T["adv"]), N.tptr(T["v"]), N.tptr(T["vt"]), N.tptr(T["ov"]), base, clip, pc,
                                    vclip, vc, vw, ew, N.tptr(losses2), N.tptr(g_lp), N.tptr(g_v2), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses2.cpu().numpy(), np.array(want), rtol=1e-4, atol=1e-6)
    lp_leaf = lp32.to(d).requires_grad_()
    sum(_torch_loss(torch, lp_leaf, T["olp"].to(d), T["adv"].to(d), T["v"].to(d), T["vt"].to(d), T["ov"].to(d), base, clip, pc, vclip, vc, vw, ew)).backward()
    np.testing.assert_allclose(g_lp.cpu().numpy(), lp_leaf.grad.cpu().numpy(), rtol=2e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(g_v2.cpu().numpy(), g_v.cpu().numpy(), rtol=1e-6, atol=0)


def test_pendulum_step_vs_oracle():
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.ppo import PendulumVecEnv

    E, L = 777, 5
    env = PendulumVecEnv(E, L, seed=3, device=dev)
    state, tt = env.state.cpu().numpy().copy(), np.zeros(E, np.int64)
    rng = np.random.default_rng(1)
    obs, rew, done = torch.empty((E, 3), device=dev), torch.empty(E, device=dev), torch.empty(E, dtype=torch.uint8, device=dev)
    for step in range(12):
        a = (rng.standard_normal(E) * 2).astype(np.float32)
        env.step(torch.as_tensor(a, device=dev), obs, rew, done)
        ns, nt, o_obs, o_rew, o_done = H.pendulum_step(state, tt, a, L)
        np.testing.assert_allclose(rew.cpu().numpy(), o_rew, rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(done.cpu().numpy().astype(bool), o_done)
        got_state = env.state.cpu().numpy()
        if o_done.any():  # time limit: every env resets together here (same episode clock)
            assert o_done.all() and (step + 1) % L == 0
            assert np.all(np.abs(got_state[:, 0]) <= np.pi) and np.all(np.abs(got_state[:, 1]) <= 1)
            assert len(np.unique(got_state[:, 0])) > E // 2
            state, tt = got_state.copy(), np.zeros(E, np.int64)
        else:
            np.testing.assert_allclose(got_state, ns, rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(obs.cpu().numpy(), o_obs, rtol=1e-5, atol=1e-5)
            state, tt = ns, nt
        np.testing.assert_allclose(obs.cpu().numpy(), np.stack([np.cos(got_state[:, 0]), np.sin(got_state[:, 0]), got_state[:, 1]], 1), rtol=1e-5, atol=1e-5)


@pytest.mark.slow
def test_ppo_engine_learns_pendulum():
    """The engine with the reference's default hyper-parameters (ppo/config.py:43-110: gamma = lambda = 0.9, the GAE
    value as v_target AND advantage, value clipping) improves the mean episode return of 1024 Pendulum environments:
    10 M environment steps in a few seconds (measured curve: -1330 -> -470)."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine

    eng = PPOEngine(PPODeviceConfig(n_envs=1024, horizon=50, seed=1), 0)
    first = None
    for it in range(200):
        eng.step()
        if (it + 1) % 20 == 0:
            r = eng.pop_mean_episode_return()
            first = r if first is None else first
    info = eng.info()
    assert all(np.isfinite(list(info.values()))), info
    assert r > first + 300, (first, r)


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SRLX_ROOT"])
from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, DistributedPPO
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["PORT"], rank=int(os.environ["RANK"]), world_size=2)
dp = DistributedPPO(PPODeviceConfig(n_envs=256, horizon=16, epochs=2, minibatches=2, seed=5), 0)
for _ in range(3):
    dp.step()
flat = torch.cat([p.detach().reshape(-1) for p in dp.engine.net.parameters()]).cpu()
obs = dp.engine.b_obs[0].cpu()
both = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(both, flat)
obs2 = [torch.empty_like(obs) for _ in range(2)]
dist.all_gather(obs2, obs)
assert torch.equal(both[0], both[1]), "parameters diverged across ranks"
assert not torch.equal(obs2[0], obs2[1]), "ranks must run different environments"
assert torch.isfinite(flat).all()
print("rank", dist.get_rank(), "ok")
"""


def test_data_parallel_ppo_two_ranks_one_gpu(tmp_path):
    """BASELINE config 5 topology on the test box: 2 ranks (sharing the GPU, gloo rendezvous) keep identical
    parameters through averaged gradients while stepping disjoint environments."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    port = str(29700 + os.getpid() % 200)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, SRLX_ROOT=ROOT, RANK=str(r), PORT=port), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_ppo_engine_hip_graphs():
    """The whole rollout (T steps + GAE) and the whole update phase replay as two HIP graphs: environments advance,
    parameters move, losses stay finite, episodes finish on schedule."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine

    eng = PPOEngine(PPODeviceConfig(n_envs=512, horizon=25, seed=2), 0)
    for _ in range(2):
        eng.step()
    eng.capture_graphs()
    before = [p.detach().clone() for p in eng.net.parameters()]
    obs0 = eng.b_obs[0].clone()
    eng.pop_mean_episode_return()
    for _ in range(16):  # 400 environment steps: every environment finishes two 200-step episodes
        eng.step()
    torch.cuda.synchronize()
    assert int(eng.finished_returns[1].item()) == 2 * 512
    assert np.isfinite(eng.pop_mean_episode_return())
    assert all(np.isfinite(list(eng.info().values())))
    assert any(float((p - q).abs().max()) > 0 for p, q in zip(eng.net.parameters(), before))
    assert not torch.equal(eng.b_obs[0], obs0)


def test_bench_ppo_line():
    """`bench.py --algo ppo` prints the contract's JSON line for the configs[4] workload (one GPU; two gloo ranks sharing it)."""
    import json
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for extra, n in ((["--gpus", "1"], 1), (["--gpus", "2", "--backend", "gloo"], 2)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--algo", "ppo", "--envs", "256", "--steps", "3", "--warmup", "2"] + extra, cwd=root,
                           capture_output=True, text=True, timeout=900, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and "PPO" in d["config"]["workload"] and d["config"]["envs_per_gpu"] == 256
        assert d["value"] > 0 and d["learner_updates_per_s"] > 0 and d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1  # (the libsrlx network's minibatch kernel)
        assert "libsrlx" in d["config"]["networks"] and "k_ppo_minibatch" in d["roofline"]["kernel"]
        assert all(np.isfinite(v) for v in d["final"].values())


def test_graph_replays_do_not_depend_on_host_synchronisation():
    """PPOEngine.step() with captured graphs, iterations queued back to back (nothing waits on the host) vs one host synchronisation per iteration: the
    same training, finite at E = 4096.  The engine draws its minibatch permutations eagerly, outside the captured update: torch.randperm as a graph
    node was what made unsynchronised replays diverge (tools/ppo_replay_bisect.py, tools/randperm_graph_repro.py; device/ppo.py:__init__)."""
    import torch

    from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine

    def run(E, sync):
        eng = PPOEngine(PPODeviceConfig(n_envs=E, seed=0), 0)
        for _ in range(2):
            eng.step()
        eng.capture_graphs()
        eng.step()
        torch.cuda.synchronize()
        for _ in range(40):
            eng.step()
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(p).all()) for p in eng.net.parameters())
        return eng.info()

    for E in (256, 4096):
        a, b = run(E, True), run(E, False)
        for k in a:
            assert math.isfinite(b[k]) and abs(a[k] - b[k]) <= 1e-4 * abs(a[k]) + 1e-6, (E, k, a, b)


def test_keyed_permutation_kernel():
    """srlx_rng_permutation: a valid permutation of 0..n-1 for awkward n, a different one at every call (the device counter advances), the same one for
    the same (seed, counter), and every position equally likely to receive any value (chi-square of one position over 4000 draws)."""
    import torch

    from simple_distributed_rl_amd import _native as N

    lib = N.lib()
    dev = torch.device("cuda:0")
    for n in (1, 2, 5, 1000, 4097, 32 * 4096):
        counter = torch.zeros(1, dtype=torch.int64, device=dev)
        out = torch.empty(n, dtype=torch.int64, device=dev)
        seen = []
        for k in range(3):
            N.check(lib.srlx_rng_permutation(1234, N.tptr(counter), n, N.tptr(out), None))
            torch.cuda.synchronize()
            assert torch.equal(out.sort().values, torch.arange(n, device=dev)) and int(counter) == k + 1
            seen.append(out.clone())
        if n > 5:
            assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
        counter.zero_()
        N.check(lib.srlx_rng_permutation(1234, N.tptr(counter), n, N.tptr(out), None))
        assert torch.equal(out, seen[0])
        counter.zero_()  # the three of them in one launch (srlx_rng_permutations): the same values, the counter advanced by three
        out3 = torch.empty((3, n), dtype=torch.int64, device=dev)
        N.check(lib.srlx_rng_permutations(1234, N.tptr(counter), n, 3, N.tptr(out3), None))
        assert all(torch.equal(out3[k], seen[k]) for k in range(3)) and int(counter) == 3
    n, draws = 16, 4000
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    hist = np.zeros((n, n))
    for _ in range(draws):
        N.check(lib.srlx_rng_permutation(7, N.tptr(counter), n, N.tptr(out), None))
        hist[np.arange(n), out.cpu().numpy()] += 1
    chi2 = ((hist - draws / n) ** 2 / (draws / n)).sum(axis=1)  # 15 degrees of freedom per position: 99.99 % quantile = 44
    assert chi2.max() < 60, chi2
