"""GPU tests of the NoisyLinear path of the matrix-core Q-network (reference: srl/rl/torch_/modules/noisy_linear.py:26-52,
turned on by rainbow.Config.set_atari_config(), rainbow.py:116-148).

Parity with the reference's noise is statistical by nature (its normals come from torch's generator): exact checks are made
where exactness exists -- sigma = 0 reproduces the plain network bit for bit; given the effective weights of a draw (read back
from the library) the forward equals torch on those weights, and the gradients of mu AND sigma equal torch autograd of
`mu + sigma * eps` with the same eps -- and the noise itself is checked as a distribution (moments, independence between draws,
Q statistics against the reference module tree under the same mu / sigma)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _ring(rows, n_frames=64, seed=1):
    F = 84 * 84
    g = torch.Generator(device="cuda").manual_seed(seed)
    ring = torch.randint(0, 256, (n_frames * F,), dtype=torch.uint8, device="cuda", generator=g)
    sel = torch.randint(0, n_frames, (rows, 4), device="cuda", generator=g)
    frames = ring.view(n_frames, 84, 84)[sel].float() / 255
    return ring, sel * F, frames, g


def _noisy_net(A=6, hidden=512, seed=0):
    from simple_distributed_rl_amd.device.qnet import EngineQNet

    torch.manual_seed(seed)
    return EngineQNet(A, (84, 84), 4, hidden, 32, "average", noisy=True).cuda()


def _torch_q(net, frames, eff):
    """The network's forward with the dense layers' weights replaced by the given effective tensors."""
    import torch.nn.functional as F

    x = F.relu(net.conv1(frames))
    x = F.relu(net.conv2(x))
    x = F.relu(net.conv3(x)).permute(0, 2, 3, 1).flatten(1)
    h = F.relu(F.linear(x, eff[0].view_as(net.fc1.weight), eff[1]))
    v = F.linear(h[:, : net.hidden], eff[2].view_as(net.v2.weight), eff[3])
    adv = F.linear(h[:, net.hidden :], eff[4].view_as(net.a2.weight), eff[5])
    return v + adv - adv.mean(dim=-1, keepdim=True)


def test_reference_initialisation_and_state_dict_round_trip():
    """sigma_0 = 0.5 / sqrt(fan_in) (noisy_linear.py:29-33) arrives through the reference's keys, and the fused / NHWC layouts
    convert back to them losslessly."""
    from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

    net = _noisy_net()
    assert torch.allclose(net.fc1_sigma_w, torch.full_like(net.fc1_sigma_w, 0.5 / 7744 ** 0.5))
    assert torch.allclose(net.v2_sigma_w, torch.full_like(net.v2_sigma_w, 0.5 / 512 ** 0.5))
    ref = atari_qnetwork(6, (84, 84), 4, 512, True, 32).cuda()
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    for k in sd:
        if "sigma" in k:
            sd[k] = torch.rand_like(sd[k]) * 0.1
    net.load_reference_state_dict(sd)
    back = net.reference_state_dict()
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k]), k


def test_sigma_zero_is_the_plain_network_bit_for_bit():
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    noisy = _noisy_net(seed=2)
    with torch.no_grad():
        for n in ("fc1", "v2", "a2"):
            getattr(noisy, n + "_sigma_w").zero_()
            getattr(noisy, n + "_sigma_b").zero_()
    plain = EngineQNet(6, (84, 84), 4, 512, 32, "average").cuda()
    plain.load_state_dict({k: v for k, v in noisy.state_dict().items() if "sigma" not in k})
    ring, off, _, _ = _ring(40)
    qa = QNetInference(noisy, 64).forward_u8(ring.data_ptr(), off).clone()
    qb = QNetInference(plain, 64).forward_u8(ring.data_ptr(), off).clone()
    torch.cuda.synchronize()
    assert torch.equal(qa, qb)


def test_draws_are_standard_normal_independent_and_counted():
    from simple_distributed_rl_amd.device.qnet import QNetInference

    net = _noisy_net()
    qn = QNetInference(net, 64, noise_seed=5)
    ring, off, frames, _ = _ring(8)
    eps, ids = [], []
    for _ in range(3):
        qn.forward_u8(ring.data_ptr(), off)
        w, d = qn.effective(0)
        eps.append(((w.view_as(net.fc1.weight) - net.fc1.weight) / net.fc1_sigma_w).detach())
        ids.append(d)
    assert ids == [0, 1, 2]  # one draw per forward call
    e = eps[0].double()
    n = e.numel()  # 7.9 M normals
    assert abs(float(e.mean())) < 4 / n ** 0.5 and abs(float(e.var()) - 1) < 4 * (2 / n) ** 0.5
    assert abs(float((e ** 3).mean())) < 4 * (15 / n) ** 0.5 and abs(float((e ** 4).mean()) - 3) < 4 * (96 / n) ** 0.5
    assert float(e.abs().max()) > 4.5  # tails exist (Box-Muller on 24-bit uniforms reaches 5.7 sigma)
    for a, b in ((eps[0], eps[1]), (eps[1], eps[2])):
        assert abs(float((a.double() * b.double()).mean())) < 4 / n ** 0.5  # independent draws
    assert abs(float((e[:, :-1] * e[:, 1:]).mean())) < 4 / n ** 0.5  # neighbours (the two outputs of one Box-Muller pair) are uncorrelated
    other = QNetInference(net, 64, noise_seed=6)
    other.forward_u8(ring.data_ptr(), off)
    w2, _ = other.effective(0)
    assert not torch.equal(w2, qn.effective(0)[0])  # another handle = another stream
    # the same (seed, draw) reproduces: a fresh handle with seed 5 starts at draw 0 again
    again = QNetInference(net, 64, noise_seed=5)
    again.forward_u8(ring.data_ptr(), off)
    assert torch.equal(((again.effective(0)[0].view_as(net.fc1.weight) - net.fc1.weight) / net.fc1_sigma_w), eps[0])


def test_forward_equals_torch_on_the_effective_weights():
    from simple_distributed_rl_amd.device.qnet import QNetInference

    net = _noisy_net(A=9, seed=4)
    with torch.no_grad():  # sizeable noise
        for nme in ("fc1", "v2", "a2"):
            getattr(net, nme + "_sigma_w").mul_(4.0)
            getattr(net, nme + "_sigma_b").mul_(4.0)
    qn = QNetInference(net, 128)
    ring, off, frames, _ = _ring(100)
    q = qn.forward_u8(ring.data_ptr(), off).clone()
    eff = [qn.effective(k)[0] for k in range(6)]
    with torch.no_grad():
        want = _torch_q(net, frames, eff)
    torch.cuda.synchronize()
    np.testing.assert_allclose(q.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("B,stride", [(32, 4), (8, 1)])
def test_training_pass_gradients_of_mu_and_sigma_match_autograd(B, stride):
    """forward over B*stride rows, redraw for the s_0 rows, backward: every gradient (12 mu / plain tensors + 6 sigmas) equals torch
    autograd of the same computation with eps read back from the library; the no-grad rows keep the FIRST draw's values."""
    from simple_distributed_rl_amd.device.qnet import QNetInference

    net = _noisy_net(seed=7)
    rows = B * stride
    qn = QNetInference(net, max(rows, 64)).enable_training(64)
    ring, off, frames, g = _ring(rows, seed=3)
    q1 = qn.forward_u8(ring.data_ptr(), off).clone()
    eff1 = [qn.effective(k)[0] for k in range(6)]
    q2 = qn.redraw_rows(B, stride).clone()[:rows]
    eff2, ids = zip(*[qn.effective(k) for k in range(6)])
    assert set(ids) == {1}
    torch.cuda.synchronize()
    mask = torch.zeros(rows, dtype=torch.bool, device="cuda")
    mask[::stride] = True
    with torch.no_grad():
        w1, w2 = _torch_q(net, frames, eff1), _torch_q(net, frames, list(eff2))
    tol = dict(rtol=1e-5, atol=1e-5 * float(w1.abs().max()))
    np.testing.assert_allclose(q2[mask].cpu().numpy(), w2[mask].cpu().numpy(), **tol)  # s_0 rows: second draw
    if stride > 1:
        np.testing.assert_allclose(q2[~mask].cpu().numpy(), w1[~mask].cpu().numpy(), **tol)  # the others: untouched first draw
        assert torch.equal(q2[~mask], q1[~mask])
    # autograd with the same eps
    mus = [net.fc1.weight, net.fc1.bias, net.v2.weight, net.v2.bias, net.a2.weight, net.a2.bias]
    sigs = [net.fc1_sigma_w, net.fc1_sigma_b, net.v2_sigma_w, net.v2_sigma_b, net.a2_sigma_w, net.a2_sigma_b]
    eps = [((e.view_as(m) - m) / s).detach() for e, m, s in zip(eff2, mus, sigs)]
    net.zero_grad(set_to_none=True)
    want_q = _torch_q(net, frames[mask], [m + s * e for m, s, e in zip(mus, sigs, eps)])
    grad_q = torch.randn((B, 6), device="cuda", generator=g)
    want_q.backward(grad_q)
    want = [p.grad.detach().clone() for p in qn._params()]
    qn.enable_training(64)  # fresh zeroed static gradient tensors (18 of them)
    qn.backward_u8(ring.data_ptr(), off, grad_q, sample_stride=stride)
    torch.cuda.synchronize()
    names = ["conv1.w", "conv1.b", "conv2.w", "conv2.b", "conv3.w", "conv3.b", "fc1.mu_w", "fc1.mu_b", "v2.mu_w", "v2.mu_b", "a2.mu_w", "a2.mu_b",
             "fc1.sigma_w", "fc1.sigma_b", "v2.sigma_w", "v2.sigma_b", "a2.sigma_w", "a2.sigma_b"]
    assert len(qn._params()) == 18
    for name, p, w in zip(names, qn._params(), want):
        scale = float(w.abs().max()) + 1e-12
        np.testing.assert_allclose(p.grad.cpu().numpy(), w.cpu().numpy(), rtol=2e-4, atol=3e-5 * scale, err_msg=name)


def test_q_statistics_match_the_reference_module():
    """Mean and spread of Q over many draws: libsrlx vs the mirrored reference module tree (torch NoisyLinear, noisy_linear.py:35-52)
    holding the same mu / sigma.  Per (row, action) the two sample means differ by a few standard errors at most."""
    from simple_distributed_rl_amd.device.qnet import QNetInference
    from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

    net = _noisy_net(seed=9)
    with torch.no_grad():
        for nme in ("fc1", "v2", "a2"):
            getattr(net, nme + "_sigma_w").mul_(3.0)
            getattr(net, nme + "_sigma_b").mul_(3.0)
    ref = atari_qnetwork(6, (84, 84), 4, 512, True, 32).cuda()
    ref.load_state_dict(net.reference_state_dict())
    qn = QNetInference(net, 64)
    ring, off, frames, _ = _ring(16, seed=11)
    K = 300
    torch.manual_seed(1)
    mine = torch.stack([qn.forward_u8(ring.data_ptr(), off).clone() for _ in range(K)]).double()
    with torch.no_grad():
        theirs = torch.stack([ref(frames, channels_first=True) for _ in range(K)]).double()
    torch.cuda.synchronize()
    sd_m, sd_t = mine.std(0), theirs.std(0)
    assert float(sd_m.min()) > 0  # the noise reaches Q
    se = (sd_m ** 2 / K + sd_t ** 2 / K).sqrt()
    z = ((mine.mean(0) - theirs.mean(0)) / se).abs()
    assert float(z.max()) < 5.0, float(z.max())  # 96 comparisons, |z| < 5
    ratio = sd_m / sd_t
    assert 0.8 < float(ratio.min()) and float(ratio.max()) < 1.25, (float(ratio.min()), float(ratio.max()))


def test_engine_and_runner_with_the_reference_atari_config():
    """rainbow.Config.set_atari_config() (noisy on) stays on the hand-written engine: learner graph, updates, finite loss, sigmas move."""
    import simple_distributed_rl_amd as srl
    from simple_distributed_rl_amd.algorithms import rainbow
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    cfg = RainbowDeviceConfig(n_envs=8, batch_size=8, memory_capacity=8 * 64, memory_warmup_size=32, target_model_update_interval=4, lr=1e-3, seed=3,
                              enable_noisy_dense=True)
    eng = RainbowEngine(cfg, 0, episode_len=9, overlap=True)
    assert eng.mfma_train and eng.noisy and float(eng.eps.max()) == 0.0
    s0 = eng.q_online.fc1_sigma_w.detach().clone()
    for _ in range(10):
        eng.step(learner_updates=1)
    eng.capture_graphs()
    for _ in range(10):
        eng.step(learner_updates=1)
    eng.join_learner()
    torch.cuda.synchronize()
    info = eng.info()
    assert info["train_count"] >= 10 and np.isfinite(info["loss"])
    assert not torch.equal(eng.q_online.fc1_sigma_w.detach(), s0)  # Adam stepped the sigmas too
    assert len(eng.optimizer.params) == 18

    rl = rainbow.Config()
    rl.set_atari_config()
    assert rl.enable_noisy_dense
    rl.window_length = 4
    rl.memory.capacity, rl.memory.warmup_size = 64 * 64, 512
    runner = srl.Runner(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=20)), rl)
    runner.set_vector_envs(64)
    st = runner.train(max_train_count=20, train_interval=64)
    assert runner.vector_reason == "" and st.train_count >= 20
    sd = runner.parameter.q_online.state_dict()
    assert any(k.endswith("w_sigma") for k in sd)
