"""GPU parity of the matrix-core Q-network forward (srlx_qnet_*) against the torch fp32 modules that mirror
the reference's blocks (and are themselves pinned to the reference by tests/golden/train_step_rainbow.npz).
Tolerance: 1e-5 relative on Q-values (north_star) -- exact-fp32 MFMA, only the summation order differs."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _mk(hw, window, hidden, A, filters=32, seed=0):
    from simple_distributed_rl_amd.device.qnet import EngineQNet

    torch.manual_seed(seed)
    return EngineQNet(A, hw, window, hidden, filters).cuda()


def test_engine_qnet_equals_reference_layout_network():
    """EngineQNet (kernel-friendly parameter layout) == the reference-layout module, and converts back losslessly."""
    from simple_distributed_rl_amd.device.qnet import EngineQNet
    from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

    torch.manual_seed(0)
    ref = atari_qnetwork(6).cuda()
    e = EngineQNet(6).cuda().load_reference_state_dict(ref.state_dict())
    assert e.conv2.weight.is_contiguous(memory_format=torch.channels_last) and e.conv3.weight.is_contiguous(memory_format=torch.channels_last)
    x = torch.rand(5, 4, 84, 84, device="cuda")
    with torch.no_grad():
        _close(e(x), ref(x, channels_first=True))
    sd = e.reference_state_dict()
    assert all(torch.equal(sd[k], v) for k, v in ref.state_dict().items())


def _close(got, want):
    scale = float(want.abs().max())
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("hw,window,hidden,A,B", [((84, 84), 4, 512, 6, 37), ((84, 84), 4, 512, 18, 1024), ((32, 40), 2, 64, 3, 5), ((84, 84), 4, 512, 6, 96), ((84, 84), 4, 512, 6, 1)])
def test_forward_f32_matches_torch(hw, window, hidden, A, B):
    from simple_distributed_rl_amd.device.qnet import QNetInference

    net = _mk(hw, window, hidden, A)
    qn = QNetInference(net, max_batch=max(B, 8))
    x = torch.rand(B, window, hw[0], hw[1], device="cuda")
    with torch.no_grad():
        want = net(x, channels_first=True)
    got = qn.forward_f32(x)
    torch.cuda.synchronize()
    _close(got, want)
    # the kernels read the live parameters: an in-place update is seen without any reload
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.01)
        want2 = net(x, channels_first=True)
    _close(qn.forward_f32(x), want2)


def test_forward_golden_network():
    """Reference-recorded network (state_dict + inputs + outputs of the reference's QNetwork on CPU)."""
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    z = np.load(os.path.join(GOLDEN, "train_step_rainbow.npz"))
    net = EngineQNet(4, (8, 8), 4, 32).cuda()
    net.load_reference_state_dict({k[7:]: torch.tensor(z[k]) for k in z.files if k.startswith("before.")})
    qn = QNetInference(net, max_batch=16)
    obs = torch.tensor(z["obs"][:, 0]).permute(0, 3, 1, 2).contiguous().cuda()  # (B, h, w, 4) -> NCHW
    got = qn.forward_f32(obs)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), z["q_all"], rtol=1e-5, atol=2e-6)


def test_forward_u8_reads_the_ring_directly():
    """conv1 through the frame-offset table == torch on the float32 stack the store would have produced
    (zero history at episode starts, ring wrap-around, gathered n-step items with terminal padding)."""
    import hot_path_oracle as H
    from simple_distributed_rl_amd import _native as N
    from simple_distributed_rl_amd.device.qnet import QNetInference
    from simple_distributed_rl_amd.device.replay import DeviceReplay

    E, W, n, A = 6, 4, 3, 5
    net = _mk((84, 84), W, 64, A, seed=1)
    qn = QNetInference(net, max_batch=64)
    r = DeviceReplay(E, 24, 84 * 84, W, n, A, batch_size=8, warmup_size=1, seed=3)
    o = H.StoreOracle(E, 24, 84 * 84, W, n, A, False, 3)
    rng = np.random.default_rng(0)
    f0 = rng.integers(0, 256, (E, 84 * 84), dtype=np.uint8)
    r.reset_all(torch.tensor(f0).cuda())
    o.reset_all(f0)
    lib = r.lib
    base, fb = N.c_p(), N.c_i64()
    N.check(lib.srlx_store_obs_base(r.h_store, ctypes.byref(base), ctypes.byref(fb)))
    assert fb.value == 84 * 84
    off = torch.zeros((E, W), dtype=torch.int64, device="cuda")
    for step in range(40):
        N.check(lib.srlx_store_frame_table_current(r.h_store, N.tptr(off), None))
        got = qn.forward_u8(base.value, off)
        stack = torch.tensor(o.stack_current()).cuda().view(E, W, 84, 84)
        with torch.no_grad():
            want = net(stack, channels_first=True)
        torch.cuda.synchronize()
        _close(got, want)
        a = rng.integers(0, A, E).astype(np.int32)
        rew = rng.standard_normal(E).astype(np.float32)
        done = (rng.random(E) < 0.15).astype(np.uint8)
        nxt = rng.integers(0, 256, (E, 84 * 84), dtype=np.uint8)
        r.commit(torch.tensor(a).cuda(), torch.tensor(rew).cuda(), torch.tensor(done).cuda(), torch.tensor(done).cuda(), torch.tensor(nxt).cuda())
        o.commit_step(a, rew, done, done, nxt)
    torch.cuda.synchronize()
    # gathered items: states 1..n through the table, state 0 as float32 pixels
    Nn = E * o.item_len
    taus = [(o.pos - 1 - k) % o.item_len for k in range(6)]
    idx = [t * E + e + Nn - 1 for t in taus for e in range(E)]
    idx = [i for i in idx if not (o.flags[o.locate(i)[0], o.locate(i)[1] % o.L] & o.INVALID)][:16]
    B = len(idx)
    t_idx = torch.tensor(idx, dtype=torch.int64).cuda()
    foff = torch.zeros((B, n, W), dtype=torch.int64, device="cuda")
    act = torch.zeros((B, n), dtype=torch.int32, device="cuda")
    rew_t = torch.zeros((B, n), dtype=torch.float32, device="cuda")
    ter = torch.zeros((B, n), dtype=torch.float32, device="cuda")
    N.check(lib.srlx_store_gather_items(r.h_store, B, N.tptr(t_idx), 1, n, N.tptr(foff), N.tptr(act), N.tptr(rew_t), N.tptr(ter), None))
    obs0 = torch.zeros((B, 1, W, 84 * 84), dtype=torch.float32, device="cuda")
    N.check(lib.srlx_store_gather_obs(r.h_store, B, 0, 1, N.tptr(obs0), None))
    got = qn.forward_u8(base.value, foff.view(B * n, W)).clone()
    oo, oa, orw, ot = o.gather_nstep(idx)
    with torch.no_grad():
        want = net(torch.tensor(oo[:, 1:]).cuda().reshape(B * n, W, 84, 84), channels_first=True)
    torch.cuda.synchronize()
    _close(got, want)
    np.testing.assert_array_equal(obs0.cpu().numpy()[:, 0], oo[:, 0])
    np.testing.assert_array_equal(act.cpu().numpy(), oa)
    np.testing.assert_array_equal(rew_t.cpu().numpy(), orw)
    np.testing.assert_array_equal(ter.cpu().numpy(), ot)


@pytest.mark.parametrize("A,dueling,B,stride,hw,hidden", [(6, "average", 32, 1, 84, 512), (18, "", 32, 4, 84, 512), (6, "average", 5, 3, 84, 512),
                                                         (4, "average", 64, 2, 84, 512), (4, "average", 32, 4, 20, 64), (3, "", 7, 1, 36, 32)])
def test_backward_u8_matches_autograd(A, dueling, B, stride, hw, hidden):
    """srlx_qnet_backward_u8 (hand-written backward of every layer, gradients written in the parameters' own memory
    layouts) vs torch autograd on the float32 stack of the same uint8 frames, incl. zero-history frames, a row
    stride (train samples interleaved with no-grad samples in one forward) and replicate-padding borders."""
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    torch.manual_seed(3)
    net = EngineQNet(A, (hw, hw), 4, hidden, 32, dueling).cuda()
    rows = B * stride
    qn = QNetInference(net, max_batch=max(rows, 64)).enable_training(64)
    F, n_frames = hw * hw, 300
    g = torch.Generator(device="cuda").manual_seed(1)
    ring = torch.randint(0, 256, (n_frames * F,), dtype=torch.uint8, device="cuda", generator=g)
    sel = torch.randint(0, n_frames, (rows, 4), device="cuda", generator=g)
    off = sel * F
    off[torch.rand((rows, 4), device="cuda", generator=g) < 0.1] = -1  # zero history
    q = qn.forward_u8(ring.data_ptr(), off).clone()
    frames = ring.view(n_frames, hw, hw)[sel.clamp(min=0)].float() / 255
    frames = torch.where((off < 0)[..., None, None], torch.zeros_like(frames), frames)
    net.zero_grad(set_to_none=True)
    want_q = net(frames, channels_first=True)
    _close(q, want_q.detach())
    grad_q = torch.randn((B, A), device="cuda", generator=g)
    want_q[::stride][:B].backward(grad_q)
    want = [p.grad.detach().clone() for p in qn._params()]
    qn.enable_training(64)  # re-creates zeroed static gradient tensors
    qn.backward_u8(ring.data_ptr(), off, grad_q, sample_stride=stride)
    torch.cuda.synchronize()
    names = ["conv1.w", "conv1.b", "conv2.w", "conv2.b", "conv3.w", "conv3.b", "fc1.w", "fc1.b", "v2.w", "v2.b", "a2.w", "a2.b"]
    for name, p, w in zip(names, qn._params(), want):
        assert p.grad.stride() == p.stride(), name  # the fused Adam walks parameter and gradient with the same strides
        scale = float(w.abs().max()) + 1e-12
        np.testing.assert_allclose(p.grad.cpu().numpy(), w.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale, err_msg=name)


def test_data_gradient_gemm_split_over_k_equals_the_unsplit_one():
    """srlx_qnet_set_dgrad_split(h, 2): conv3's data-gradient GEMM as twice the workgroups over half the K range each (the pad-fold kernel adds the two partial slabs)
    -- the same gradients as unsplit up to float32 association: 1e-6 of each tensor's largest entry (the fast engines run split, the fifteen-launch engine unsplit)."""
    from simple_distributed_rl_amd import _native as N
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    B, F, n_frames = 32, 84 * 84, 200
    g = torch.Generator(device="cuda").manual_seed(5)
    ring = torch.randint(0, 256, (n_frames * F,), dtype=torch.uint8, device="cuda", generator=g)
    off = torch.randint(0, n_frames, (B, 4), device="cuda", generator=g) * F
    grad_q = torch.randn((B, 6), device="cuda", generator=g)
    grads = []
    for split in (1, 2):
        torch.manual_seed(3)
        net = EngineQNet(6).cuda()
        qn = QNetInference(net, max_batch=64).enable_training(64)
        N.check(qn.lib.srlx_qnet_set_dgrad_split(qn.h, split))
        qn.forward_u8(ring.data_ptr(), off)
        qn.backward_u8(ring.data_ptr(), off, grad_q)
        torch.cuda.synchronize()
        grads.append([p.grad.detach().clone() for p in qn._params()])
    moved = 0.0
    for a, b in zip(*grads):
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 1e-6 * scale
        moved += float((a - b).abs().max())
    assert moved > 0  # (the split really ran: the sums associate differently)


def test_the_bf16_pipe_equals_the_float32_pipe(tmp_path):
    """The forward pass evaluates its float32 products on the 16-bit matrix pipe as exact partial products, float32 accumulation: the convolutions of TWO float16
    parts per operand (round 6; conv1: the pixel is one f16, the filter two parts; conv2 / conv3: three of the four products), the first dense layer of three bf16
    parts (six of the nine products); SRLX_CONV_BF16X3=1 = the convolutions of rounds 3-5 (three bf16 parts: 3 / 6 products).  Against the same
    kernels on the float32 pipe (SRLX_CONV1_F32=1 for the convolutions, SRLX_FC1_F32=1 for the dense layer; switches are read once per process, hence
    the subprocesses): Q-values within 2e-6 of max |Q| -- float32 round-off of different summation orders; and the float32-pipe fused kernel stays
    bit-identical to the three-launch path."""
    script = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference\n"
        "torch.manual_seed(0)\n"
        "net = EngineQNet(6).cuda()\n"
        "qn = QNetInference(net, 160)\n"
        "g = torch.Generator(device='cuda').manual_seed(1)\n"
        "F = 84 * 84\n"
        "ring = torch.randint(0, 256, (600 * F,), dtype=torch.uint8, device='cuda', generator=g)\n"
        "off = torch.randint(0, 600, (160, 4), device='cuda', generator=g) * F\n"
        "off[torch.rand((160, 4), device='cuda', generator=g) < 0.1] = -1\n"
        "torch.save(qn.forward_u8(ring.data_ptr(), off).cpu(), sys.argv[1])\n" % ROOT
    )
    outs = {}
    for name, env in (("bf16", {}), ("f32", {"SRLX_CONV1_F32": "1"}), ("three_launches", {"SRLX_NO_FUSED_CONV": "1"}),
                      ("all_f32", {"SRLX_CONV1_F32": "1", "SRLX_FC1_F32": "1"}), ("convs_bf16_only", {"SRLX_FC1_F32": "1"}),
                      ("bf16x3", {"SRLX_CONV_BF16X3": "1"})):  # (round 6: "bf16" = the default = two float16 parts in the convolutions; bf16x3 = rounds 3-5's three bf16 parts)
        path = str(tmp_path / f"q_{name}.pt")
        e = {k: v for k, v in os.environ.items() if k not in ("SRLX_CONV1_F32", "SRLX_CONV23_F32", "SRLX_FC1_F32", "SRLX_NO_FUSED_CONV", "SRLX_CONV_BF16X3")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", script, path], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = torch.load(path)
    assert torch.equal(outs["f32"], outs["three_launches"])
    scale = float(outs["all_f32"].abs().max())
    for a, b in (("bf16", "f32"), ("bf16", "all_f32"), ("convs_bf16_only", "all_f32"), ("f32", "all_f32"), ("bf16x3", "all_f32"), ("bf16x3", "bf16")):
        diff = float((outs[a] - outs[b]).abs().max())
        assert 0 < scale and 0 < diff <= 2e-6 * scale, (a, b, diff, scale)


def test_fc1_operand_planes_path_is_bit_identical():
    """The chip-filling launches' first dense layer on pre-split bf16 operand planes (srlx_fc1_planes.hip: LDS-DMA tiles, no conversions) computes the same
    six partial products in the same order as the staging-split GEMM: Q-values bit for bit, also after the weights changed (refresh_from / a state-dict
    load) -- and a handle whose planes were never refreshed must not use stale ones."""
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    E, F = 1024, 84 * 84
    net, other = _mk((84, 84), 4, 512, 6, seed=0), _mk((84, 84), 4, 512, 6, seed=1)
    g = torch.Generator(device="cuda").manual_seed(1)
    ring = torch.randint(0, 256, (512 * F,), dtype=torch.uint8, device="cuda", generator=g)
    off = torch.randint(0, 512, (E, 4), device="cuda", generator=g) * F
    off[5, :2] = -1  # zero-history frames
    plain = QNetInference(net, E)
    want = plain.forward_u8(ring.data_ptr(), off).clone()
    shared = QNetInference(net, E)
    shared.enable_fc1_planes(private_weights=False)
    assert torch.equal(shared.forward_u8(ring.data_ptr(), off), want)
    with torch.no_grad():  # a learner updates the shared weights in place: the next forward must see them
        net.fc1.weight.mul_(1.5)
        net.conv3.bias.add_(0.01)
    want2 = plain.forward_u8(ring.data_ptr(), off).clone()
    assert not torch.equal(want, want2) and torch.equal(shared.forward_u8(ring.data_ptr(), off), want2)
    # a private actor copy: refreshed from the online network in one pass (float32 copy + planes)
    actor_net = _mk((84, 84), 4, 512, 6, seed=2)
    priv = QNetInference(actor_net, E)
    priv.enable_fc1_planes(private_weights=True)
    priv.refresh_from(net)
    assert all(torch.equal(a, b) for a, b in zip(actor_net.kernel_parameters(), net.kernel_parameters()))
    assert torch.equal(priv.forward_u8(ring.data_ptr(), off), want2)
    actor_net.load_reference_state_dict(other.reference_state_dict())  # weights replaced behind the handle's back: planes are re-split, not reused
    plain_other = QNetInference(other, E)
    assert torch.equal(priv.forward_u8(ring.data_ptr(), off), plain_other.forward_u8(ring.data_ptr(), off))
    for rows in (640, 96):  # 5 row tiles (another K split than 1024 rows); small launches keep the staging-split kernel
        o = off[:rows].contiguous()
        assert torch.equal(priv.forward_u8(ring.data_ptr(), o), plain_other.forward_u8(ring.data_ptr(), o))


def test_adam_inside_the_gradient_launches_equals_the_optimiser_launch():
    """srlx_qnet_fuse_adam_rest (round 5): Adam for the eleven tensors besides the first dense layer's weight inside the launches that finish their gradients
    (convolution tensors: the gradient reductions' epilogues; small vectors: the packing launch of srlx_qnet_publish) against srlx_adam_step over the same
    gradients -- parameters and both moment estimates bit-equal after three steps (model_torch.py:71,109: torch.optim.Adam, one step per train()); and a backward
    pass whose optimiser step was never completed by a publish must make the next one fail loudly."""
    from simple_distributed_rl_amd import _native as N
    from simple_distributed_rl_amd.device.qnet import DeviceAdam, EngineQNet, QNetInference

    def build():
        torch.manual_seed(11)
        net = EngineQNet(6, (84, 84), 4, 512, 32, "average").cuda()
        qn = QNetInference(net, max_batch=128).enable_training(32)
        opt = DeviceAdam(qn._params(), lr=2.5e-4)
        steps = torch.zeros(1, dtype=torch.int64, device="cuda")
        opt.fuse_first_dense(qn, steps)
        return net, qn, opt, steps

    F, n_frames, B, stride = 84 * 84, 200, 32, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    ring = torch.randint(0, 256, (n_frames * F,), dtype=torch.uint8, device="cuda", generator=g)
    off = torch.randint(0, n_frames, (B * stride, 4), device="cuda", generator=g) * F
    grads = [torch.randn((B, 6), device="cuda", generator=g) for _ in range(3)]
    (net_a, qa, opt_a, steps_a), (net_b, qb, opt_b, steps_b) = build(), build()
    opt_b.fuse_rest(qb)
    w0 = [p_.detach().clone() for p_ in net_b.parameters()]
    for k in range(3):
        qa.forward_u8(ring.data_ptr(), off)
        qa.backward_u8(ring.data_ptr(), off, grads[k], sample_stride=stride)
        opt_a.step(steps_a)
        qa.publish_to(None, 0, bump=steps_a)
        qb.forward_u8(ring.data_ptr(), off)
        qb.backward_u8(ring.data_ptr(), off, grads[k], sample_stride=stride)
        opt_b.step(steps_b)  # (launches nothing)
        qb.publish_to(None, 0, bump=steps_b)
    torch.cuda.synchronize()
    assert int(steps_a.item()) == int(steps_b.item()) == 3
    for (name, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        assert torch.equal(pa, pb), name
    for k, (m1, m2, v1, v2) in enumerate(zip(opt_a.exp_avg, opt_b.exp_avg, opt_a.exp_avg_sq, opt_b.exp_avg_sq)):
        assert torch.equal(m1, m2) and torch.equal(v1, v2), k
    assert all(float((p_.detach() - q_).abs().max()) > 0 for p_, q_ in zip(net_b.parameters(), w0))  # (every tensor did take its steps)
    qb.forward_u8(ring.data_ptr(), off)
    qb.backward_u8(ring.data_ptr(), off, grads[0], sample_stride=stride)
    with pytest.raises(Exception, match="never completed"):
        qb.backward_u8(ring.data_ptr(), off, grads[0], sample_stride=stride)
    qb.publish_to(None, 0, bump=steps_b)
    torch.cuda.synchronize()


def test_two_part_float16_split_error_and_its_range_flag():
    """Round 6: the fused convolution kernel's products are three exact products of two float16 parts per operand (x = hi + lo / 2048).  (a) Against a float64
    evaluation of the same network the Q-values are within 1e-6 of max |Q| (measured 2.9e-7: float32 accumulation round-off, the same as the three-part bf16 split
    and the float32 pipe, tools/conv_split_error.py) -- also with filters 8 x larger and 20 x smaller than the initialisation's.  (b) An activation above 65 504 does
    not fit float16: the kernel then sets a bit of the handle's range word and `check_ranges` (called by every engine's info()) raises -- never a silent wrong value."""
    import copy

    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference, check_ranges

    B, F = 96, 84 * 84
    g = torch.Generator(device="cuda").manual_seed(1)
    ring = torch.randint(0, 256, (300 * F,), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randint(0, 300, (B, 4), device="cuda", generator=g)
    frames = ring.view(300, 84, 84).cpu()[idx.cpu()].double() / 255.0
    for scale in (1.0, 8.0, 0.05):
        torch.manual_seed(0)
        net = EngineQNet(6).cuda()
        with torch.no_grad():
            for name, p in net.named_parameters():
                if "conv" in name and p.dim() == 4:
                    p.mul_(scale)
        qn = QNetInference(net, B)
        q = qn.forward_u8(ring.data_ptr(), idx * F).double().cpu()
        with torch.no_grad():
            q64 = copy.deepcopy(net).double().cpu()(frames)
        assert float((q - q64).abs().max()) <= 1e-6 * float(q64.abs().max()), scale
        check_ranges()
        del qn
    # (b) conv1 filters large enough that act1 passes 65 504
    torch.manual_seed(0)
    net = EngineQNet(6).cuda()
    with torch.no_grad():
        net.conv1.weight.fill_(600.0)  # 256 inputs x ~0.5 x 600 = 76 800
    qn = QNetInference(net, B)
    qn.forward_u8(ring.data_ptr(), idx * F)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="65504"):
        check_ranges()
    del qn
    import gc

    gc.collect()
    check_ranges()  # the flagged handle is gone: nothing to report
    # conv3's output feeds the first dense layer as two-part float16 operand planes (chip-filling launches): the same flag, bit 2
    torch.manual_seed(0)
    net = EngineQNet(6).cuda()
    with torch.no_grad():
        net.conv3.weight.fill_(1.0e4)
    qn = QNetInference(net, 512)
    qn.enable_fc1_planes(private_weights=True)
    idx = torch.randint(0, 300, (512, 4), device="cuda", generator=g)
    qn.forward_u8(ring.data_ptr(), idx * F)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="conv3"):
        check_ranges()
    del qn
    gc.collect()
    check_ranges()
