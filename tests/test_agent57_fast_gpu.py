"""Round 6: Agent57_light with every network pass, optimiser step and array operation in libsrlx (device/agent57_fast.py; csrc/srlx_agent57.hip; the UVFA /
hidden-layer modes of the srlx_qnet handle).  Yardsticks: float64 torch on the CPU for the handles, torch autograd + torch.optim.Adam for the tails, the reference's
recorded Trainer.train() (tests/golden/train_step_agent57_light.npz, made by oracle/gen_golden_agent57.py importing the reference) for the tails and the TD
arithmetic on reference numbers, and the round-5 learner with torch tails (device/agent57_light.py:Agent57LightLearner) for one whole update at 84 x 84."""
import copy
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _env():
    import torch

    from simple_distributed_rl_amd import _native as N

    return N, N.lib(), torch, torch.device("cuda:0")


def _frames(torch, rows, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    base = torch.randint(0, 256, (rows * 4, 84 * 84), dtype=torch.uint8, device="cuda", generator=g)
    off = (torch.arange(rows * 4, device="cuda", dtype=torch.int64) * (84 * 84)).view(rows, 4).contiguous()
    stack = (base.view(rows, 4, 84, 84).double() / 255.0).cpu()
    return base, off, stack


def _randomise(torch, module, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            fan = p[0].numel() if p.dim() > 1 else p.numel()
            p.copy_((torch.randn(p.shape, generator=g) * (scale / max(fan, 1) ** 0.5)).to(p.device))
    return module


def _close(torch, got, want, rel=1e-5, what=""):
    want = want.double().cpu()
    torch.testing.assert_close(got.double().cpu(), want, rtol=rel, atol=rel * float(want.abs().max()) + 1e-12, msg=lambda m: f"{what}: {m}")


def test_uvfa_q_network_handle_against_float64():
    """A Q-network with UVFA columns (agent57_light/model_torch.py:35-64) on a srlx_qnet handle: Q rows and every gradient -- the UVFA columns' included -- against
    the same network evaluated in float64 on the CPU with the inputs concatenated as the reference does (1e-5 of the largest element)."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    A, Na, B, Hh = 5, 6, 8, 64
    X = 1 + 1 + A + Na
    layout = (0, 1, 2, A, 2 + A, Na)
    net = _randomise(torch, EngineQNet(A, hidden=Hh, uvfa_cols=X).to(dev), 3, scale=1.4)
    net.fix_formats()
    inf = QNetInference(net, 2 * B, 0, uvfa_layout=layout)
    inf.enable_training(B)
    base, off, stack = _frames(torch, 2 * B, 1)
    g = torch.Generator().manual_seed(5)
    r_ext, r_int = torch.randn(2 * B, generator=g), torch.rand(2 * B, generator=g)
    act, actor = torch.randint(0, A, (2 * B,), generator=g), torch.randint(0, Na, (2 * B,), generator=g)
    dev_in = [r_ext.cuda(), r_int.cuda(), act.int().cuda(), actor.int().cuda()]
    inf.set_uvfa_inputs(*dev_in)
    q = inf.forward_u8(base.data_ptr(), off).clone()
    extras = torch.cat([r_ext.view(-1, 1), r_int.view(-1, 1), torch.eye(A)[act], torch.eye(Na)[actor]], dim=1).double()
    ref = copy.deepcopy(net).double().cpu()
    want = ref(stack, extras=extras)
    _close(torch, q, want.detach(), what="q")
    G = torch.randn((B, A), generator=g)
    (want[0::2] * G.double()).sum().backward()
    inf.backward_u8(base.data_ptr(), off, G.cuda().contiguous(), sample_stride=2)
    torch.cuda.synchronize()
    names = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias", "fc1.weight", "fc1.bias", "v2.weight", "v2.bias", "a2.weight", "a2.bias", "fcx"]
    rp = dict(ref.named_parameters())
    for name, p in zip(names, inf._params()):
        _close(torch, p.grad, rp[name].grad, rel=2e-5, what=name)


def test_hidden_layer_handle_with_layernorm_against_float64():
    """head_mode 1 (the embedding / lifelong networks' trunk + one dense layer, model_torch.py:70-117): output with and without the LayerNorm, and the gradients
    for a gradient arriving at the first `units` post-ReLU units of the padded layer (32 of 128)."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.qnet import EngineHiddenNet, QNetInference

    B, U = 8, 32
    net = EngineHiddenNet(U, tail_shapes=[(U,), (U,)]).to(dev)
    ref_sd = {}
    g = torch.Generator().manual_seed(9)
    for k, shape in (("in_block.image_block.image_layers.0", (32, 4, 8, 8)), ("in_block.image_block.image_layers.2", (64, 32, 4, 4)), ("in_block.image_block.image_layers.4", (64, 64, 3, 3))):
        ref_sd[k + ".weight"] = torch.randn(shape, generator=g) * (1.4 / (shape[1] * shape[2] * shape[3]) ** 0.5)
        ref_sd[k + ".bias"] = torch.randn(shape[0], generator=g) * 0.05
    ref_sd["dense.weight"] = torch.randn((U, net.flat), generator=g) * (1.4 / net.flat ** 0.5)
    ref_sd["dense.bias"] = torch.randn(U, generator=g) * 0.05
    ref_sd["ln.weight"], ref_sd["ln.bias"] = 1 + 0.1 * torch.randn(U, generator=g), 0.1 * torch.randn(U, generator=g)
    net.load_reference(ref_sd, "dense", ("ln.weight", "ln.bias"))
    back = net.reference_tensors("dense", ("ln.weight", "ln.bias"))
    for k, v in ref_sd.items():
        torch.testing.assert_close(back[k].cpu(), v, rtol=0, atol=0)  # the layout conversion round-trips exactly
    assert net.units_padded == 128 and float(net.fc1.weight[U:].abs().max()) == 0.0
    inf = QNetInference(net, B, 0)
    inf.enable_training(B)
    inf.set_head_mode(1, U)
    base, off, stack = _frames(torch, B, 2)
    out = torch.zeros((B, U), device=dev)
    inf.forward_u8(base.data_ptr(), off, out=out)
    ref = copy.deepcopy(net).double().cpu()
    want = ref(stack)
    _close(torch, out, want.detach(), what="hidden")
    G = torch.randn((B, U), generator=g)
    (want * G.double()).sum().backward()
    inf.backward_u8(base.data_ptr(), off, G.cuda().contiguous(), sample_stride=1)
    torch.cuda.synchronize()
    rp = dict(ref.named_parameters())
    for name, p in zip(["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias", "fc1.weight", "fc1.bias"], inf._params()):
        _close(torch, p.grad, rp[name].grad, rel=2e-5, what=name)
    # an inference handle with the LayerNorm fused (the lifelong networks' last layer, :112,116)
    # (LayerNorm over ALL units of the handle: a padded layer would normalise over its zero units too -- the engines only fuse it where units == units_padded)
    net128 = EngineHiddenNet(128, tail_shapes=[(128,), (128,)]).to(dev)
    sd128 = dict(ref_sd)
    sd128["dense.weight"] = torch.randn((128, net.flat), generator=g) * (1.4 / net.flat ** 0.5)
    sd128["dense.bias"] = torch.randn(128, generator=g) * 0.05
    sd128["ln.weight"], sd128["ln.bias"] = 1 + 0.1 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
    net128.load_reference(sd128, "dense", ("ln.weight", "ln.bias"))
    inf3 = QNetInference(net128, B, 0)
    inf3.set_head_mode(1, 128, net128.tail[0], net128.tail[1])
    out3 = torch.zeros((B, 128), device=dev)
    inf3.forward_u8(base.data_ptr(), off, out=out3)
    r128 = copy.deepcopy(net128).double().cpu()
    want3 = torch.nn.functional.layer_norm(r128(stack), (128,), r128.tail[0], r128.tail[1], 1e-5)
    _close(torch, out3, want3.detach(), what="layernorm")


def _emb_tail_torch(torch, emb, actions, params, A):
    w1, b1, lw, lb, w2, b2 = params
    x = torch.cat([emb[0::2], emb[1::2]], dim=1)
    h = torch.relu(x @ w1.t() + b1)
    y = torch.nn.functional.layer_norm(h, (h.shape[1],), lw, lb, 1e-5)
    p = torch.softmax(y @ w2.t() + b2, dim=1)
    return torch.nn.functional.mse_loss(p, torch.eye(A, dtype=p.dtype)[actions.long()])


def _run_emb_tail(N, lib, torch, emb, actions, params, lr, steps_taken, adam=True):
    dev = torch.device("cuda:0")
    B2, D = emb.shape
    Hd, A = params[0].shape[0], params[4].shape[0]
    ps = [p.detach().clone().float().to(dev).contiguous() for p in params]
    gs, ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    tab = lambda ts: ctypes.cast((N.c_p * 6)(*[t.data_ptr() for t in ts]), N.c_p)  # noqa: E731
    loss, gemb = torch.zeros(1, device=dev), torch.zeros((B2, D), device=dev)
    step = torch.tensor([steps_taken], dtype=torch.int64, device=dev)
    e, a = emb.float().to(dev).contiguous(), actions.int().to(dev).contiguous()
    N.check(lib.srlx_agent57_emb_tail(B2 // 2, D, Hd, A, N.tptr(e), N.tptr(a), tab(ps), tab(gs), tab(ms) if adam else None, tab(vs) if adam else None, 1e-5, lr, 0.9, 0.999, 1e-8,
                                      N.tptr(step) if adam else None, N.tptr(loss), N.tptr(gemb), None))
    torch.cuda.synchronize()
    return float(loss.item()), gemb.cpu(), [g.cpu() for g in gs], [p.cpu() for p in ps]


def test_embedding_tail_and_rnd_tail_against_autograd_and_adam():
    N, lib, torch, dev = _env()
    g = torch.Generator().manual_seed(21)
    B, D, Hd, A = 16, 32, 128, 6
    emb = torch.relu(torch.randn((2 * B, D), generator=g))
    actions = torch.randint(0, A, (B,), generator=g)
    params = [torch.randn((Hd, 2 * D), generator=g) * 0.2, torch.randn(Hd, generator=g) * 0.1, 1 + 0.1 * torch.randn(Hd, generator=g), 0.1 * torch.randn(Hd, generator=g),
              torch.randn((A, Hd), generator=g) * 0.2, torch.randn(A, generator=g) * 0.1]
    e64 = emb.double().requires_grad_(True)
    p64 = [p.double().requires_grad_(True) for p in params]
    want = _emb_tail_torch(torch, e64, actions, p64, A)
    want.backward()
    loss, gemb, gs, _ = _run_emb_tail(N, lib, torch, emb, actions, params, 5e-4, 0, adam=False)
    assert abs(loss - float(want)) <= 1e-5 * abs(float(want))
    _close(torch, gemb, e64.grad, rel=2e-5, what="d emb")
    for k, (got, p) in enumerate(zip(gs, p64)):
        _close(torch, got, p.grad, rel=2e-5, what=f"tail gradient {k}")
    # with Adam: three steps; torch.optim.Adam is fed the KERNEL's own gradients (the gradients were held to autograd above; an optimiser fed autograd's would
    # differ by the full learning rate wherever a gradient is a rounding residue: Adam's first steps move a weight by lr * sign(g))
    tp = [p.clone().requires_grad_(True) for p in params]
    opt = torch.optim.Adam(tp, lr=5e-4)
    ps = [p.detach().clone().float().cuda().contiguous() for p in params]
    ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    gsd = [torch.zeros_like(p) for p in ps]
    tab = lambda ts: ctypes.cast((N.c_p * 6)(*[t.data_ptr() for t in ts]), N.c_p)  # noqa: E731
    for step in range(3):
        loss_d, gemb_d = torch.zeros(1, device=dev), torch.zeros((2 * B, D), device=dev)
        st = torch.tensor([step], dtype=torch.int64, device=dev)
        N.check(lib.srlx_agent57_emb_tail(B, D, Hd, A, N.tptr(emb.cuda().contiguous()), N.tptr(actions.int().cuda()), tab(ps), tab(gsd), tab(ms), tab(vs), 1e-5, 5e-4, 0.9, 0.999,
                                          1e-8, N.tptr(st), N.tptr(loss_d), N.tptr(gemb_d), None))
        torch.cuda.synchronize()
        for t_, g_ in zip(tp, gsd):
            t_.grad = g_.cpu().clone()
        opt.step()
        for k, (got, t_) in enumerate(zip(ps, tp)):
            torch.testing.assert_close(got.cpu(), t_.detach(), rtol=2e-6, atol=1e-8, msg=lambda m: f"step {step} tensor {k}: {m}")
    # RND tail: LayerNorm + MSE against the target, rows every second row of a [2B] pass
    Dr = 128
    h2 = torch.relu(torch.randn((2 * B, Dr), generator=g))
    t2 = torch.randn((2 * B, Dr), generator=g)
    lw, lb = 1 + 0.1 * torch.randn(Dr, generator=g), 0.1 * torch.randn(Dr, generator=g)
    h64, lw64, lb64 = h2[0::2].double().requires_grad_(True), lw.double().requires_grad_(True), lb.double().requires_grad_(True)
    want = torch.nn.functional.mse_loss(t2[0::2].double(), torch.nn.functional.layer_norm(h64, (Dr,), lw64, lb64, 1e-5))
    want.backward()
    dl = [t.float().cuda().contiguous() for t in (h2, t2, lw, lb)]
    gw, gb, loss_d, gh = torch.zeros(Dr, device=dev), torch.zeros(Dr, device=dev), torch.zeros(1, device=dev), torch.zeros((B, Dr), device=dev)
    N.check(lib.srlx_agent57_rnd_tail(B, Dr, 2 * Dr, N.tptr(dl[0]), N.tptr(dl[1]), N.tptr(dl[2]), N.tptr(dl[3]), N.tptr(gw), N.tptr(gb), None, None, None, None, None, None, 1e-5,
                                      5e-4, 0.9, 0.999, 1e-8, None, N.tptr(loss_d), N.tptr(gh), None))
    torch.cuda.synchronize()
    assert abs(float(loss_d.item()) - float(want)) <= 1e-5 * abs(float(want))
    _close(torch, gh, h64.grad, rel=2e-5, what="d hidden")
    _close(torch, gw, lw64.grad, rel=2e-5, what="d ln weight")
    _close(torch, gb, lb64.grad, rel=2e-5, what="d ln bias")


def test_tails_and_td_arithmetic_on_the_reference_trainer_step():
    """The reference's recorded Agent57_light Trainer.train() (tests/golden/train_step_agent57_light.npz: made by oracle/gen_golden_agent57.py from the imported
    reference; 8 x 8 float states, so the trunks are evaluated by torch here) through the NEW arithmetic: the embedding tail's loss and Adam step, the RND tail's, and
    the fused TD prologue with the per-actor discount and the signed TD errors -> the mixed priorities.  rel 1e-5 on losses / TD errors / priorities."""
    N, lib, torch, dev = _env()
    from test_agent57_gpu import _load_plugin

    z = np.load(os.path.join(GOLDEN, "train_step_agent57_light.npz"))
    runner, param, trainer, nets = _load_plugin(z)
    from simple_distributed_rl_amd.rl import functions as funcs

    B, A, Na = len(z["actions"]), int(z["n_actions"]), int(z["actor_num"])
    t = lambda k, dt=None: torch.tensor(z[k] if dt is None else z[k].astype(dt)).to(dev)  # noqa: E731
    states, n_states = t("states"), t("n_states")
    with torch.no_grad():
        emb = torch.stack([nets["emb"].predict(states), nets["emb"].predict(n_states)], dim=1).reshape(2 * B, -1).contiguous()  # rows 2 b = f(s), 2 b + 1 = f(s')
        lt = nets["lifelong_target"](states)
        lp_hidden = nets["lifelong_train"].hidden_block(nets["lifelong_train"].in_block(states))
    sd = nets["emb"].state_dict()
    keys = ("out_block.hidden_layers.0.weight", "out_block.hidden_layers.0.bias", "out_block_normalize.weight", "out_block_normalize.bias", "out_block_out1.weight",
            "out_block_out1.bias")
    loss, _, _, after = _run_emb_tail(N, lib, torch, emb.cpu(), t("actions").cpu(), [sd[k].cpu() for k in keys], float(z["episodic_lr"]), 0)
    np.testing.assert_allclose(loss, float(z["emb_loss"]), rtol=1e-5)
    for k, got in zip(keys, after):
        np.testing.assert_allclose(got.numpy(), z["after.emb." + k], rtol=1e-5, atol=4e-5, err_msg=k)  # (Adam's first step: see test_agent57_gpu.py)
        assert np.mean(np.abs(got.numpy() - z["after.emb." + k]) > 5e-6) < 2e-2, k
    sl = nets["lifelong_train"].state_dict()
    lw, lb = sl["hidden_normalize.weight"].clone().contiguous(), sl["hidden_normalize.bias"].clone().contiguous()
    m = [torch.zeros_like(lw) for _ in range(4)]
    loss_d, gh = torch.zeros(1, device=dev), torch.zeros((B, lw.numel()), device=dev)
    st = torch.zeros(1, dtype=torch.int64, device=dev)
    N.check(lib.srlx_agent57_rnd_tail(B, lw.numel(), lw.numel(), N.tptr(lp_hidden.contiguous()), N.tptr(lt.contiguous()), N.tptr(lw), N.tptr(lb), None, None, N.tptr(m[0]), N.tptr(m[1]),
                                      N.tptr(m[2]), N.tptr(m[3]), None, None, 1e-5, float(z["lifelong_lr"]), 0.9, 0.999, 1e-8, N.tptr(st), N.tptr(loss_d), N.tptr(gh), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss_d.item()), float(z["lifelong_loss"]), rtol=1e-5)
    np.testing.assert_allclose(lw.cpu().numpy(), z["after.lifelong_train.hidden_normalize.weight"], rtol=1e-5, atol=4e-5)
    np.testing.assert_allclose(lb.cpu().numpy(), z["after.lifelong_train.hidden_normalize.bias"], rtol=1e-5, atol=4e-5)
    # ---- TD: the stand-alone fused kernel cannot take the extras; the backward entry point can -- exercised end to end in the 84 x 84 test below.  Here: the
    # reference's numbers through srlx_dqn_target-free arithmetic = the n = 1 prologue, via srlx_qnet_backward_td_u8's sibling srlx_nstep_td_huber_priority is
    # covered by tests/test_agent57_gpu.py; the signed errors -> priorities mix:
    disc = np.array(funcs.create_discount_list(Na), np.float32)[z["actor_idx"]]
    beta = torch.tensor(np.array(funcs.create_beta_list(Na), np.float32)).to(dev)
    pri = torch.zeros(B, device=dev)
    te, ti, ai = t("td_ext"), t("td_int"), t("actor_idx", np.int32)  # (kept alive: the launch is asynchronous)
    N.check(lib.srlx_agent57_priority(B, A, N.tptr(te), None, N.tptr(ti), None, None, N.tptr(ai), N.tptr(beta), None, None, N.tptr(pri), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pri.cpu().numpy(), z["priorities"], rtol=1e-5, atol=1e-7)
    assert disc.shape == (B,)


def _cfg84(batch=16, E=16, capacity=None, warmup=64, hidden=64, actor_num=4):
    import simple_distributed_rl_amd as srl
    from simple_distributed_rl_amd.algorithms import agent57_light

    cfg = agent57_light.Config(batch_size=batch, actor_num=actor_num, target_model_update_interval=5, episodic_memory_capacity=64, ucb_window_size=6)
    cfg.window_length = 4
    cfg.memory.capacity, cfg.memory.warmup_size = capacity or E * 40, warmup
    cfg.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1000)
    cfg.input_block.image.set_dqn_block()
    cfg.hidden_block.set_dueling_network((hidden,))
    env = srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(84, 84), n_actions=4, episode_len=11)))
    cfg.setup(env)
    return cfg


def test_one_update_equals_the_round5_learner_with_torch_tails():
    """One whole update (two Q-networks, embedding, RND, priorities) of the all-libsrlx engine against the round-5 learner (torch dense tails, torch.optim.Adam,
    torch.cat UVFA inputs: device/agent57_light.py:Agent57LightLearner) on the same weights and the same sampled batch, at 84 x 84: the four losses, the signed TD
    errors and the priorities to 1e-5; the exported parameters after the step against the torch learner's (Adam's first step: lr * sign for most weights)."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine
    from simple_distributed_rl_amd.device.agent57_light import Agent57LightLearner

    torch.manual_seed(4)
    cfg = _cfg84()
    eng = Agent57LightFastEngine(cfg, 16, 0, episode_len=11, seed=3)
    assert not eng.overlap
    for _ in range(12):
        eng.step(learner_updates=0)
    ref_param = copy.deepcopy(eng.parameter)
    ref_param.to_device(dev)
    eng._learner_body(None)
    torch.cuda.synchronize()
    rp, B, W = eng.replay, eng.replay.B, 4
    b = rp.batch
    obs = torch.zeros((B, 2, W, 84 * 84), dtype=torch.float32, device=dev)
    act, rew, term = torch.zeros((B, 1), dtype=torch.int32, device=dev), torch.zeros((B, 1), device=dev), torch.zeros((B, 1), device=dev)
    N.check(lib.srlx_store_gather_nstep(rp.h_store, B, N.tptr(b.indices), N.tptr(obs), N.tptr(act), N.tptr(rew), N.tptr(term), None))
    torch.cuda.synchronize()
    s, e = eng.loc_slot, eng.loc_env
    learner = Agent57LightLearner(cfg, ref_param, dev, channels_first=True)
    stack = obs.view(B, 2, W, 84, 84)
    with torch.backends.cudnn.flags(enabled=True, benchmark=False):
        pri = learner.update_networks(stack[:, 0], stack[:, 1], act.view(-1), rew.view(-1), eng.x_r_int[s, e], 1.0 - term.view(-1), eng.x_prev_action[s, e].long(),
                                      eng.x_prev_r_ext[s, e], eng.x_prev_r_int[s, e], eng.x_actor[s, e].long(), b.weights)
    torch.cuda.synchronize()
    want = learner.losses()
    got = eng.losses()
    for k in ("ext_loss", "int_loss", "emb_loss", "lifelong_loss"):
        assert abs(got[k] - want[k]) <= 1e-5 * max(abs(want[k]), 1e-2), (k, got[k], want[k])
    np.testing.assert_allclose(eng.out["q_ext"]["td"].cpu().numpy(), learner.td_ext.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(eng.out["q_int"]["td"].cpu().numpy(), learner.td_int.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(eng.priorities.cpu().numpy(), pri.cpu().numpy(), rtol=1e-5, atol=2e-6)
    out = eng.export_parameter(copy.deepcopy(eng.parameter))
    for name in ("q_ext_online", "q_int_online", "emb_network", "lifelong_train", "lifelong_target"):
        a, c = dict(getattr(out, name).state_dict()), dict(getattr(ref_param, name).state_dict())
        for k in a:
            x, y = a[k].float().cpu().numpy(), c[k].float().cpu().numpy()
            np.testing.assert_allclose(x, y, rtol=1e-5, atol=4e-5 * (cfg.episodic_lr / 1e-4 if "emb" in name or "lifelong" in name else 1.0), err_msg=f"{name}.{k}")
            assert np.mean(np.abs(x - y) > 5e-6 * (5 if "emb" in name or "lifelong" in name else 1)) < 3e-2, (name, k)


def test_forwards_beside_the_previous_backward_change_no_bit():
    """The update with every network's forward passes (and the dense tails) on the side stream beside the previous network's backward pass (`hoist_forwards`, the
    default) against the same update with one network after the other: the same launches on the same data in another stream order -- every parameter, every
    optimiser moment, losses and priorities bit for bit, over several updates."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine

    def run(hoist):
        torch.manual_seed(4)
        eng = Agent57LightFastEngine(_cfg84(), 16, 0, episode_len=11, seed=3)
        eng.hoist_forwards = hoist
        for _ in range(12):
            eng.step(learner_updates=0)
        for _ in range(3):
            eng._learner_body(None)
            eng._after_update()
        torch.cuda.synchronize()
        return eng

    a, b = run(True), run(False)
    assert a.losses() == b.losses()
    assert torch.equal(a.priorities, b.priorities)
    for n1, n2 in zip(a.nets.values(), b.nets.values()):
        for p, q in zip(n1.module.parameters(), n2.module.parameters()):
            assert torch.equal(p, q), n1.name
        for x, y in zip(n1.opt.exp_avg + n1.opt.exp_avg_sq, n2.opt.exp_avg + n2.opt.exp_avg_sq):
            assert torch.equal(x, y), n1.name


def test_small_engine_trains_items_consistent_and_reproducible():
    """The engine without overlap (16 lanes): 30 lock-steps with updates; the per-slot item fields are what the lanes held when they acted; finite losses; two
    instances from one seed walk one trajectory bit for bit; evaluation mode uses arm 0 / test_beta."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine

    def run():
        torch.manual_seed(0)
        cfg = _cfg84()
        eng = Agent57LightFastEngine(cfg, 16, 0, episode_len=11, seed=3)
        hist = []
        for t in range(30):
            slot = eng.replay._steps_committed % eng.L
            pa, arm, reset = eng.prev_action.clone(), eng.arm().clone(), eng.reset_lane.clone()
            eng.step(learner_updates=1)
            hist.append((slot, pa.cpu().numpy(), arm.cpu().numpy(), eng.actions.cpu().numpy().copy(), reset.cpu().numpy(), eng.x_r_int[slot].cpu().numpy().copy(),
                         eng.env.done.cpu().numpy().copy()))
        torch.cuda.synchronize()
        return eng, cfg, hist

    eng, cfg, hist = run()
    info = eng.info()
    assert eng.train_count > 10 and all(np.isfinite(info[k]) for k in ("ext_loss", "int_loss", "emb_loss", "lifelong_loss")), info
    for t in range(1, 30):
        slot, pa, arm, act, reset, r_int, done = hist[t]
        _, _, arm_prev, act_prev, reset_prev, _, done_prev = hist[t - 1]
        np.testing.assert_array_equal(eng.x_prev_action[slot].cpu().numpy(), pa)
        np.testing.assert_array_equal(eng.x_actor[slot].cpu().numpy(), arm)
        keep = (reset_prev == 0) & (done_prev == 0)
        np.testing.assert_array_equal(pa[keep], act_prev[keep])
        np.testing.assert_array_equal(arm[done_prev == 0], arm_prev[done_prev == 0])
        assert (r_int[reset == 1] == 0).all() and (r_int[reset == 0] >= 0).all()
        if (reset == 0).any():
            assert (r_int[reset == 0] > 0).any()
    assert int(eng.ucb.arm.min()) >= 0 and int(eng.ucb.arm.max()) < cfg.actor_num
    eng2, _, hist2 = run()
    for a, b in zip(hist, hist2):
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
    assert eng.info() == eng2.info()
    eng.training = False
    q_ext, q_int, q = eng.policy_q()
    torch.testing.assert_close(q, q_ext + cfg.test_beta * q_int)


@pytest.mark.parametrize("graphs", [False, True])
def test_overlapped_engine_on_published_sets(graphs):
    """512 lanes: the update beside the actors on published parameter sets (captured graphs or eager).  After every join the set the actors read equals the master
    parameters (packed first dense layer planes = the float32 weight split into three bf16 parts; UVFA columns; the RND LayerNorm mirror), the run trains, and
    two instances walk one trajectory -- the second with the five image blocks of a lock-step as ONE launch (`multi_trunk`: the same kernel body per sample, so the
    same bits)."""
    N, lib, torch, dev = _env()
    from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine

    def run(multi_trunk=False):
        torch.manual_seed(1)
        cfg = _cfg84(batch=16, E=512, capacity=512 * 12, warmup=512 * 4)
        eng = Agent57LightFastEngine(cfg, 512, 0, episode_len=11, seed=5)
        eng.multi_trunk = multi_trunk
        assert eng.overlap and eng.sets and eng.actor_stream is not None  # (the actors on the engine's low-priority stream)
        for k in range(14):
            if k == 8 and graphs:
                eng.capture_graphs()
            eng.step(learner_updates=1)
        eng.join_learner()
        torch.cuda.synchronize()
        return eng

    eng = run()
    info = eng.info()
    assert eng.train_count >= 8 and all(np.isfinite(info[k]) for k in ("ext_loss", "int_loss", "emb_loss", "lifelong_loss")), info
    # what the actors read now is the master: evaluate the policy pass on the set and on the master (deselect) -- bit-identical Q rows
    eng._q_ready = False  # (the Q rows cached for the next policy step were evaluated on the set BEFORE the last flip)
    q_set = [t.clone() for t in eng.policy_q()]
    for n in eng.nets.values():
        n.actor.select_set(-1)
        n.actor.weights_changed()
    rn = eng.nets["rnd"]
    torch.testing.assert_close(rn.ln_sets[eng._set][0], rn.module.tail[0], rtol=0, atol=0)
    torch.testing.assert_close(rn.ln_sets[eng._set][1], rn.module.tail[1], rtol=0, atol=0)
    q_master = eng.policy_q()
    for a, b in zip(q_set, q_master):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    eng2 = run(multi_trunk=True)
    assert eng2.info() == info
    for n1, n2 in zip(eng.nets.values(), eng2.nets.values()):
        for p, q in zip(n1.module.parameters(), n2.module.parameters()):
            assert torch.equal(p, q), n1.name
    eng2.close()  # (in reverse order of construction: each hands the thread back to the stream it found)
    eng.close()


def _golden84_engine(z, fused_adam):
    """An engine whose five networks carry the golden's recipe weights and whose buffers hold the golden's batch: item b's five frames are written into lane b of
    the ring through the store's own commits (positions 0..4), the frame tables point at them."""
    N, lib, torch, dev = _env()
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import simple_distributed_rl_amd as srl
    from gen_golden_agent57_84 import NETS, recipe_networks
    from simple_distributed_rl_amd.algorithms import agent57_light
    from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine
    from simple_distributed_rl_amd.rl import functions as funcs

    B, A, Na, E = len(z["actions"]), int(z["n_actions"]), int(z["actor_num"]), 16
    cfg = agent57_light.Config(batch_size=B, actor_num=Na, target_model_update_interval=5, lr_ext=float(z["lr_ext"]), lr_int=float(z["lr_int"]),
                               episodic_lr=float(z["episodic_lr"]), lifelong_lr=float(z["lifelong_lr"]))
    cfg.window_length = 4
    cfg.memory.capacity, cfg.memory.warmup_size = E * 40, 8
    cfg.input_block.image.set_dqn_block()
    cfg.hidden_block.set_dueling_network((512,))
    env = srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(84, 84), n_actions=A, episode_len=50)))
    cfg.setup(env)
    param = cfg.make_parameter()
    ks = {n: [(str(k), tuple(int(x) for x in str(s).strip("()").split(",") if x.strip())) for k, s in zip(z["keys." + n], z["shapes." + n])] for n in NETS}
    weights = recipe_networks(ks)
    mods = dict(q_ext=param.q_ext_online, q_int=param.q_int_online, q_ext_target=param.q_ext_target, q_int_target=param.q_int_target, emb=param.emb_network,
                lifelong_target=param.lifelong_target, lifelong_train=param.lifelong_train)
    for n, m in mods.items():
        assert [k for k, _ in ks[n]] == list(m.state_dict().keys()), n  # the plugin's modules carry the reference's keys
        m.load_state_dict({k: torch.tensor(v) for k, v in weights[n].items()})
    eng = Agent57LightFastEngine(cfg, E, 0, episode_len=50, seed=1, parameter=param, fused_adam=fused_adam)
    r = eng.replay
    frames = torch.tensor(z["frames"]).to(dev)  # [B][5][84][84]
    F = 84 * 84
    lane = torch.zeros((E, 5, F), dtype=torch.uint8, device=dev)
    lane[:B] = frames.view(B, 5, F)
    r.reset_all(lane[:, 0].contiguous())
    zi, zf, zb = torch.zeros(E, dtype=torch.int32, device=dev), torch.zeros(E, device=dev), torch.zeros(E, dtype=torch.uint8, device=dev)
    for k in range(1, 5):
        r.commit(zi, zf, zb, zb, lane[:, k].contiguous())
    L = r.L
    off = torch.zeros((B, 2, 4), dtype=torch.int64)
    for b in range(B):
        for s in range(2):
            for c in range(4):
                off[b, s, c] = ((b * L) + s + c) * F
    r.frame_off_all.copy_(off.to(dev))
    r.frame_off_next.copy_(off[:, 1:2].contiguous().to(dev))
    t = lambda k, dt: torch.tensor(z[k].astype(dt)).to(dev)  # noqa: E731
    bt = r.batch
    bt.actions.copy_(t("actions", np.int32).view(B, 1))
    bt.rewards.copy_(t("rewards_ext", np.float32).view(B, 1))
    bt.terminated.copy_((1.0 - t("dones", np.float32)).view(B, 1))
    bt.weights.copy_(t("weights", np.float32))
    bt.indices.copy_(torch.arange(B, device=dev) + (r.capacity - 1))
    actor = t("actor_idx", np.int32)
    eng.on_r_ext[0::2], eng.on_r_ext[1::2] = t("prev_rewards_ext", np.float32), t("rewards_ext", np.float32)
    eng.on_r_int[0::2], eng.on_r_int[1::2] = t("prev_rewards_int", np.float32), t("rewards_int", np.float32)
    eng.on_action[0::2], eng.on_action[1::2] = t("prev_actions", np.int32), t("actions", np.int32)
    eng.on_actor[0::2], eng.on_actor[1::2] = actor, actor
    eng.tg_r_ext.copy_(t("rewards_ext", np.float32)), eng.tg_r_int.copy_(t("rewards_int", np.float32)), eng.tg_action.copy_(t("actions", np.int32)), eng.tg_actor.copy_(actor)
    eng.b_discount.copy_(torch.tensor(np.array(funcs.create_discount_list(Na), np.float32)[z["actor_idx"]]).to(dev))
    eng.b_r_int.copy_(t("rewards_int", np.float32))
    return eng, weights, ks


def _engine_tensor(eng, net, key):
    """The engine-side tensor (value, gradient) of a reference state_dict key, in the REFERENCE's layout."""
    import torch

    from simple_distributed_rl_amd.device.agent57_fast import _EMB_TAIL_KEYS, _RND_TAIL_KEYS

    n = eng.nets[{"q_ext": "q_ext", "q_int": "q_int", "emb": "emb", "lifelong_train": "rnd"}[net]]
    m = n.module
    conv = {"in_block.image_block.image_layers.0": m.conv1, "in_block.image_block.image_layers.2": m.conv2, "in_block.image_block.image_layers.4": m.conv3}
    base, leaf = key.rsplit(".", 1)
    if base in conv:
        p = getattr(conv[base], leaf)
        return p.detach().contiguous(), p.grad.contiguous()
    C, P = m.out_c, m.out_p
    if net in ("q_ext", "q_int"):
        H = m.hidden
        hd = "hidden_block.hidden_layers.0."
        stream, layer = key[len(hd):].split(".")[0], key[len(hd):].split(".")[1]
        lo = 0 if stream == "v_layers" else H
        if layer == "0" and leaf == "weight":
            def conv_w(w, x):
                w = w[lo:lo + H].reshape(H, P, C).permute(0, 2, 1).reshape(H, C * P)
                return torch.cat([w, x[:, lo:lo + H].t()], dim=1)
            return conv_w(m.fc1.weight.detach(), m.fcx.detach()), (conv_w(m.fc1.weight.grad, m.fcx.grad) if m.fc1.weight.grad is not None else None)
        if layer == "0":
            return m.fc1.bias.detach()[lo:lo + H], m.fc1.bias.grad[lo:lo + H]
        p = getattr(m.v2 if stream == "v_layers" else m.a2, leaf)
        return p.detach(), p.grad
    dense = "emb_block.hidden_layers.0" if net == "emb" else "hidden_block.hidden_layers.0"
    U = m.units
    if base == dense:
        if leaf == "weight":
            cv = lambda w: w[:U].reshape(U, P, C).permute(0, 2, 1).reshape(U, C * P)  # noqa: E731
            return cv(m.fc1.weight.detach()), cv(m.fc1.weight.grad)
        return m.fc1.bias.detach()[:U], m.fc1.bias.grad[:U]
    keys = _EMB_TAIL_KEYS if net == "emb" else _RND_TAIL_KEYS
    i = keys.index(key)
    return m.tail[i].detach(), n.tail_g[i]


@pytest.mark.parametrize("fused", [False, True])
def test_update_on_the_reference_trainer_step_at_84(fused):
    """One whole update of the all-libsrlx engine against the reference's recorded Agent57_light `Trainer.train()` at the benchmark geometry
    (tests/golden/train_step_agent57_light84.npz, oracle/gen_golden_agent57_84.py: the imported reference on CPU torch): the four losses, the signed TD errors and
    the mixed priorities to rel 1e-5; with the optimiser steps as launches of their own (fused=False) 2048 sampled entries of EVERY parameter gradient to 1e-5 of
    the tensor's largest gradient entry + rel 1e-4 (float32 sums of ~10^4 terms in another order); the Adam steps of both variants: lr * sign(g) for most weights --
    within 2 % of lr except where a gradient is a rounding residue (<= 3 % of the sampled entries)."""
    N, lib, torch, dev = _env()
    z = np.load(os.path.join(GOLDEN, "train_step_agent57_light84.npz"))
    eng, before, ks = _golden84_engine(z, fused_adam=fused)
    eng._learner_body(None, drawn=True)
    torch.cuda.synchronize()
    got = eng.losses()
    for k in ("ext_loss", "int_loss", "emb_loss", "lifelong_loss"):
        np.testing.assert_allclose(got[k], float(z[k]), rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(eng.out["q_ext"]["td"].cpu().numpy(), z["td_ext"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(eng.out["q_int"]["td"].cpu().numpy(), z["td_int"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(eng.priorities.cpu().numpy(), z["priorities"], rtol=1e-5, atol=2e-6)
    lrs = dict(q_ext=float(z["lr_ext"]), q_int=float(z["lr_int"]), emb=float(z["episodic_lr"]), lifelong_train=float(z["lifelong_lr"]))
    for net in ("q_ext", "q_int", "emb", "lifelong_train"):
        for key, _ in ks[net]:
            pos = z[f"pos.{net}.{key}"]
            val, grad = _engine_tensor(eng, net, key)
            if not fused and grad is not None:
                g = grad.reshape(-1)[torch.tensor(pos).to(grad.device)].float().cpu().numpy()
                np.testing.assert_allclose(g, z[f"grad.{net}.{key}"], rtol=1e-4, atol=1e-5 * float(z[f"gmax.{net}.{key}"]), err_msg=f"grad {net}.{key}")
            upd = val.reshape(-1)[torch.tensor(pos).to(val.device)].double().cpu().numpy() - before[net][key].reshape(-1)[pos].astype(np.float64)
            want = z[f"upd.{net}.{key}"].astype(np.float64)
            off = np.abs(upd - want) > 0.02 * lrs[net]
            assert off.mean() <= 0.03, (net, key, float(off.mean()))
            assert np.abs(upd - want).max() <= 2.001 * lrs[net], (net, key)


def test_runner_train_and_train_mp_reach_the_fast_engine():
    """`srl.Runner(env, agent57_light.Config()).train()` / `.train_mp()` with 84 x 84 x 4 frames run on the all-libsrlx engine (device/vector_runner.py picks it by
    `why_not_fast`) and on `DistributedAgent57Light` (2 ranks time-sharing the test GPU over gloo); the trained networks come back in the Runner's own Parameter."""
    N, lib, torch, dev = _env()
    import simple_distributed_rl_amd as srl
    from simple_distributed_rl_amd.algorithms import agent57_light
    from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine

    cfg = agent57_light.Config(batch_size=8, actor_num=4, target_model_update_interval=5, episodic_memory_capacity=64, ucb_window_size=6)
    cfg.window_length = 4
    cfg.memory.capacity, cfg.memory.warmup_size = 2 * 8 * 30, 32
    cfg.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1000)
    cfg.hidden_block.set_dueling_network((64,))
    runner = srl.Runner(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(84, 84), n_actions=3, episode_len=7)), cfg)
    runner.set_vector_envs(8)
    before = {k: v.detach().clone() for k, v in runner.parameter.q_ext_online.state_dict().items()}
    st = runner.train(max_train_count=10, train_interval=8)
    assert runner.vector_reason == "" and st.end_reason == "max_train_count over." and st.train_count >= 10
    assert st.episode_count > 0 and st.memory.length() > 32
    after = runner.parameter.q_ext_online.state_dict()
    assert any(not torch.equal(before[k], after[k].to(before[k].device)) for k in before)  # the Runner's own parameter object carries the trained networks
    st = runner.train_mp(actor_num=2, actor_devices=["cuda:0", "cuda:0"], max_train_count=12, timeout=300, sync_interval_steps=4)
    assert runner.vector_reason == "" and st.end_reason == "max_train_count over." and st.train_count >= 12 and st.trainer_recv_q > 0
    assert isinstance(Agent57LightFastEngine, type)
