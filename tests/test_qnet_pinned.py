"""The Q-network at the BENCHMARK geometry (84 x 84 x 4 frames, 6 actions, dueling 512) against the reference's own QNetwork
(tests/golden/qnet84_{init,wide}.npz, recorded by oracle/gen_golden_qnet84.py from srl/algorithms/rainbow/model_torch.py:15-29 on CPU torch).
The 8.0 M weights are regenerated from the generator's numpy recipe (bit for bit), only inputs and Q-values are stored.

CPU part: the torch module that mirrors the reference's blocks reproduces the recorded Q-values (so every GPU test that uses the mirror as its
yardstick is anchored to the reference at this geometry too, not only at the 8 x 8 toy of train_step_rainbow.npz).
GPU part: the hand-written forward (split-bf16 matrix pipe, through the float32 entry AND through the uint8 ring) against the recording:
  * rtol 1e-5 (north_star) on every Q-value; the only absolute slack is twice the reference's own float32 uncertainty (|f32 - f64| of ITS evaluation);
  * against the reference evaluated in FLOAT64: the kernels' error is at most 2x the error of the reference's own float32 evaluation
    (+ one ulp of max |Q|).  float32 products are evaluated as 6 of their 9 bf16 partial products; the three dropped ones are <= 2^-24 |ab| each,
    i.e. of the size of float32's own rounding -- this is the test that says so on weights spanning 2^-6 .. 2^6 with sign-alternating runs
    (`wide`), not only on an initialisation."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
KINDS = ("init", "wide")


def _golden(kind):
    import ast

    from gen_golden_qnet84 import recipe_state_dict

    z = np.load(os.path.join(GOLDEN, f"qnet84_{kind}.npz"))
    keys_shapes = [(str(k), ast.literal_eval(str(s))) for k, s in zip(z["keys"], z["shapes"])]
    sd = {k: torch.tensor(v) for k, v in recipe_state_dict(keys_shapes, kind, int(z["seed"])).items()}
    return z, sd


@pytest.mark.parametrize("kind", KINDS)
def test_the_mirror_module_reproduces_the_reference_network(kind):
    from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

    z, sd = _golden(kind)
    net = atari_qnetwork(6)
    assert list(net.state_dict().keys()) == list(sd.keys()), "the mirror keeps the reference's parameter names and order"
    net.load_state_dict(sd)
    x = torch.tensor(z["frames"].astype(np.float32) / 255)  # (B, H, W, C) like the reference's input
    with torch.no_grad():
        q = net(x).numpy()
    np.testing.assert_allclose(q, z["q_ref_f32"], rtol=2e-6, atol=2e-6 * np.abs(z["q_ref_f32"]).max())


def _check(q, z, label):
    """rtol 1e-5 against the reference's float32 Q-values, with the reference's OWN float32 uncertainty (its distance to its float64 evaluation) as the
    only absolute slack -- on the `wide` set Q-values are residues of sums a thousand times larger, and two correct float32 implementations differ by
    that much (the float32 matrix pipe does: 2e-5 relative on |Q| > 1e-3 max |Q|; tools/qnet_accuracy.py prints the table); and against float64: at most
    twice the reference's float32 error + one ulp of max |Q|.  Measured (tools/qnet_accuracy.py, `wide`): split-bf16 pipe 9.0e-7 max / 3.1e-7 rms of
    max |Q|, float32 pipe 9.1e-7 / 2.8e-7, the reference's float32 evaluation 9.8e-7 max."""
    ref32, ref64 = z["q_ref_f32"].astype(np.float64), z["q_ref_f64"]
    q = q.astype(np.float64)
    scale = np.abs(ref64).max()
    err_ref = np.abs(ref32 - ref64).max()
    excess = np.abs(q - ref32) - 1e-5 * np.abs(ref32)
    assert excess.max() <= 2.0 * err_ref, f"{label}: {excess.max():.3e} beyond rtol 1e-5, reference uncertainty {err_ref:.3e}"
    err = np.abs(q - ref64).max()
    assert err <= 2.0 * err_ref + 2.0 ** -23 * scale, f"{label}: error vs float64 {err:.3e}, the reference's float32 evaluation {err_ref:.3e}"
    return err, err_ref


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_device_forward_against_the_reference_network(kind):
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    z, sd = _golden(kind)
    net = EngineQNet(6).cuda().load_reference_state_dict(sd)
    B = z["frames"].shape[0]
    qn = QNetInference(net, max_batch=1024)
    x = torch.tensor(z["frames"].astype(np.float32) / 255).permute(0, 3, 1, 2).contiguous().cuda()
    q_f32_entry = qn.forward_f32(x).cpu().numpy()
    _check(q_f32_entry, z, "forward_f32")
    # through the uint8 ring: channel c of sample b is frame 4 b + c; the two zero-history channels of the last sample as "no frame" (-1)
    ring = torch.tensor(z["frames"]).permute(0, 3, 1, 2).contiguous().view(B * 4, 84 * 84).cuda()
    off = (torch.arange(B * 4, device="cuda", dtype=torch.int64) * (84 * 84)).view(B, 4).clone()
    off[5, :2] = -1
    q_u8 = qn.forward_u8(ring.data_ptr(), off).cpu().numpy()
    _check(q_u8, z, "forward_u8")
    # the chip-filling launch (1024 rows: other tile shapes, operand planes for the first dense layer): the six samples tiled
    rep = (1024 + B - 1) // B
    off_big = off.repeat(rep, 1)[:1024].contiguous()
    qn.enable_fc1_planes(private_weights=True)
    qn.weights_changed()
    q_big = qn.forward_u8(ring.data_ptr(), off_big).cpu().numpy()
    for r in range(0, 1024 - B + 1, B * 37):
        _check(q_big[r:r + B], z, f"forward_u8 rows {r}..")
    assert np.array_equal(q_big[:B], q_big[B * 5:B * 6]), "the same sample gives the same bits wherever it sits in the batch"


@pytest.mark.gpu
def test_split_products_on_a_trained_network_against_float64():
    """Weights that are no longer an initialisation: the engine trains for 300 updates on synthetic frames, then the hand-written forward is
    compared with the SAME weights evaluated in float64 by torch on the CPU: error <= 2x that of torch's own float32 evaluation (+ 1 ulp)."""
    from simple_distributed_rl_amd.device.qnet import QNetInference
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine
    from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

    cfg = RainbowDeviceConfig(n_envs=64, batch_size=32, memory_capacity=64 * 128, memory_warmup_size=256, target_model_update_interval=50, lr=1e-3)
    eng = RainbowEngine(cfg, 0, episode_len=50)
    for _ in range(320):
        eng.step(learner_updates=1)
    torch.cuda.synchronize()
    assert eng.info()["train_count"] >= 250
    sd = {k: v.detach().cpu() for k, v in eng.q_online.reference_state_dict().items()}
    mirror = atari_qnetwork(6)
    mirror.load_state_dict(sd)
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (8, 84, 84, 4), dtype=np.uint8)
    x = torch.tensor(frames.astype(np.float32) / 255)
    with torch.no_grad():
        q32 = mirror(x).numpy().astype(np.float64)
        q64 = mirror.double()(x.double()).numpy()
    qn = QNetInference(eng.q_online, max_batch=8)
    ring = torch.tensor(frames).permute(0, 3, 1, 2).contiguous().view(32, 84 * 84).cuda()  # through the uint8 ring: the fused split-bf16 convolutions
    off = (torch.arange(32, device="cuda", dtype=torch.int64) * (84 * 84)).view(8, 4).contiguous()
    q = qn.forward_u8(ring.data_ptr(), off).cpu().numpy().astype(np.float64)
    scale = np.abs(q64).max()
    err, err_ref = np.abs(q - q64).max(), np.abs(q32 - q64).max()
    assert err <= 2.0 * err_ref + 2.0 ** -23 * scale, (err, err_ref, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("neighbour", [4, 0])
def test_the_shipped_policy_pass_against_the_reference_network(kind, neighbour):
    """The combination bench.py and the actor ranks run (round-4 lock-step): a PUBLISHED parameter set (packed filters, first dense layer as bf16 operand planes, small
    vectors: srlx_qnet_publish) read by srlx_qnet_forward_u8_policy -- fused convolutions writing operand planes, the planes GEMM as half-CU workgroups with 4 K splits
    (`neighbour` 4: beside a learner) or CU-filling workgroups (0: an actor rank), the head kernel that also selects the actions -- against the reference's recorded
    Q-values, directly (not through the equivalence with the round-3 path)."""
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    z, sd = _golden(kind)
    net = EngineQNet(6).cuda().load_reference_state_dict(sd)
    B, E = z["frames"].shape[0], 1024
    actor = QNetInference(net, max_batch=E)
    actor.enable_fc1_planes(private_weights=True)
    actor.enable_actor_sets()
    actor.set_fc1_neighbour(neighbour)
    source = QNetInference(net, max_batch=64)
    source.weights_changed()
    source.publish_to(actor, 1, with_fc1=True)
    actor.select_set(1)
    ring = torch.tensor(z["frames"]).permute(0, 3, 1, 2).contiguous().view(B * 4, 84 * 84).cuda()
    off = (torch.arange(B * 4, device="cuda", dtype=torch.int64) * (84 * 84)).view(B, 4).clone()
    off[5, :2] = -1
    off_big = off.repeat((E + B - 1) // B, 1)[:E].contiguous()
    eps = torch.zeros(E, dtype=torch.float32, device="cuda")  # greedy: the selected action is the first argmax of the row
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    actions = torch.full((E,), -1, dtype=torch.int32, device="cuda")
    q = actor.forward_u8_policy(ring.data_ptr(), off_big, eps, 1234, counter, actions).cpu().numpy()
    for r in range(0, E - B + 1, B * 29):
        _check(q[r:r + B], z, f"forward_u8_policy rows {r}.. ({'half-CU' if neighbour else 'CU-filling'} planes kernel)")
    assert np.array_equal(actions.cpu().numpy(), q.argmax(axis=1).astype(np.int32))
    assert np.array_equal(q[:B], q[B * 7:B * 8]), "the same sample gives the same bits wherever it sits in the batch"


def _golden84_engine(fast: bool, schedule=None):
    """A RainbowEngine carrying the golden's recipe weights whose next `_learner_body` trains on the golden's 16 items (7 frames each, written into the engine's ring
    by ordinary commits), with the golden's importance weights."""
    import ast

    from gen_golden_qnet84 import recipe_state_dict
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    z = np.load(os.path.join(GOLDEN, "train_step_rainbow84.npz"))
    keys_shapes = [(str(k), ast.literal_eval(str(s))) for k, s in zip(z["keys"], z["shapes"])]
    sd_on = {k: torch.tensor(v) for k, v in recipe_state_dict(keys_shapes, "init", int(z["seed_online"])).items()}
    sd_tg = {k: torch.tensor(v) for k, v in recipe_state_dict(keys_shapes, "init", int(z["seed_target"])).items()}
    frames, actions, reward, done = z["frames"], z["actions"], z["reward"], z["done"]
    B, n, E = frames.shape[0], 3, 512
    cfg = RainbowDeviceConfig(n_envs=E, batch_size=B, memory_capacity=E * 8, memory_warmup_size=E, lr=float(z["lr"]), discount=float(z["discount"]),
                              target_model_update_interval=1000, enable_reward_clip=False)
    if schedule is not None:
        cfg.schedule = schedule
    eng = RainbowEngine(cfg, 0, episode_len=1000, overlap=fast, fast=fast)
    assert eng.fast == fast
    eng.q_online.load_reference_state_dict(sd_on)
    eng.q_target.load_reference_state_dict(sd_tg)
    rp = eng.replay
    dev = eng.dev

    def lanes(x, dtype, fill=0):  # the 16 items are environments 0..15; the other lanes idle on black frames
        t = torch.full((E,) + tuple(x.shape[1:]), fill, dtype=dtype, device=dev)
        t[:B] = torch.tensor(x).to(dev).to(dtype)
        return t

    rp.reset_all(lanes(frames[:, 0].reshape(B, -1), torch.uint8))
    for i in range(1, n + 4):  # frame i arrives with the transition (frame i - 1 -> i); the item's transitions are commits 4, 5, 6
        k = i - 4
        a = lanes(actions[:, k] if k >= 0 else np.zeros(B, np.int32), torch.int32)
        r = lanes(reward[:, k] if k >= 0 else np.zeros(B, np.float32), torch.float32)
        d = lanes(done[:, k] if k >= 0 else np.zeros(B, np.float32), torch.uint8)
        rp.commit(a, r, d, d, lanes(frames[:, i].reshape(B, -1), torch.uint8))
    # the item of environment e that starts at ring time 3 was completed -- and its leaf appended -- by commit number 3 + n - 1 = 5 (leaf slot = commit * E + e)
    idx = ((3 + n - 1) * E + torch.arange(B, device=dev, dtype=torch.int64)) + rp.capacity - 1
    w = torch.tensor(z["weights"]).to(dev)

    def fixed_batch(*a, **k):
        rp.batch.indices.copy_(idx)
        rp.batch.weights.copy_(w)
        return rp.gather_drawn(all_states=True)

    rp.sample_items = fixed_batch
    if fast:
        eng._check_versions()  # (the loaded weights: packed filters and published set follow)
    else:
        eng.inf_online.weights_changed()
        eng.inf_target.weights_changed()
    return eng, z, keys_shapes


@pytest.mark.gpu
def test_fast_engine_learner_step_against_the_reference_trainer():
    """One update of `RainbowEngine(fast=True)` -- the shipped learner path: fused draw + gather tables, split-bf16 forward of s_0..s_n and the target pass, TD / Huber /
    priorities in the head kernel of the hand-written backward, Adam fused into the launches that finish each gradient (writing the next published set) -- against ONE
    Trainer.train() of the reference at the benchmark geometry (tests/golden/train_step_rainbow84.npz, oracle/gen_golden_train84.py): the same weights (numpy recipe),
    the same 16 items, the same importance weights.  Target, Q, loss and priorities to rel 1e-5; the Adam step: lr * sign(g) for most weights, i.e. this test pins the
    SIGN of 2048 gradient entries per tensor (to 1 % of a step, 2 % of the entries exempt) and a sum -- the gradient MAGNITUDES are pinned on the reference's own
    `p.grad` by test_learner_gradients_against_the_reference_trainer below."""
    eng, z, keys_shapes = _golden84_engine(True)
    B, n = z["frames"].shape[0], 3
    before = {k: v.clone() for k, v in eng.q_online.reference_state_dict().items()}
    eng._learner_body(publish=1)
    torch.cuda.synchronize()
    q0 = eng.inf_online.q[: B * (n + 1)].view(B, n + 1, -1)[:, 0].cpu().numpy()
    np.testing.assert_allclose(q0, z["q0"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(eng.target.cpu().numpy(), z["target_q"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(float(eng.loss.item()), float(z["loss"]), rtol=1e-5)
    # |target - q| is a difference of O(1) numbers: 1e-5 of their size is the bar for the residue
    np.testing.assert_allclose(eng.priorities.cpu().numpy(), z["priorities"], rtol=1e-5, atol=1e-5 * float(np.abs(z["target_q"]).max()))
    after = eng.q_online.reference_state_dict()
    lr = float(z["lr"])
    for k, _ in keys_shapes:
        pos = torch.tensor(z["pos." + k])
        got = (after[k].double().cpu().reshape(-1)[pos] - before[k].double().cpu().reshape(-1)[pos]).numpy()
        want = z["upd." + k].astype(np.float64)
        # Adam's first step moves a weight by lr * g / (|g| + eps): compare the UPDATE; where |g| ~ eps (1e-8) the direction amplifies last-ulp differences of the
        # gradient, so up to 2 % of the sampled entries may sit anywhere within one lr-sized step -- the rest agree to 1 % of a step
        bad = np.abs(got - want) > 1e-2 * lr
        assert bad.mean() <= 0.02, (k, float(bad.mean()))
        assert np.abs(got - want).max() <= 2.0 * lr * (1 + 1e-3), (k, float(np.abs(got - want).max()))
        total = float((after[k].double() - before[k].double()).sum().item())
        assert abs(total - float(z["sum." + k])) <= 2e-2 * float(z["abs." + k]) + 1e-12, (k, total, float(z["sum." + k]))


@pytest.mark.gpu
def test_learner_gradients_against_the_reference_trainer():
    """The hand-written backward pass at 84 x 84 against the reference's own gradients: 2048 sampled entries of EVERY `p.grad` that `loss.backward()` left in the
    reference's Trainer.train() (`grad.<key>` of the golden, caught at `optimizer.step()`: model_torch.py:107-109), with the optimiser as a launch of its own
    (EngineSchedule(fused_adam=False): the same kernels write the gradients out instead of consuming them in their epilogues).  Bar: rel 1e-5 of the tensor's largest gradient
    entry + rel 1e-4 per entry -- a gradient entry is a float32 sum of up to 16 x 441 products whose partial sums cancel (MIOpen-free CPU torch on the reference's
    side, ticketed MFMA partial sums here): the documented cancellation slack."""
    from simple_distributed_rl_amd.device.rainbow import EngineSchedule

    eng, z, keys_shapes = _golden84_engine(False, EngineSchedule(fused_adam=False))
    assert not eng.fast and eng.mfma_train
    eng._learner_body()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(eng.loss.item()), float(z["loss"]), rtol=1e-5)
    net = eng.q_online
    H = net.hidden
    conv = {"in_block.image_block.image_layers.0": net.conv1, "in_block.image_block.image_layers.2": net.conv2, "in_block.image_block.image_layers.4": net.conv3}
    wv, wa = net._split_fc1(net.fc1.weight.grad)
    hd = "hidden_block.hidden_layers.0."
    got = {hd + "v_layers.0.weight": wv, hd + "adv_layers.0.weight": wa, hd + "v_layers.0.bias": net.fc1.bias.grad[:H], hd + "adv_layers.0.bias": net.fc1.bias.grad[H:],
           hd + "v_layers.2.weight": net.v2.weight.grad, hd + "v_layers.2.bias": net.v2.bias.grad, hd + "adv_layers.2.weight": net.a2.weight.grad,
           hd + "adv_layers.2.bias": net.a2.bias.grad}
    for ref, c in conv.items():
        got[ref + ".weight"], got[ref + ".bias"] = c.weight.grad.contiguous(), c.bias.grad
    for k, _ in keys_shapes:
        pos = torch.tensor(z["pos." + k])
        g = got[k].detach().float().cpu().reshape(-1)[pos].numpy()
        np.testing.assert_allclose(g, z["grad." + k], rtol=1e-4, atol=1e-5 * float(z["gmax." + k]), err_msg=k)
        total = float(got[k].double().sum().item())
        assert abs(total - float(z["gsum." + k])) <= 1e-4 * float(np.abs(z["grad." + k]).sum() / len(pos) * got[k].numel()) + 1e-9, k
