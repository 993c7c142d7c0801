"""PPO parity pins.

(1) Reference outputs: tests/golden/ppo_v_step_{discrete,continuous}.npz were recorded by oracle/gen_golden_ppo.py from real
    `Trainer.train()` steps of the reference's importable torch PPO (srl/algorithms/ppo_v/torch_model.py:111-178) and its
    torch distributions.  The oracle's `ppo_loss` (clipped surrogate + entropy, the arithmetic ppo.py:126-137,152,166-167
    shares with ppo_v) and `normal_logprob` must reproduce what the reference reported; the srlx_ppo_* kernels must too.
(2) Known answers, derived by hand below, for the two pieces no importable reference code executes: the GAE recursion of
    srl/algorithms/ppo/ppo.py:389-404 (incl. its "no bootstrap on the last stored step, even when truncated" rule) and the
    value-clip branch of ppo.py:155-157.
"""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hot_path_oracle as H  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _steps(name):
    z = np.load(os.path.join(GOLDEN, f"ppo_v_step_{name}.npz"))
    for k in range(int(z["n_steps"])):
        yield z, {key[len(f"s{k}_"):]: z[key] for key in z.files if key.startswith(f"s{k}_")}


def _advantage(z, s):
    """ppo_v torch_model.py:150-151: q = r + not_terminated * discount * n_v; adv = (q - v).detach()"""
    f = np.float32
    q = s["reward"] + s["not_terminated"] * f(z["discount"]) * s["n_v"]
    return (q - s["v"]).astype(f)[:, 0]


@pytest.mark.parametrize("name", ["discrete", "continuous"])
def test_oracle_surrogate_and_entropy_match_reference_steps(name):
    for z, s in _steps(name):
        adv = _advantage(z, s)
        pol, _, ent = H.ppo_loss(s["new_logpi"], s["old_logpi"], adv, s["v"][:, 0], s["v"][:, 0], s["v"][:, 0], False, True, float(z["clip_range"]),
                                 False, 0.0, 1.0, 1.0)
        np.testing.assert_allclose(pol, s["loss_policy"], rtol=2e-6, atol=1e-7)  # :161-163 == ppo.py:128-137,152
        np.testing.assert_allclose(ent, s["loss_e"], rtol=2e-6, atol=1e-7)  # :168-169 == ppo.py:166-167 (before the weight)


def test_oracle_normal_logprob_matches_reference_distribution():
    """srl/rl/torch_/distributions/normal_dist_block.py:56-57 -> srl/rl/functions.py:232-238 (the torch twin of the TF block ppo.py uses)."""
    for z, s in _steps("continuous"):
        got = H.normal_logprob(s["action"], s["loc"], s["log_scale"])
        np.testing.assert_allclose(got, s["plain_logprob"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(s["new_logpi"], s["plain_logprob"], rtol=0, atol=0)  # the recorded run used the plain Normal policy


def test_gae_known_answer():
    """ppo.py:389-404 on a 3-step episode, discount g = 0.5, gae_discount l = 0.5 (g*l = 0.25):
         r = [1, 0, 2]   v = V(s_t) = [0.5, 0.25, 1.0]   n_v = V(s_{t+1}) = [0.25, 1.0, 7.0]
       i=2 (last stored step):  delta = r2 - v2            = 2 - 1            = 1        gae = 1
                                (n_v[2] = 7 is NOT used: no bootstrap on the last step, terminated or truncated alike)
       i=1:                     delta = r1 + g n_v1 - v1   = 0 + 0.5 - 0.25   = 0.25     gae = 0.25 + 0.25 * 1   = 0.5
       i=0:                     delta = r0 + g n_v0 - v0   = 1 + 0.125 - 0.5  = 0.625    gae = 0.625 + 0.25 * 0.5 = 0.75
       Two such episodes back to back in one [T=6] column (done flags at t = 2 and t = 5) give the same numbers twice; a column
       whose last step is not flagged (rollout cut) and that passes last_values bootstraps from them instead:
         i=2 with last_values = 7: delta = 2 + 0.5*7 - 1 = 4.5, gae = 4.5; i=1: 0.25 + 0.25*4.5 = 1.375; i=0: 0.625 + 0.25*1.375 = 0.96875."""
    r = np.array([[1], [0], [2]], np.float32)
    v = np.array([[0.5], [0.25], [1.0]], np.float32)
    done = np.array([[0], [0], [1]], np.uint8)
    np.testing.assert_array_equal(H.gae(r, v, done, None, 0.5, 0.5)[:, 0], np.array([0.75, 0.5, 1.0], np.float32))
    np.testing.assert_array_equal(H.gae(r, v, done, np.array([7.0], np.float32), 0.5, 0.5)[:, 0], np.array([0.75, 0.5, 1.0], np.float32))
    r2, v2 = np.concatenate([r, r]), np.concatenate([v, v])
    d2 = np.array([[0], [0], [1], [0], [0], [1]], np.uint8)
    np.testing.assert_array_equal(H.gae(r2, v2, d2, None, 0.5, 0.5)[:, 0], np.array([0.75, 0.5, 1.0] * 2, np.float32))
    cut = np.zeros((3, 1), np.uint8)
    np.testing.assert_array_equal(H.gae(r, v, cut, np.array([7.0], np.float32), 0.5, 0.5)[:, 0], np.array([0.96875, 1.375, 4.5], np.float32))


def test_value_clip_known_answer():
    """ppo.py:155-161 with value_clip_range c = 0.2, value_loss_weight 1:
         v = [1.0, 0.0]  old_v = [0.5, 0.1]  v_target = [2.0, 0.05]
         sample 0: v_clipped = clip(1.0, 0.3, 0.7) = 0.7; max((1-2)^2, (0.7-2)^2) = max(1, 1.69) = 1.69
         sample 1: v_clipped = clip(0.0, -0.1, 0.3) = 0.0; max(0.0025, 0.0025) = 0.0025
         value_loss = mean = 0.84625;  without the clip: mean(1, 0.0025) = 0.50125"""
    lp = np.zeros((2, 1), np.float32)
    a = np.zeros(2, np.float32)
    v, ov, vt = np.array([1.0, 0.0], np.float32), np.array([0.5, 0.1], np.float32), np.array([2.0, 0.05], np.float32)
    _, val, _ = H.ppo_loss(lp, lp, a, v, vt, ov, False, True, 0.2, True, 0.2, 1.0, 0.0)
    np.testing.assert_allclose(val, 0.84625, rtol=1e-6)
    _, val, _ = H.ppo_loss(lp, lp, a, v, vt, ov, False, True, 0.2, False, 0.2, 1.0, 0.0)
    np.testing.assert_allclose(val, 0.50125, rtol=1e-6)


# ---- the kernels on the same pins ------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["discrete", "continuous"])
def test_ppo_loss_kernel_matches_reference_steps(name):
    import torch

    from simple_distributed_rl_amd import _native as N

    lib, dev = N.lib(), torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)  # noqa: E731
    for z, s in _steps(name):
        B, K = s["new_logpi"].shape
        adv = _advantage(z, s)
        # device tensors held in locals: a temporary would be freed (and its memory reused) before the kernel runs
        lp_t, olp_t, adv_t, v_t = t(s["new_logpi"]), t(s["old_logpi"]), t(adv), t(s["v"][:, 0])
        losses, g_lp, g_v = torch.zeros(3, device=dev), torch.empty((B, K), device=dev), torch.empty(B, device=dev)
        N.check(lib.srlx_ppo_loss_logpi(B, K, N.tptr(lp_t), N.tptr(olp_t), N.tptr(adv_t), N.tptr(v_t), N.tptr(v_t), N.tptr(v_t), 0, 1,
                                        float(z["clip_range"]), 0, 0.0, 1.0, 1.0, N.tptr(losses), N.tptr(g_lp), N.tptr(g_v), None))
        torch.cuda.synchronize()
        got = losses.cpu().numpy()
        np.testing.assert_allclose(got[0], s["loss_policy"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(got[2], s["loss_e"], rtol=1e-5, atol=1e-7)
        if name == "continuous":  # the fused Normal variant computes the log-probability itself
            losses.zero_()
            g_loc, g_ls = torch.empty((B, K), device=dev), torch.empty((B, K), device=dev)
            loc_t, ls_t, act_t = t(s["loc"]), t(s["log_scale"]), t(s["action"])
            N.check(lib.srlx_ppo_loss_normal(B, K, N.tptr(loc_t), N.tptr(ls_t), math.log(1e-10), math.log(10), N.tptr(act_t),
                                             N.tptr(olp_t), N.tptr(adv_t), N.tptr(v_t), N.tptr(v_t), N.tptr(v_t), 0, 1, float(z["clip_range"]), 0, 0.0,
                                             1.0, 1.0, N.tptr(losses), N.tptr(g_loc), N.tptr(g_ls), N.tptr(g_v), None))
            torch.cuda.synchronize()
            got = losses.cpu().numpy()
            np.testing.assert_allclose(got[0], s["loss_policy"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(got[2], s["loss_e"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_gae_kernel_known_answer():
    import torch

    from simple_distributed_rl_amd import _native as N

    lib, dev = N.lib(), torch.device("cuda:0")
    r = torch.tensor([[1.0], [0.0], [2.0], [1.0], [0.0], [2.0]], device=dev)
    v = torch.tensor([[0.5], [0.25], [1.0], [0.5], [0.25], [1.0]], device=dev)
    d = torch.tensor([[0], [0], [1], [0], [0], [1]], dtype=torch.uint8, device=dev)
    out = torch.empty((6, 1), device=dev)
    N.check(lib.srlx_gae_scan(1, 6, N.tptr(r), N.tptr(v), N.tptr(d), None, 0.5, 0.5, N.tptr(out), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy()[:, 0], np.array([0.75, 0.5, 1.0] * 2, np.float32))
    cut = torch.zeros((3, 1), dtype=torch.uint8, device=dev)
    last = torch.tensor([7.0], device=dev)
    out3 = torch.empty((3, 1), device=dev)
    r3, v3 = r[:3].contiguous(), v[:3].contiguous()
    N.check(lib.srlx_gae_scan(1, 3, N.tptr(r3), N.tptr(v3), N.tptr(cut), N.tptr(last), 0.5, 0.5, N.tptr(out3), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out3.cpu().numpy()[:, 0], np.array([0.96875, 1.375, 4.5], np.float32))
