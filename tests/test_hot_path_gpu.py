"""GPU parity tests of the rollout / gather / learner-arithmetic kernels through the C ABI, against
golden vectors recorded from the reference and against oracle/hot_path_oracle.py on seeded inputs.
Bar: integer/byte/index outputs and float32 frames bit-exact; TD targets / loss / priorities within
1e-5 relative (north_star), in practice bit-exact or 1 ulp."""
import ctypes
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hot_path_oracle as H  # noqa: E402
from test_hot_path_oracle_golden import replay_log_into_store  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
RTOL = 1e-5  # north_star tolerance for losses / Q-derived values


def _env():
    import torch

    from simple_distributed_rl_amd import _native as N

    return N, N.lib(), torch, torch.device("cuda:0")


_KEEP = []  # device inputs must outlive the (asynchronous) kernels that read them


def T(x, dtype=None):
    import torch

    t = torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


class GpuStore:
    def __init__(self, E, L, F, W, n, A, clip, seed, u8=True):
        N, lib, torch, dev = _env()
        self.N, self.lib, self.torch, self.dev = N, lib, torch, dev
        self.E, self.L, self.F, self.W, self.n, self.A, self.u8 = E, L, F, W, n, A, u8
        h = N.c_p()
        N.check(lib.srlx_store_create(ctypes.byref(h), E, L, F, N.OBS_U8 if u8 else N.OBS_F32, W, n, A, int(clip), seed, 0))
        self.h = h
        self.st = None  # NULL stream = the handle's own stream; we sync explicitly

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.srlx_store_destroy(self.h)
            self.h = None

    def sync(self):
        self.torch.cuda.synchronize()

    def reset_all(self, first):
        t = T(first, self.torch.uint8 if self.u8 else self.torch.float32)
        self.N.check(self.lib.srlx_store_reset_all(self.h, self.N.tptr(t), None))
        self.sync()

    def stack_current(self):
        out = self.torch.empty((self.E, self.W, self.F), dtype=self.torch.float32, device=self.dev)
        self.N.check(self.lib.srlx_store_stack_current(self.h, self.N.tptr(out), None))
        self.sync()
        return out.cpu().numpy()

    def commit_step(self, actions, rewards, term, done, next_obs):
        tt = self.torch
        a, r = T(actions, tt.int32), T(rewards, tt.float32)
        t_, d = T(term, tt.uint8), T(done, tt.uint8)
        o = T(next_obs, tt.uint8 if self.u8 else tt.float32)
        mask = tt.zeros(self.E, dtype=tt.uint8, device=self.dev)
        N = self.N
        N.check(self.lib.srlx_store_commit_step(self.h, N.tptr(a), N.tptr(r), N.tptr(t_), N.tptr(d), N.tptr(o), N.tptr(mask), None))
        self.sync()
        return mask.cpu().numpy()

    def gather(self, tree_idx):
        tt, N = self.torch, self.N
        B = len(tree_idx)
        idx = T(tree_idx, tt.int64)
        obs = tt.empty((B, self.n + 1, self.W, self.F), dtype=tt.float32, device=self.dev)
        act = tt.empty((B, self.n), dtype=tt.int32, device=self.dev)
        rew = tt.empty((B, self.n), dtype=tt.float32, device=self.dev)
        ter = tt.empty((B, self.n), dtype=tt.float32, device=self.dev)
        N.check(self.lib.srlx_store_gather_nstep(self.h, B, N.tptr(idx), N.tptr(obs), N.tptr(act), N.tptr(rew), N.tptr(ter), None))
        self.sync()
        return obs.cpu().numpy(), act.cpu().numpy(), rew.cpu().numpy(), ter.cpu().numpy()

    def synth_step(self, episode_len):
        tt, N = self.torch, self.N
        o = tt.empty((self.E, self.F), dtype=tt.uint8 if self.u8 else tt.float32, device=self.dev)
        r = tt.empty(self.E, dtype=tt.float32, device=self.dev)
        t_ = tt.empty(self.E, dtype=tt.uint8, device=self.dev)
        d = tt.empty(self.E, dtype=tt.uint8, device=self.dev)
        N.check(self.lib.srlx_synth_env_step(self.h, episode_len, N.tptr(o), N.tptr(r), N.tptr(t_), N.tptr(d), None))
        self.sync()
        return o.cpu().numpy(), r.cpu().numpy(), t_.cpu().numpy(), d.cpu().numpy()


def test_u8_normalisation_is_numpy_exact():
    """u8/255 in float32 for all 256 byte values == numpy `astype(float32) / 255` (image_processor.py:140-142)."""
    s = GpuStore(1, 16, 256, 1, 1, 2, False, 0)
    s.reset_all(np.arange(256, dtype=np.uint8)[None])
    got = s.stack_current()[0, 0]
    want = np.arange(256, dtype=np.uint8).astype(np.float32)
    want /= 255.0
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("name", ["terminated", "truncated"])
def test_store_reproduces_reference_items(name):
    """Golden: the reference Rainbow worker's emitted n-step items (rollout_items_*.npz) vs the HIP store."""
    z = np.load(os.path.join(GOLDEN, f"rollout_items_{name}.npz"))
    g = GpuStore(1, 128, 64, 4, 3, 4, True, 123)
    o = H.StoreOracle(1, 128, 64, 4, 3, 4, True, 123)
    replay_log_into_store(z, g)
    replay_log_into_store(z, o)
    valid = [q for q in range(o.pos) if not (o.flags[0, q] & o.INVALID)]
    n_items = z["item_obs"].shape[0]
    # PER slot of position q (E=1): added at commit p = q+n-1 -> tau = p % item_len; tree idx = tau + N-1
    N = o.E * o.item_len
    idx = [((q + o.n - 1) % o.item_len) + N - 1 for q in valid[:n_items]]
    assert all(o.locate(i) == (0, q) for i, q in zip(idx, valid))
    obs, act, rew, ter = g.gather(idx)
    np.testing.assert_array_equal(obs, z["item_obs"])
    np.testing.assert_array_equal(rew, z["item_rewards"])
    np.testing.assert_array_equal(ter, z["item_terminated"])
    oo, oa, orw, ot = o.gather_nstep(idx)
    np.testing.assert_array_equal(act, oa)  # incl. the keyed pseudo-random padded actions
    for i in range(n_items):
        real = min(o.gather_item(0, valid[i])[4] + 1, 3)  # rows after the episode end carry random actions
        np.testing.assert_array_equal(act[i][:real], z["item_actions"][i][:real])


@pytest.mark.parametrize("E,F,W,n,u8", [(5, 64, 4, 3, True), (3, 7056, 4, 3, True), (4, 10, 2, 1, True), (6, 8, 1, 5, False), (2, 6, 3, 2, False)])
def test_store_random_rollout_vs_oracle(E, F, W, n, u8):
    """Lock-step rollout with ring wrap-around, random episode ends, both frame dtypes, vector and scalar
    copy paths: stacked policy input, item masks and gathered batches are bit-equal to the store model."""
    rng = np.random.default_rng(E * 1000 + F)
    L = n + W + 9
    A = 5
    g = GpuStore(E, L, F, W, n, A, True, 77, u8)
    o = H.StoreOracle(E, L, F, W, n, A, True, 77, u8)

    def frame():
        return rng.integers(0, 256, (E, F), dtype=np.uint8) if u8 else rng.standard_normal((E, F)).astype(np.float32)

    f0 = frame()
    g.reset_all(f0)
    o.reset_all(f0)
    for step in range(3 * L):
        np.testing.assert_array_equal(g.stack_current(), o.stack_current())
        a = rng.integers(0, A, E).astype(np.int32)
        r = (rng.standard_normal(E) * 2).astype(np.float32)
        done = (rng.random(E) < 0.2).astype(np.uint8)
        term = (done & (rng.random(E) < 0.7)).astype(np.uint8)
        nxt = frame()
        mg = g.commit_step(a, r, term, done, nxt)
        mo = o.commit_step(a, r, term, done, nxt)
        np.testing.assert_array_equal(mg, mo)
        if step >= n + 2 and step % 3 == 0:
            N = E * o.item_len
            # every slot whose item is alive
            alive = min(step + 1, o.item_len)
            taus = [(o.pos - 1 - k) % o.item_len for k in range(alive)]
            idx = [t * E + e + N - 1 for t in taus for e in range(E)]
            keep = [i for i in idx if o.locate(i)[1] >= 0 and not (o.flags[o.locate(i)[0], o.locate(i)[1] % L] & o.INVALID)]
            if keep:
                for got, want in zip(g.gather(keep), o.gather_nstep(keep)):
                    np.testing.assert_array_equal(got, want)


def test_synthetic_env_matches_definition():
    g = GpuStore(7, 40, 7056, 4, 3, 6, True, 5)
    o = H.StoreOracle(7, 40, 7056, 4, 3, 6, True, 5)
    f0 = np.zeros((7, 7056), np.uint8)
    g.reset_all(f0)
    o.reset_all(f0)
    saw_done = False
    for step in range(14):
        got = g.synth_step(5)
        want = H.synth_env_step(o, 5)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)
        saw_done |= bool(want[3].any())
        acts = np.full(7, step % 6, np.int32)
        g.commit_step(acts, want[1], want[2], want[3], want[0])
        o.commit_step(acts, want[1], want[2], want[3], want[0])
    assert saw_done
    fr = want[0]
    assert abs(fr.mean() - 127.5) < 2 and fr.min() == 0 and fr.max() == 255


def test_epsilon_greedy_vs_oracle():
    N, lib, torch, dev = _env()
    rng = np.random.default_rng(1)
    E, A = 1000, 6
    q = rng.standard_normal((E, A)).astype(np.float32)
    q[5, 2] = q[5, 4] = 9.0  # tie -> first maximum like np.argmax
    eps = np.where(rng.random(E) < 0.5, 0.3, 0.0).astype(np.float32)
    u = rng.random((E, 2))
    invalid = rng.random((E, A)) < 0.2
    invalid[:, 0] &= ~invalid.all(axis=1)  # keep at least one valid action
    for inv in (None, invalid):
        out = torch.empty(E, dtype=torch.int32, device=dev)
        N.check(
            lib.srlx_policy_epsilon_greedy(E, A, N.tptr(T(q)), N.tptr(T(eps)), N.tptr(T(u)), N.tptr(T(inv, torch.uint8)) if inv is not None else None, N.tptr(out), None)
        )
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), H.epsilon_greedy(q, eps, u, inv))


def test_rng_uniform_vs_definition():
    N, lib, torch, dev = _env()
    counter = torch.tensor([41], dtype=torch.int64, device=dev)
    out = torch.empty(5000, dtype=torch.float64, device=dev)
    N.check(lib.srlx_rng_uniform(1234, N.tptr(counter), 5000, N.tptr(out), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), H.rng_uniform(1234, 41, 5000))
    assert int(counter.item()) == 42


def _run_td(z_q_on, z_q_tg, q0, actions, reward, done, invalid, w, discount, h, double_dqn, rescale):
    N, lib, torch, dev = _env()
    B, n, A = z_q_tg.shape
    q_on = z_q_on
    if q_on.shape[1] < n:  # non-double: the reference only evaluates the online net on s_1..s_{n-1}
        q_on = np.concatenate([q_on, np.zeros((B, n - q_on.shape[1], A), np.float32)], axis=1)
    tgt = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    grad = torch.empty((B, A), dtype=torch.float32, device=dev)
    pri = torch.empty(B, dtype=torch.float32, device=dev)
    N.check(
        lib.srlx_nstep_td_huber_priority(
            B, n, A, N.tptr(T(q_on)), N.tptr(T(z_q_tg)), N.tptr(T(q0)), N.tptr(T(actions, torch.int32)), N.tptr(T(reward)), N.tptr(T(done)),
            N.tptr(T(invalid, torch.uint8)) if invalid is not None else None, N.tptr(T(w)), float(discount), float(h), int(double_dqn), int(rescale),
            N.tptr(tgt), N.tptr(loss), N.tptr(grad), N.tptr(pri), None,
        )
    )
    torch.cuda.synchronize()
    return tgt.cpu().numpy(), float(loss.item()), grad.cpu().numpy(), pri.cpu().numpy()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "target_q_*.npz"))), ids=lambda p: os.path.basename(p)[9:-4])
def test_nstep_target_golden(path):
    """Golden calc_target_q vectors from the reference (rainbow.py:185-287)."""
    z = np.load(path)
    B, n, A = z["q_target"].shape
    q0 = np.zeros((B, A), np.float32)
    w = np.ones(B, np.float32)
    tgt, *_ = _run_td(z["q_online"], z["q_target"], q0, z["actions"], z["reward"], z["done"], z["invalid"], w,
                      z["discount"], z["retrace_h"], z["double_dqn"], z["rescale"])
    np.testing.assert_allclose(tgt, z["target_q"], rtol=RTOL, atol=1e-7)
    # in practice: bit-exact except where device pow/sqrt differ in the last ulp
    assert np.mean(tgt == z["target_q"]) > 0.9


def test_train_step_arithmetic_golden():
    """Golden Trainer.train() vectors (rainbow/model_torch.py:103-114): target, loss, d loss/d q, priorities."""
    z = np.load(os.path.join(GOLDEN, "train_step_rainbow.npz"))
    B = z["q_all"].shape[0]
    n, A = 3, int(z["n_actions"])
    # q_on_next / q_tg_next are not in this fixture: feed the recorded target through a degenerate
    # 1-step problem (reward = target, terminated = 1) so that the Huber half is checked on the recorded q rows
    tgt, loss, grad, pri = _run_td(
        np.zeros((B, 1, A), np.float32), np.zeros((B, 1, A), np.float32), z["q_all"], z["actions"][:, :1], z["target_q"][:, None],
        np.ones((B, 1), np.float32), None, z["weights"], 0.99, 1.0, True, False,
    )
    np.testing.assert_array_equal(tgt, z["target_q"])
    np.testing.assert_allclose(loss, z["loss"], rtol=RTOL)
    np.testing.assert_allclose(grad, z["grad_q"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(pri, z["priorities"], rtol=RTOL, atol=1e-7)


def test_nstep_td_random_vs_oracle():
    rng = np.random.default_rng(9)
    for B, n, A, dd, rs in [(32, 3, 6, True, False), (32, 3, 18, False, False), (100, 5, 4, True, True), (700, 1, 3, True, False)]:
        q_on = rng.standard_normal((B, n, A)).astype(np.float32)
        q_tg = rng.standard_normal((B, n, A)).astype(np.float32)
        q0 = rng.standard_normal((B, A)).astype(np.float32)
        act = rng.integers(0, A, (B, n)).astype(np.int32)
        rew = rng.integers(-1, 2, (B, n)).astype(np.float32)
        done = (rng.random((B, n)) < 0.2).astype(np.float32)
        inv = rng.random((B, n, A)) < 0.15
        inv[:, :, 0] = False
        w = rng.random(B).astype(np.float32)
        tgt, loss, grad, pri = _run_td(q_on, q_tg, q0, act, rew, done, inv, w, 0.99, 0.95, dd, rs)
        want = H.nstep_target(q_on if dd else q_on[:, : n - 1], q_tg, act, rew, done, inv, 0.99, 0.95, dd, rs)
        np.testing.assert_allclose(tgt, want, rtol=RTOL, atol=1e-6)
        wl, wg, wp = H.huber_loss_grad_priority(q0, act[:, 0], tgt, w)
        np.testing.assert_allclose(loss, wl, rtol=RTOL)
        np.testing.assert_allclose(grad, wg, rtol=RTOL, atol=1e-9)
        np.testing.assert_array_equal(pri, wp)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "dqn_target_*.npz"))), ids=lambda p: os.path.basename(p)[11:-4])
def test_dqn_target_golden(path):
    N, lib, torch, dev = _env()
    z = np.load(path)
    B, A = z["dqn_q_target"].shape
    for pre, f64 in (("dqn", 1), ("rb", 0)):
        out = torch.empty(B, dtype=torch.float32, device=dev)
        N.check(
            lib.srlx_dqn_target(B, A, N.tptr(T(z[pre + "_q_online"])), N.tptr(T(z[pre + "_q_target"])), N.tptr(T(z["reward"])),
                                N.tptr(T(z["undone"].astype(np.float32))), N.tptr(T(z["invalid"], torch.uint8)), float(z["discount"]),
                                None, int(z["double_dqn"]), int(z["rescale"]), f64, N.tptr(out), None)
        )
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), z[pre + "_target"], rtol=RTOL, atol=1e-7)


def test_gae_scan_vs_oracle():
    N, lib, torch, dev = _env()
    rng = np.random.default_rng(2)
    for E, Tn, boot in [(4096, 32, True), (33, 7, False), (1, 1, False)]:
        r = rng.standard_normal((Tn, E)).astype(np.float32)
        v = rng.standard_normal((Tn, E)).astype(np.float32)
        d = (rng.random((Tn, E)) < 0.1).astype(np.uint8)
        lv = rng.standard_normal(E).astype(np.float32) if boot else None
        out = torch.empty((Tn, E), dtype=torch.float32, device=dev)
        N.check(lib.srlx_gae_scan(E, Tn, N.tptr(T(r)), N.tptr(T(v)), N.tptr(T(d)), N.tptr(T(lv)) if boot else None, 0.9, 0.95, N.tptr(out), None))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), H.gae(r, v, d, lv, 0.9, 0.95))
