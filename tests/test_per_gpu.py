"""GPU parity tests of the sum-tree kernels, through the C ABI (libsrlx.so), against
(1) golden traces recorded from the imported reference and (2) the CPU oracle on seeded inputs.
Bar: tree indices / tree contents / uniform consumption bit-exact; IS weights rel 1e-13 (the
device and glibc `pow` may differ in the last ulp); transformed priorities <= 1 ulp (exact at alpha=0.5)."""
import ctypes
import glob
import os
import random

import numpy as np
import pytest

from oracle_bindings import OraclePER, iter_trace

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TRACES = sorted(glob.glob(os.path.join(GOLDEN, "per_trace_*.npz")))
IDS = [os.path.basename(p)[10:-4] for p in TRACES]
W_RTOL = 1e-13


def _N():
    from simple_distributed_rl_amd import _native as N

    return N


class AbiPER:
    """Minimal direct caller of the C ABI with host arrays (no Python shim logic)."""

    def __init__(self, capacity, alpha, beta_initial, beta_steps, has_duplicate, epsilon):
        N = _N()
        self.N, self.lib = N, N.lib()
        self.capacity = int(capacity)
        h = N.c_p()
        N.check(self.lib.srlx_per_create(ctypes.byref(h), self.capacity, alpha, beta_initial, beta_steps, int(has_duplicate), epsilon, 0))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.srlx_per_destroy(self.h)
            self.h = None

    def add(self, values=None, kind=None, n=None):
        N = self.N
        if values is None:
            N.check(self.lib.srlx_per_add(self.h, int(n or 1), None, N.PRIO_NONE, 0, None))
        else:
            v = np.ascontiguousarray(values)
            N.check(self.lib.srlx_per_add(self.h, v.size, N.np_ptr(v), kind, 0, None))

    def sample(self, B, step, uniforms):
        N = self.N
        u = np.ascontiguousarray(uniforms, np.float64)
        idx = np.empty(B, np.int64)
        w = np.empty(B, np.float64)
        w32 = np.empty(B, np.float32)
        used = N.c_i64(0)
        st = self.lib.srlx_per_sample(self.h, B, int(step), None, N.np_ptr(u), u.size, N.np_ptr(idx), N.np_ptr(w), N.np_ptr(w32), ctypes.byref(used), 0, None)
        return st, used.value, idx, w, w32

    def update(self, indices, pri, kind):
        N = self.N
        i = np.ascontiguousarray(indices, np.int64)
        p = np.ascontiguousarray(pri)
        N.check(self.lib.srlx_per_update(self.h, i.size, N.np_ptr(i), N.np_ptr(p), kind, 0, None))

    def state(self):
        N = self.N
        mp, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
        tree = np.empty(2 * self.capacity - 1, np.float64)
        N.check(self.lib.srlx_per_backup(self.h, ctypes.byref(mp), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
        return mp.value, size.value, write.value, tree


def _mk(cls, z):
    return cls(int(z["capacity"]), float(z["alpha"]), float(z["beta_initial"]), float(z["beta_steps"]), bool(z["has_duplicate"]), float(z["epsilon"]))


@pytest.mark.parametrize("path", TRACES, ids=IDS)
def test_abi_replays_reference_trace_bit_exact(path):
    """Golden traces from the imported reference through the C ABI: indices, number of consumed
    uniforms (retries included) and the final tree are bit-equal."""
    N = _N()
    z = np.load(path)
    m = _mk(AbiPER, z)
    eps, alpha = float(z["epsilon"]), float(z["alpha"])
    slack = np.random.default_rng(0).random(5)
    for kind, p in iter_trace(z):
        if kind == "add":
            if p["priority"] is None:
                m.add(None)
            else:
                m.add(np.array([(abs(p["priority"]) + eps) ** alpha]), N.PRIO_RAW)  # proportional_memory.py:124 on the host
        elif kind == "sample":
            u = np.concatenate([p["uniforms"], slack])  # extra uniforms must NOT be consumed
            st, used, idx, w, w32 = m.sample(p["batch_size"], p["step"], u)
            assert st == 0
            assert used == p["uniforms"].size
            np.testing.assert_array_equal(idx, p["indices"])
            np.testing.assert_allclose(w, p["weights"], rtol=W_RTOL, atol=0)
            np.testing.assert_array_equal(w32, w.astype(np.float32))
            if used > p["batch_size"]:  # with one uniform too few the call must report exhaustion
                st2, used2, *_ = m.sample(p["batch_size"], p["step"], p["uniforms"][:-1])
                assert st2 == N.ERR_UNIFORMS_EXHAUSTED and used2 == -1
        else:
            m.update(p["indices"], p["transformed"], N.PRIO_RAW)
    mp, size, write, tree = m.state()
    assert (size, write) == (int(z["final_size"]), int(z["final_write"]))
    assert mp == float(z["final_max_priority"])
    np.testing.assert_array_equal(tree, z["final_tree"])


@pytest.mark.parametrize("path", [p for p in TRACES if "speedtest" not in p], ids=[i for i in IDS if "speedtest" not in i])
def test_shim_lockstep_with_python_random(path):
    """The Python shim (IPriorityMemory) under random.seed(s): same indices as the reference and the
    `random` generator ends in the same state (rejected draws consume a value each, :146-157)."""
    from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory

    z = np.load(path)
    m = _mk(ProportionalMemory, z)
    random.seed(int(z["seed"]))
    item = 0
    for kind, p in iter_trace(z):
        if kind == "add":
            m.add(item, p["priority"])
            item += 1
        elif kind == "sample":
            batches, w, idx = m.sample(p["batch_size"], p["step"])
            assert idx == p["indices"].tolist()
            assert isinstance(w, np.ndarray) and w.dtype == np.float64
            np.testing.assert_allclose(w, p["weights"], rtol=W_RTOL, atol=0)
            cap = int(z["capacity"])
            assert all(b is not None for b in batches)
            assert len(batches) == p["batch_size"]
        else:
            m.update(p["indices"].tolist(), p["priorities"])
    assert random.random() == float(z["final_next_random"])
    np.testing.assert_array_equal(m.tree_array(), z["final_tree"])
    assert m.length() == int(z["final_size"])


@pytest.mark.parametrize("alpha", [0.5, 0.6, 1.0, 0.0])
@pytest.mark.parametrize("kind_name", ["f32", "f64"])
def test_device_transform_matches_oracle(alpha, kind_name):
    """Device-side (|p|+eps)^alpha (SRLX_PRIO_F32 / _F64) vs the oracle: exact for alpha in {0, .5, 1},
    <= 1 ulp of the priority dtype otherwise."""
    N = _N()
    cap = 777
    rng = np.random.default_rng(3)
    pri = (rng.standard_normal(cap) * 2).astype(np.float32 if kind_name == "f32" else np.float64)
    g = AbiPER(cap, alpha, 0.4, 1000, True, 1e-4)
    o = OraclePER(cap, alpha, 0.4, 1000, True, 1e-4)
    g.add(None, n=cap)
    for _ in range(cap):
        o.add(None)
    leaves = np.arange(cap, dtype=np.int64) + cap - 1
    g.update(leaves, pri, N.PRIO_F32 if kind_name == "f32" else N.PRIO_F64)
    o.update(leaves, pri)
    got, want = g.state()[3][leaves], o.tree()[leaves]
    if alpha in (0.0, 0.5, 1.0):
        np.testing.assert_array_equal(got, want)
    else:
        ulp = np.spacing(want.astype(np.float32)).astype(np.float64) if kind_name == "f32" else np.spacing(want)
        assert np.all(np.abs(got - want) <= ulp)
    assert g.state()[0] == pytest.approx(o.max_priority, rel=1e-15)


@pytest.mark.parametrize("capacity", [1, 2, 3, 5, 64, 1000, 4097])
def test_random_op_mix_vs_oracle(capacity):
    """Seeded add/sample/update mixes (ring wrap-around, both leaf depths, partially filled tree,
    bulk adds, duplicate indices) against the oracle: everything bit-exact (alpha=0.5 path)."""
    N = _N()
    rng = np.random.default_rng(capacity)
    g = AbiPER(capacity, 0.5, 0.4, 5000, True, 1e-4)
    o = OraclePER(capacity, 0.5, 0.4, 5000, True, 1e-4)
    step = 0
    for it in range(60):
        n = int(rng.integers(1, max(2, min(capacity, 300) + 1)))
        n = min(n, capacity)
        mode = it % 3
        if mode == 0:
            g.add(None, n=n)
            for _ in range(n):
                o.add(None)
        else:
            v = rng.random(n) * 3
            g.add(v, N.PRIO_F64)
            for x in v:
                o.add(float(np.sqrt(abs(x) + 1e-4)), mode=2)  # numpy-f64 semantics: sqrt at alpha=.5
        B = int(rng.integers(1, 70))
        u = rng.random(B + 40)
        st, used, idx, w, _ = g.sample(B, step, u)
        oused, oidx, ow, _ = o.sample(B, step, u)
        assert st == 0 and used == oused
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_allclose(w, ow, rtol=W_RTOL, atol=0)
        upd = np.concatenate([idx, idx[: B // 3]])
        pri = np.abs(rng.standard_normal(upd.size)).astype(np.float32)
        g.update(upd, pri, N.PRIO_F32)
        o.update(upd, pri)
        step += 37
    mp, size, write, tree = g.state()
    omp, osize, owrite, otree = o.get_state()
    np.testing.assert_array_equal(tree, otree)
    assert (mp, size, write) == (omp, osize, owrite)


def test_full_size_1M_bulk_paths():
    """BASELINE size: capacity 1,000,000 (tree 16 MB).  Bulk add of 1M priorities, bulk sample of
    200k draws (LDS-staged multi-workgroup path), chunked update of 5000 indices, all vs the oracle."""
    N = _N()
    cap = 1_000_000
    rng = np.random.default_rng(0)
    g = AbiPER(cap, 0.5, 0.4, 1_000_000, True, 1e-4)
    o = OraclePER(cap, 0.5, 0.4, 1_000_000, True, 1e-4)
    pri = rng.random(cap)  # |delta| ~ U(0,1) like speedtest.py:40-41
    tx = np.sqrt(np.abs(pri) + 1e-4)
    g.add(pri[:600_000], N.PRIO_F64)
    g.add(pri[600_000:], N.PRIO_F64)
    for x in tx:
        o.add(float(x), mode=2)
    np.testing.assert_array_equal(g.state()[3], o.tree())
    # wrap-around bulk add with priority=None
    g.add(None, n=123_457)
    for _ in range(123_457):
        o.add(None)
    np.testing.assert_array_equal(g.state()[3], o.tree())
    B = 200_000
    u = rng.random(B)
    st, used, idx, w, w32 = g.sample(B, 250_000, u)
    oused, oidx, ow, _ = o.sample(B, 250_000, u)
    assert st == 0 and used == oused == B
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_allclose(w, ow, rtol=W_RTOL, atol=0)
    # size-independent properties: indices are leaves; weights in (0,1], max == 1
    assert idx.min() >= cap - 1 and idx.max() <= 2 * cap - 2
    assert w.max() == 1.0 and w.min() > 0
    upd = idx[:5000]
    p32 = np.abs(rng.standard_normal(5000)).astype(np.float32)
    g.update(upd, p32, N.PRIO_F32)
    o.update(upd, p32)
    mp, size, write, tree = g.state()
    np.testing.assert_array_equal(tree, o.tree())
    assert mp == o.max_priority and size == cap and write == o.write
    # root == what the reference's delta propagation gives, and close to the exact leaf sum
    assert abs(tree[0] - tree[cap - 1 :].sum()) / tree[0] < 1e-9


@pytest.mark.parametrize("capacity,n", [(20_000, 8192), (9000, 5000), (1_000_448, 8192), (6000, 3000)])
def test_multi_gpu_sized_adds_vs_oracle(capacity, n):
    """The learner rank of an N-GPU run adds N x E consecutive slots per step (8192 = 8 x 1024): the single-workgroup
    add with its per-leaf changes in 64 KB of LDS, incl. ring wrap-around, masked (zero-priority) slots and both leaf
    depths -- tree bit-equal to the oracle's one-by-one adds."""
    N = _N()
    rng = np.random.default_rng(n)
    g = AbiPER(capacity, 0.5, 0.4, 5000, True, 1e-4)
    o = OraclePER(capacity, 0.5, 0.4, 5000, True, 1e-4)
    rounds = 4 if capacity < 100_000 else 2
    for it in range(rounds):
        if it % 2 == 0:
            v = rng.random(n) * 2
            g.add(v, N.PRIO_F64)
            for x in v:
                o.add(float(np.sqrt(abs(x) + 1e-4)), mode=2)
        else:
            g.add(None, n=n)
            for _ in range(n):
                o.add(None)
    mp, size, write, tree = g.state()
    omp, osize, owrite, otree = o.get_state()
    np.testing.assert_array_equal(tree, otree)
    assert (mp, size, write) == (omp, osize, owrite)


@pytest.mark.parametrize("cap", [1_000_000, 300_001])
def test_bulk_walk_per_cu_configuration_equals_oracle(cap):
    """Calls of >= 2^19 draws run the one-workgroup-per-CU walk (14 top levels of left children staged in 150 KB of
    LDS).  Same indices, uniform consumption and weights as the oracle on the rejecting path, twice in a row (the
    kernels re-arm their own counters), then on the no-rejection path, and for smaller calls (256-thread
    configuration) on the same handle.  A zero-priority leaf has zero width, so rejections are forced the way the
    reference can meet them: the leftmost leaf holds priority 0 and some uniforms are exactly 0.0 (always-left walk)."""
    N = _N()
    rng = np.random.default_rng(5)
    g = AbiPER(cap, 0.5, 0.4, 1_000_000, True, 1e-4)
    o = OraclePER(cap, 0.5, 0.4, 1_000_000, True, 1e-4)
    pri = rng.random(cap)
    pri[rng.random(cap) < 0.02] = 0.0
    node = 0
    while 2 * node + 1 < 2 * cap - 1:
        node = 2 * node + 1
    leftmost = node - (cap - 1)
    pri[leftmost] = 0.0
    g.add(pri, N.PRIO_RAW)
    for x in pri:
        o.add(float(x), mode=2)
    np.testing.assert_array_equal(g.state()[3], o.tree())

    def same(M, B, step):
        u = rng.random(M)
        u[rng.random(M) < 0.03] = 0.0
        st, used, idx, w, w32 = g.sample(B, step, u)
        oused, oidx, ow, _ = o.sample(B, step, u)
        assert st == 0 and used == oused
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_allclose(w, ow, rtol=W_RTOL, atol=0)
        return used

    assert same(600_000, 560_000, 1000) > 560_000
    assert same(600_000, 560_000, 2000) > 560_000
    # more accepted draws wanted than the uniforms can give: reported, and the next call is unaffected
    u = rng.random(600_000)
    u[rng.random(600_000) < 0.03] = 0.0
    st, used, *_ = g.sample(599_000, 2500, u)
    assert st == N.ERR_UNIFORMS_EXHAUSTED and used == -1
    assert same(150_000, 120_000, 3000) > 120_000
    pri2 = rng.random(cap) + 0.01  # the leftmost leaf is drawable again: u == 0.0 is accepted
    g.add(pri2, N.PRIO_RAW)
    for x in pri2:
        o.add(float(x), mode=2)
    np.testing.assert_array_equal(g.state()[3], o.tree())
    assert same(600_000, 600_000, 4000) == 600_000
    assert same(1 << 20, 1 << 20, 5000) == 1 << 20
    assert same(100_000, 100_000, 6000) == 100_000


def test_on_device_pointers_with_torch():
    """on_device=1: device pointers (torch tensors), work enqueued on torch's current stream,
    step read from a device scalar -- same results as the host-pointer path."""
    import torch

    N = _N()
    cap = 5000
    rng = np.random.default_rng(5)
    g = AbiPER(cap, 0.5, 0.4, 10_000, True, 1e-4)
    o = OraclePER(cap, 0.5, 0.4, 10_000, True, 1e-4)
    dev = torch.device("cuda:0")
    st = N.torch_stream_ptr()
    pri = torch.tensor(rng.random(cap), dtype=torch.float64, device=dev)
    N.check(g.lib.srlx_per_add(g.h, cap, N.c_p(pri.data_ptr()), N.PRIO_F64, 1, st))
    for x in np.sqrt(pri.cpu().numpy() + 1e-4):
        o.add(float(x), mode=2)
    B = 64
    u_np = rng.random(B + 8)
    u = torch.tensor(u_np, device=dev)
    idx = torch.empty(B, dtype=torch.int64, device=dev)
    w = torch.empty(B, dtype=torch.float64, device=dev)
    w32 = torch.empty(B, dtype=torch.float32, device=dev)
    used = torch.zeros(1, dtype=torch.int64, device=dev)
    d_step = torch.tensor([1234], dtype=torch.int64, device=dev)
    N.check(
        g.lib.srlx_per_sample(g.h, B, 0, N.c_p(d_step.data_ptr()), N.c_p(u.data_ptr()), B + 8, N.c_p(idx.data_ptr()), N.c_p(w.data_ptr()), N.c_p(w32.data_ptr()), N.c_p(used.data_ptr()), 1, st)
    )
    oused, oidx, ow, _ = o.sample(B, 1234, u_np)
    torch.cuda.synchronize()
    assert int(used.item()) == oused
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    np.testing.assert_allclose(w.cpu().numpy(), ow, rtol=W_RTOL)
    p32 = torch.tensor(np.abs(rng.standard_normal(B)), dtype=torch.float32, device=dev)
    N.check(g.lib.srlx_per_update(g.h, B, N.c_p(idx.data_ptr()), N.c_p(p32.data_ptr()), N.PRIO_F32, 1, st))
    o.update(oidx, p32.cpu().numpy())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(g.state()[3], o.tree())
    assert g.state()[0] == o.max_priority


def test_reference_statistical_scenario():
    """tests/quick/rl/memories/test_priority_memories.py:29-91 ported onto the HIP memory: hit counts
    strictly increase with priority, no duplicates when has_duplicate=False, backup/restore round trip."""
    import collections

    from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory

    random.seed(11)
    for check_dup in (True, False):
        memory = ProportionalMemory(10, 0.8, 1, 10, has_duplicate=not check_dup)
        for i in range(100):
            memory.add((i, i, i, i), 0)
        assert memory.length() == 10
        for i in range(10):
            i += 1
            memory.add((i, i, i, i), i)
        counter = []
        for i in range(4000):
            batches, weights, update_args = memory.sample(5, step=1)
            assert len(batches) == 5 and len(weights) == 5
            if check_dup:
                assert len(set(batches)) == 5
            counter.extend(b[0] for b in batches)
            memory.update(update_args, np.array([b[3] for b in batches]))
            if i % 50 == 0:
                l1 = memory.length()
                memory.restore(memory.backup())
                assert l1 == memory.length()
        c = collections.Counter(counter)
        keys = sorted(c.keys())
        if check_dup:
            assert keys == list(range(1, 11))
        vals = [c[k] for k in keys]
        assert all(vals[i] < vals[i + 1] for i in range(len(vals) - 1))


@pytest.mark.parametrize("alpha", [0, 0.2, 0.5, 0.8, 1.0])
def test_reference_is_weight_known_answer(alpha):
    """tests/quick/rl/memories/test_priority_memories.py:97-117,150-176 on the HIP memory (rel 1e-7)."""
    import math

    from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory

    eps = 1e-4
    memory = ProportionalMemory(capacity=10, alpha=alpha, beta_initial=1, epsilon=eps, has_duplicate=False)
    pri = [1, 2, 4, 3]
    true_p = [(t + eps) ** alpha for t in pri]
    s = sum(true_p)
    tw = np.array([(4 * (p / s)) ** -1 for p in true_p])
    tw /= tw.max()
    for i, p in enumerate(pri):
        memory.add((i, i, i, i), priority=p)
    batches, weights, _ = memory.sample(4, step=1)
    for i, b in enumerate(batches):
        assert math.isclose(weights[i], tw[b[0]], rel_tol=1e-7)


def test_restore_into_different_capacity():
    from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory

    a = ProportionalMemory(7, alpha=0.5)
    for i in range(11):
        a.add(("item", i), float(i + 1))
    b = ProportionalMemory(5, alpha=0.5)
    b.restore(a.backup())
    o = OraclePER(7, alpha=0.5)
    for i in range(11):
        o.add(float(i + 1))
    o2 = OraclePER(5, alpha=0.5)
    _, size, _, tree = o.get_state()
    o2.restore_resized(7, size, tree)
    np.testing.assert_array_equal(b.tree_array(), o2.tree())
    assert b.length() == 5
    bk = b.backup()
    assert bk[0] == 5 and bk[3] == o2.write
    c = ProportionalMemory(5, alpha=0.5)
    c.restore(bk)
    np.testing.assert_array_equal(c.tree_array(), o2.tree())
    assert c.data == b.data


@pytest.mark.parametrize("capacity,first,n", [(1000, 0, 1000), (1000, 990, 300), (5, 3, 4), (1 << 12, 17, 3000), (1_000_000, 999_000, 200_000)])
def test_set_range_equals_update(capacity, first, n):
    """srlx_per_set_range (bulk contiguous-run kernels) == update() of the same consecutive leaves,
    incl. ring wrap-around, both leaf depths and max_priority tracking."""
    N = _N()
    rng = np.random.default_rng(capacity + n)
    g = AbiPER(capacity, 0.5, 0.4, 1000, True, 1e-4)
    o = OraclePER(capacity, 0.5, 0.4, 1000, True, 1e-4)
    g.add(None, n=capacity)
    for _ in range(capacity):
        o.add(None)
    pri = (rng.random(n) * 3).astype(np.float32)
    N.check(g.lib.srlx_per_set_range(g.h, first, n, N.np_ptr(pri), N.PRIO_F32, 0, None))
    idx = (np.arange(first, first + n) % capacity) + capacity - 1
    o.update(idx, pri)
    mp, size, write, tree = g.state()
    np.testing.assert_array_equal(tree, o.tree())
    assert mp == o.max_priority and size == capacity and write == o.write


def test_rankbased_memory_matches_reference_trace():
    """The HIP-backed RankBasedMemory (device radix sort for rank -> index) replays the reference's recorded trace under
    the same numpy seed: same sampled indices (the stored items), same weights, same final priorities; backup/restore."""
    from simple_distributed_rl_amd.rl.memories.priority_memories.rankbased_memory import RankBasedMemory

    z = np.load(os.path.join(GOLDEN, "rankbased_trace.npz"))
    mem = RankBasedMemory(int(z["capacity"]), float(z["alpha"]), float(z["beta_initial"]), int(z["beta_steps"]))
    np.random.seed(int(z["seed"]))
    k = 0
    for rnd in range(len(z["n_add"])):
        for _ in range(int(z["n_add"][rnd])):
            mem.add(int(k), float(z["add_priorities"][k]))
            k += 1
        batches, weights, idx = mem.sample(16, 100 * rnd)
        np.testing.assert_array_equal(np.asarray(idx), z["indices"][rnd])
        np.testing.assert_array_equal(np.asarray(batches), z["batches"][rnd])
        np.testing.assert_array_equal(weights, z["weights"][rnd])
        mem.update(idx, z["new_priorities"][rnd])
    np.testing.assert_array_equal(mem.priorities, z["final_priorities"])
    snap = mem.backup()
    m2 = RankBasedMemory(int(z["capacity"]), float(z["alpha"]), float(z["beta_initial"]), int(z["beta_steps"]))
    m2.restore(snap)
    np.random.seed(5)
    a = mem.sample(16, 0)
    np.random.seed(5)
    b = m2.sample(16, 0)
    np.testing.assert_array_equal(a[2], b[2])
    assert m2.length() == mem.length() == int(z["capacity"])


def test_rankbased_select_large_vs_numpy():
    """srlx_rank_select at N = 1e6 distinct priorities: every rank maps to np.argsort(-p)[rank]."""
    import ctypes

    import torch

    N = _N()
    lib = N.lib()
    n = 1_000_000
    rng = np.random.default_rng(0)
    pri = rng.permutation(n).astype(np.float32) / 3
    h = N.c_p()
    N.check(lib.srlx_rank_create(ctypes.byref(h), n, 0))
    d = torch.from_numpy(pri).cuda()
    N.check(lib.srlx_rank_set(h, n, None, N.tptr(d), 0, None))
    ranks = rng.integers(0, n, 4096)
    dr = torch.from_numpy(ranks).cuda()
    out = torch.empty(4096, dtype=torch.int64, device="cuda")
    N.check(lib.srlx_rank_select(h, n, 4096, N.tptr(dr), N.tptr(out), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), np.argsort(-pri, kind="stable")[ranks])
    lib.srlx_rank_destroy(h)


def test_no_duplicate_sampling_with_too_few_leaves_completes_like_the_reference():
    """has_duplicate=False with fewer distinct non-zero leaves than the batch: the reference tries 9999 times per draw and then takes the draw
    anyway (proportional_memory.py:146-158) -- it keeps training.  The shim completes the batch with duplicates instead of raising; with
    enough leaves the no-duplicate rule holds."""
    from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory

    m = ProportionalMemory(64, alpha=0.5, has_duplicate=False)
    for i in range(10):
        m.add(("item", i), float(i + 1))
    random.seed(3)
    batches, w, idx = m.sample(16, 0)
    assert len(idx) == 16 and len(set(idx)) <= 10 and all(b is not None for b in batches) and np.isfinite(w).all()
    batches, w, idx = m.sample(10, 0)  # exactly as many as there are leaves: all distinct (coupon collector inside the 8192-uniform bound)
    assert len(set(idx)) == 10
    for i in range(10, 40):
        m.add(("item", i), 1.0)
    _, _, idx = m.sample(16, 0)
    assert len(set(idx)) == 16


@pytest.mark.parametrize("has_duplicate", [True, False])
def test_keyed_sample_and_update_counter_equal_the_separate_launches(has_duplicate):
    """srlx_per_sample_keyed = srlx_rng_uniform + srlx_per_sample (same indices, weights, consumed uniforms, counter) -- also
    with zero-priority and duplicate rejects; srlx_per_set_update_counter: every device-side update adds one."""
    import torch

    N = _N()
    lib = N.lib()
    dev = torch.device("cuda:0")
    per = AbiPER(300, 0.7, 0.4, 1000, has_duplicate, 1e-4)
    rng = np.random.default_rng(5)
    vals = rng.random(300)
    vals[rng.random(300) < 0.3] = 0.0  # zero-priority leaves: rejected draws
    N.check(lib.srlx_per_add(per.h, 300, N.np_ptr(vals), N.PRIO_RAW, 0, None))
    B, M, seed = 32, 96, 0xABCDEF
    step = torch.tensor([17], dtype=torch.int64, device=dev)
    c_a, c_b = torch.tensor([5], dtype=torch.int64, device=dev), torch.tensor([5], dtype=torch.int64, device=dev)
    u = torch.zeros(M, dtype=torch.float64, device=dev)
    out = []
    for rounds in range(3):
        ia, wa, wa32, ua = (torch.zeros(B, dtype=torch.int64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev),
                            torch.zeros(B, dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
        ib, wb, wb32, ub = torch.zeros_like(ia), torch.zeros_like(wa), torch.zeros_like(wa32), torch.zeros_like(ua)
        N.check(lib.srlx_rng_uniform(seed, N.tptr(c_a), M, N.tptr(u), None))
        N.check(lib.srlx_per_sample(per.h, B, 0, N.tptr(step), N.tptr(u), M, N.tptr(ia), N.tptr(wa), N.tptr(wa32), N.tptr(ua), 1, None))
        N.check(lib.srlx_per_sample_keyed(per.h, B, N.tptr(step), seed, N.tptr(c_b), M, N.tptr(ib), N.tptr(wb), N.tptr(wb32), N.tptr(ub), None))
        torch.cuda.synchronize()
        assert int(ua) > 0 and torch.equal(ua, ub) and torch.equal(c_a, c_b) and int(c_a) == 6 + rounds
        assert torch.equal(ia, ib) and torch.equal(wa, wb) and torch.equal(wa32, wb32)
        out.append(ia.cpu().numpy().copy())
    assert not np.array_equal(out[0], out[1])  # the counter moved: fresh uniforms every call
    if not has_duplicate:
        assert all(len(set(o.tolist())) == B for o in out)
    count = torch.tensor([40], dtype=torch.int64, device=dev)
    N.check(lib.srlx_per_set_update_counter(per.h, N.tptr(count)))
    pri = torch.rand(B, dtype=torch.float32, device=dev)
    for _ in range(3):
        N.check(lib.srlx_per_update(per.h, B, N.tptr(ib), N.tptr(pri), N.PRIO_F32, 1, None))
    torch.cuda.synchronize()
    assert int(count) == 43
    N.check(lib.srlx_per_set_update_counter(per.h, None))
    N.check(lib.srlx_per_update(per.h, B, N.tptr(ib), N.tptr(pri), N.PRIO_F32, 1, None))
    torch.cuda.synchronize()
    assert int(count) == 43


@pytest.mark.parametrize("capacity,n", [(1_000_000, 64), (1_000_000, 32), (300_001, 64), (5000, 48)])
def test_learner_sized_updates_200_repeats_vs_oracle(capacity, n):
    """The learner's priority write-back (k_update_wg at 32 / 64 indices: tree values prefetched ahead of the ownership scans -- round 5) 200 times over on one
    tree, against the oracle call by call: random leaves with clustered neighbours (shared ancestors far down) and repeated indices, the final tree bit-equal and
    max_priority equal after every 50th call.  (Round 4's first rewrite of this kernel raced once in ~100 calls; hence the repeats.)"""
    N = _N()
    rng = np.random.default_rng(capacity + n)
    g = AbiPER(capacity, 0.5, 0.4, 5000, True, 1e-4)
    o = OraclePER(capacity, 0.5, 0.4, 5000, True, 1e-4)
    v = rng.random(capacity) * 2
    g.add(v, N.PRIO_F64)
    for x in np.sqrt(np.abs(v) + 1e-4):
        o.add(float(x), mode=2)
    for rep in range(200):
        base = rng.integers(0, capacity, n)
        k = rep % 4
        if k == 1:  # neighbours: paths that part only in the last levels
            base[n // 2:] = np.minimum(base[: n - n // 2] + rng.integers(0, 3, n - n // 2), capacity - 1)
        elif k == 2:  # repeated indices (the later occurrence sees the earlier one's write)
            base[n // 3:] = base[rng.integers(0, n // 3, n - n // 3)]
        idx = base + capacity - 1
        pri = (np.abs(rng.standard_normal(n)) * (3.0 if rep % 17 == 0 else 1.0)).astype(np.float32)
        g.update(idx, pri, N.PRIO_F32)
        o.update(idx, pri)
        if rep % 50 == 49:
            mp, size, write, tree = g.state()
            omp, osize, owrite, otree = o.get_state()
            np.testing.assert_array_equal(tree, otree)
            assert (mp, size, write) == (omp, osize, owrite)
