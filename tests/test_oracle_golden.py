"""Pins the CPU restatement (oracle/per_oracle.c) against vectors recorded from the imported
reference (oracle/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle_bindings import OraclePER, iter_trace

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TRACES = sorted(glob.glob(os.path.join(GOLDEN, "per_trace_*.npz")))


def _make(z):
    return OraclePER(
        int(z["capacity"]),
        float(z["alpha"]),
        float(z["beta_initial"]),
        float(z["beta_steps"]),
        bool(z["has_duplicate"]),
        float(z["epsilon"]),
    )


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[10:-4] for p in TRACES])
def test_oracle_replays_reference_trace_bit_exact(path):
    """Replaying the recorded op trace (with the reference host's own numpy transform for `update`,
    proportional_memory.py:172) must give bit-equal indices, uniform consumption and tree."""
    z = np.load(path)
    m = _make(z)
    n_samples = 0
    for kind, p in iter_trace(z):
        if kind == "add":
            m.add(p["priority"])
        elif kind == "sample":
            used, idx, w, _ = m.sample(p["batch_size"], p["step"], p["uniforms"])
            assert used == p["uniforms"].size  # same number of random.random() calls, retries included
            np.testing.assert_array_equal(idx, p["indices"])
            np.testing.assert_allclose(w, p["weights"], rtol=1e-15, atol=0)
            n_samples += 1
        else:
            m.update(p["indices"], p["transformed"], raw=True)
    assert n_samples > 0
    mp, size, write, tree = m.get_state()
    assert size == int(z["final_size"]) and write == int(z["final_write"])
    assert mp == float(z["final_max_priority"])
    np.testing.assert_array_equal(tree, z["final_tree"])


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[10:-4] for p in TRACES])
def test_oracle_own_transform_within_one_ulp(path):
    """The oracle's own (|p|+eps)^alpha (correctly rounded) vs the reference host's numpy value:
    equal for alpha=0.5 (sqrt fast path), <= 1 ulp of the priority dtype otherwise."""
    z = np.load(path)
    m = _make(z)
    alpha = float(z["alpha"])
    for kind, p in iter_trace(z):
        if kind != "update":
            continue
        idx = np.arange(p["priorities"].size, dtype=np.int64) % (2 * int(z["capacity"]) - 1)
        m.clear()
        # write to distinct leaves so each transformed value can be read back
        cap = int(z["capacity"])
        n = min(p["priorities"].size, cap)
        leaves = np.arange(n, dtype=np.int64) + cap - 1
        m.update(leaves, p["priorities"][:n])
        got = m.tree()[leaves]
        want = p["transformed"][:n]
        if alpha == 0.5:
            np.testing.assert_array_equal(got, want)
        else:
            ulp = np.spacing(want.astype(np.float32)).astype(np.float64) if p["priorities"].dtype == np.float32 else np.spacing(want)
            assert np.all(np.abs(got - want) <= ulp)


def test_oracle_is_weight_known_answer():
    """IS-weight KAT of the reference's own test (tests/quick/rl/memories/test_priority_memories.py:97-117,150-176)."""
    z = np.load(os.path.join(GOLDEN, "per_is_kat.npz"))
    import random

    for k, alpha in enumerate(z["alpha"]):
        m = OraclePER(10, alpha=float(alpha), beta_initial=1, has_duplicate=False, epsilon=1e-4)
        for p in [1, 2, 4, 3]:
            m.add(float(p))
        random.seed(7)
        u = [random.random() for _ in range(4000)]
        used, idx, w, _ = m.sample(4, 1, u)
        np.testing.assert_array_equal(idx, z["indices"][k])
        items = idx - (10 - 1)
        np.testing.assert_array_equal(items, z["item"][k])
        np.testing.assert_allclose(w, z["weights"][k], rtol=1e-15)
        np.testing.assert_allclose(w, z["true_weights"][k][items], rtol=1e-7)


def test_oracle_backup_restore_and_resize():
    m = OraclePER(7, alpha=0.5)
    rng = np.random.default_rng(0)
    for _ in range(11):
        m.add(float(rng.random()))
    mp, size, write, tree = m.get_state()
    m2 = OraclePER(7, alpha=0.5)
    m2.set_state(mp, size, write, tree)
    np.testing.assert_array_equal(m2.tree(), tree)
    assert m2.length() == 7 and m2.write == write
    # different capacity: proportional_memory.py:195-205
    m3 = OraclePER(5, alpha=0.5)
    m3.restore_resized(7, size, tree)
    assert m3.length() == 5
    assert m3.write == 7 % 5
    leaves_old = tree[6:]
    got = m3.tree()[4:]
    # slots 0,1 overwritten by old leaves 5,6
    np.testing.assert_array_equal(got, np.array([leaves_old[5], leaves_old[6], leaves_old[2], leaves_old[3], leaves_old[4]]))


def test_rankbased_oracle_matches_reference_trace():
    """rankbased_memory.py:42-58 replayed with the oracle restatement under the same numpy seed: indices and weights equal."""
    import os as _os
    import sys as _sys

    _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "oracle"))
    import hot_path_oracle as H

    z = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "rankbased_trace.npz"))
    cap, alpha = int(z["capacity"]), float(z["alpha"])
    pri = np.zeros(cap, np.float32)
    pos = size = k = 0
    np.random.seed(int(z["seed"]))
    for rnd in range(len(z["n_add"])):
        for _ in range(int(z["n_add"][rnd])):
            pri[pos] = z["add_priorities"][k]
            pos, size, k = (pos + 1) % cap, min(size + 1, cap), k + 1
        beta = min(1, float(z["beta_initial"]) + (1 - float(z["beta_initial"])) * (100 * rnd) / int(z["beta_steps"]))
        idx, w = H.rankbased_sample(pri[:size], 16, alpha, beta)
        np.testing.assert_array_equal(idx, z["indices"][rnd])
        np.testing.assert_array_equal(w, z["weights"][rnd])
        pri[idx] = z["new_priorities"][rnd]
    np.testing.assert_array_equal(pri, z["final_priorities"])
