"""Pins oracle/hot_path_oracle.py (numpy restatements of the reference's host arithmetic) against
vectors recorded from the imported reference (oracle/gen_golden_algo.py).  CPU only."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hot_path_oracle as H  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_rescaling_functions():
    z = np.load(os.path.join(GOLDEN, "functions.npz"))
    np.testing.assert_array_equal(H.rescaling(z["x32"]), z["rescaling32"])
    np.testing.assert_array_equal(H.inverse_rescaling(z["x32"]), z["inverse_rescaling32"])
    np.testing.assert_array_equal(H.rescaling(z["x64"]), z["rescaling64"])
    np.testing.assert_array_equal(H.inverse_rescaling(z["x64"]), z["inverse_rescaling64"])
    assert H.rescaling(z["x32"]).dtype == np.float32


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "target_q_*.npz"))), ids=lambda p: os.path.basename(p)[9:-4])
def test_nstep_target_matches_reference(path):
    """rainbow.py:185-287 on recorded network outputs: bit-equal float32 targets."""
    z = np.load(path)
    got = H.nstep_target(
        z["q_online"], z["q_target"], z["actions"], z["reward"], z["done"], z["invalid"],
        float(z["discount"]), float(z["retrace_h"]), bool(z["double_dqn"]), bool(z["rescale"]),
    )
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, z["target_q"])


def test_train_step_arithmetic_matches_reference():
    """rainbow/model_torch.py:103-114 on recorded q rows: loss and d loss/d q to 1e-6, priorities exact."""
    z = np.load(os.path.join(GOLDEN, "train_step_rainbow.npz"))
    a0 = z["actions"][:, 0]
    loss, grad, pri = H.huber_loss_grad_priority(z["q_all"], a0, z["target_q"], z["weights"])
    np.testing.assert_allclose(loss, z["loss"], rtol=1e-6)
    np.testing.assert_allclose(grad, z["grad_q"], rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(pri, z["priorities"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "dqn_target_*.npz"))), ids=lambda p: os.path.basename(p)[11:-4])
def test_dqn_targets_match_reference(path):
    z = np.load(path)
    kw = dict(discount=float(z["discount"]), double_dqn=bool(z["double_dqn"]), rescale=bool(z["rescale"]))
    got = H.dqn_target(z["dqn_q_online"], z["dqn_q_target"], z["reward"], z["undone"], z["invalid"], f64_accum=True, **kw)
    np.testing.assert_array_equal(got, z["dqn_target"])
    got = H.dqn_target(z["rb_q_online"], z["rb_q_target"], z["reward"], z["undone"], z["invalid"], f64_accum=False, **kw)
    np.testing.assert_array_equal(got, z["rb_target"])


def replay_log_into_store(z, store):
    """Feeds a recorded single-env trajectory (TinyImageEnv log) into a lock-step store model."""
    frames, actions, rewards = z["frames"], z["actions"], z["rewards"]
    term, trunc = z["terminated"], z["truncated"]
    store.reset_all(frames[0][None])
    for i in range(1, len(frames)):
        if actions[i] < 0:  # reset record: the env's reset step
            store.commit_step([0], [0.0], [0], [0], frames[i][None])
        else:
            store.commit_step([actions[i]], [rewards[i]], [term[i]], [term[i] or trunc[i]], frames[i][None])


@pytest.mark.parametrize("name", ["terminated", "truncated"])
def test_store_model_reproduces_reference_items(name):
    """The n-step items the reference's Rainbow worker emitted (frame stacking with zero history,
    n-step assembly, terminal padding, reward clip; rainbow.py:331-400, worker_run.py:310-358) equal
    what the lock-step store model gathers for the same positions."""
    z = np.load(os.path.join(GOLDEN, f"rollout_items_{name}.npz"))
    store = H.StoreOracle(1, 128, 64, 4, 3, 4, True, 123)
    replay_log_into_store(z, store)
    valid = [q for q in range(store.pos) if not (store.flags[0, q] & store.INVALID)]
    n_items = z["item_obs"].shape[0]
    assert n_items <= len(valid)
    for i in range(n_items):
        obs, act, rew, ter, jd = store.gather_item(0, valid[i])
        np.testing.assert_array_equal(obs, z["item_obs"][i])
        np.testing.assert_array_equal(rew, z["item_rewards"][i])
        np.testing.assert_array_equal(ter, z["item_terminated"][i])
        real = min(jd + 1, 3)
        np.testing.assert_array_equal(act[:real], z["item_actions"][i][:real])  # padded actions are random in both


def test_rng_is_uniform_and_keyed():
    u = H.rng_uniform(7, 3, 200_000)
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 5e-3
    assert not np.array_equal(u[:100], H.rng_uniform(7, 4, 100))
    assert not np.array_equal(u[:100], H.rng_uniform(8, 3, 100))
