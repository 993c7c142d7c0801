"""CPU tests of the NGU / Agent57_light row: the oracle restatements against vectors recorded
from the imported reference (oracle/gen_golden_agent57.py), the host-side meta-controller against its recorded
trace, the plugin's registration and reference-compatible state_dict keys."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hot_path_oracle as H  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_oracle_episodic_reward_matches_reference():
    """agent57_light.py:473-513 on scripted embedding sequences (revisits, near-duplicates, a full deque,
    k > len, all-duplicate memory): bit-equal with the per-entry dot order, 1e-6 with the vectorised one."""
    z = np.load(os.path.join(GOLDEN, "ngu_episodic.npz"))
    for name in z["names"]:
        emb, want = z[name + ".emb"], z[name + ".reward"]
        mem = H.EpisodicMemoryOracle(int(z[name + ".capacity"]), int(z[name + ".k"]), float(z["epsilon"]), float(z["cluster_distance"]), float(z["pseudo_counts"]))
        got = np.array([mem.step(e) for e in emb], np.float64)
        np.testing.assert_array_equal(got, want, err_msg=str(name))
        mem.reset()
        got = np.array([mem.step(e, exact_dot=False) for e in emb], np.float64)
        np.testing.assert_allclose(got, want, rtol=1e-6, err_msg=str(name))


def test_oracle_lifelong_reward_matches_reference():
    z = np.load(os.path.join(GOLDEN, "ngu_lifelong.npz"))
    got = H.ngu_lifelong_reward(z["target"], z["train"], float(z["lifelong_max"]))
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got.astype(np.float64), z["reward"])


@pytest.mark.parametrize("name", ["double", "single_inv", "double_rescale_inv"])
def test_oracle_agent57_target_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, f"agent57_light_target_{name}.npz"))
    got = H.agent57_target(z["q_online"], z["q_target"], z["rewards"], z["dones"], z["discount"], z["invalid"], bool(z["double_dqn"]), bool(z["rescale"]))
    np.testing.assert_array_equal(got, z["target"])


def test_oracle_priority_matches_reference_train_step():
    from simple_distributed_rl_amd.rl import functions as F

    z = np.load(os.path.join(GOLDEN, "train_step_agent57_light.npz"))
    beta = np.array(F.create_beta_list(int(z["actor_num"])), np.float32)[z["actor_idx"]]
    np.testing.assert_array_equal(H.agent57_priority(z["td_ext"], z["td_int"], beta), z["priorities"])


def test_ucb_meta_controller_matches_reference_trace():
    """agent57_light.py:317-353: the same seeded `random` stream and episode returns give the same actor sequence."""
    from simple_distributed_rl_amd.algorithms.agent57_light import UcbMetaController

    z = np.load(os.path.join(GOLDEN, "agent57_ucb.npz"))
    random.seed(int(z["seed"]))
    ucb = UcbMetaController(int(z["actor_num"]), int(z["window"]), float(z["ucb_epsilon"]), float(z["ucb_beta"]))
    last, got = 0.0, []
    for r in z["episode_rewards"]:
        got.append(ucb.next_actor(last))
        last = float(r)
    np.testing.assert_array_equal(np.array(got), z["actor_index"])


def test_plugin_registered_with_reference_compatible_networks():
    """"Agent57_light:torch" resolves to this build's classes; the five networks carry the reference's
    state_dict keys and shapes (so a reference checkpoint loads, model_torch.py:139-156)."""
    import simple_distributed_rl_amd as srl
    from simple_distributed_rl_amd.algorithms import agent57_light
    from simple_distributed_rl_amd.base.env import registration
    from test_plugin_surface import TinyImg  # noqa: F401

    registration.register("TinyImg", "test_plugin_surface:TinyImg", check_duplicate=False)
    z = np.load(os.path.join(GOLDEN, "train_step_agent57_light.npz"))
    rl = agent57_light.Config(batch_size=16, actor_num=int(z["actor_num"]))
    rl.window_length = 4
    rl.hidden_block.set_dueling_network((32,))
    assert rl.get_name() == "Agent57_light"
    runner = srl.Runner(srl.EnvConfig("TinyImg"), rl)
    param = runner.make_parameter()
    assert type(param).__module__ == agent57_light.__name__
    nets = dict(q_ext=param.q_ext_online, q_int=param.q_int_online, emb=param.emb_network, lifelong_target=param.lifelong_target, lifelong_train=param.lifelong_train)
    for name, net in nets.items():
        want = {k[len("before.") + len(name) + 1:]: z[k].shape for k in z.files if k.startswith(f"before.{name}.")}
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert got == want, name
    data = param.backup(serialized=True)
    assert len(data) == 5 and all(v.device.type == "cpu" for sd in data for v in sd.values())
    param.restore(data, from_serialized=True)
