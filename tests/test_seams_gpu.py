"""GPU tests of the drop-in seams: the memory is built the way a reference user would select it --
`cfg.memory.set_custom(entry_point, kwargs)` (srl/rl/memories/priority_replay_buffer.py:111-117,149-152),
`set_proportional_cpp(...)` (:63-81, the pybind11 twin's constructor surface) and `set_proportional(...)` -- and must replay a
trace recorded from the reference bit-exactly; memory and parameter FILES written by the reference load into this build."""
import lzma
import os
import pickle
import random
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_bindings import iter_trace  # noqa: E402

ENTRY = "simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory:ProportionalMemory"


def _config(seam, z):
    from simple_distributed_rl_amd.rl.memories.priority_replay_buffer import PriorityReplayBufferConfig

    kw = dict(alpha=float(z["alpha"]), beta_initial=float(z["beta_initial"]), beta_steps=int(z["beta_steps"]), has_duplicate=bool(z["has_duplicate"]),
              epsilon=float(z["epsilon"]))
    cfg = PriorityReplayBufferConfig(int(z["capacity"]), 1, False)
    if seam == "set_custom":
        cfg.set_custom(ENTRY, kw)
    elif seam == "set_proportional_cpp":
        cfg.set_proportional_cpp(**kw)
    else:
        cfg.set_proportional(**kw)
    return cfg


@pytest.mark.parametrize("seam", ["set_custom", "set_proportional_cpp", "set_proportional"])
@pytest.mark.parametrize("trace", ["per_trace_rainbow_cap3000", "per_trace_small_nodup", "per_trace_dupupdate_cap257"])
def test_memory_selected_through_a_reference_seam_replays_the_reference_trace(seam, trace):
    from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory
    from simple_distributed_rl_amd.rl.memories.priority_replay_buffer import PriorityReplayBuffer

    z = np.load(os.path.join(GOLDEN, trace + ".npz"))
    buf = PriorityReplayBuffer(_config(seam, z), 1)
    mem = buf.memory
    assert isinstance(mem, ProportionalMemory) and mem.capacity == int(z["capacity"])
    random.seed(int(z["seed"]))
    item = 0
    for kind, p in iter_trace(z):
        if kind == "add":
            buf.add(("item", item), p["priority"])  # through the wrapper, as Worker.on_step does (rainbow.py:400)
            item += 1
        elif kind == "sample":
            batches, w, idx = mem.sample(p["batch_size"], p["step"])
            assert idx == p["indices"].tolist()
            np.testing.assert_allclose(w, p["weights"], rtol=1e-13, atol=0)
            assert all(b[0] == "item" for b in batches)
        else:
            mem.update(p["indices"].tolist(), p["priorities"])
    np.testing.assert_array_equal(mem.tree_array(), z["final_tree"])
    assert mem.length() == int(z["final_size"]) and mem.max_priority == float(z["final_max_priority"])


def test_pybind11_twin_constructor_signature():
    """b2: `ProportionalMemory(capacity, alpha=.6, beta_initial=.4, beta_steps=1e6, has_duplicate=True, epsilon=1e-4)` with
    `clear/length/add(batch, priority=None, restore_skip)/sample/update/backup/restore` (cpp_module/src/proportional_memory.cpp:250-275)."""
    from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory

    m = ProportionalMemory(10, 0.8, 1, 10)  # the call the survey made against the compiled reference module
    for i in range(10):
        m.add(i, float(i + 1))
    b, w, a = m.sample(4, 1)
    assert len(b) == len(a) == 4 and len(w) == 4
    m.update(a, np.ones(4, np.float32))
    bk = m.backup()
    m.clear()
    assert m.length() == 0
    m.restore(bk)
    assert m.length() == 10


def _rl_cfg(z):
    from simple_distributed_rl_amd.base.rl.config import DummyRLConfig
    from simple_distributed_rl_amd.rl.memories.priority_replay_buffer import PriorityReplayBufferConfig

    cfg = DummyRLConfig()
    cfg.batch_size = int(z["batch"])
    cfg.memory = PriorityReplayBufferConfig(int(z["capacity"]), int(z["warmup"]), bool(z["compress"]))
    cfg.memory.set_proportional(alpha=float(z["alpha"]), beta_initial=float(z["beta_initial"]), beta_steps=int(z["beta_steps"]))
    return cfg


@pytest.mark.parametrize("name", ["plain_items", "compressed_items"])
@pytest.mark.parametrize("which", ["backup_file", "backup_plain"])
def test_memory_file_written_by_the_reference_loads_into_the_device_tree(name, which, tmp_path):
    """f1: `memory.save()` of the reference (lzma pickle of [[capacity, max_priority, size, write, tree list, data list], None],
    proportional_memory.py:179-187 / priority_replay_buffer.py:252-253 / common.py:117-134) -> `memory.load()` here: the HBM tree
    equals the tree the file describes, and the next seeded sample() equals what the reference's own restored memory returned."""
    from simple_distributed_rl_amd.rl.memories.priority_replay_buffer import RLPriorityReplayBuffer

    z = np.load(os.path.join(GOLDEN, f"f1_memory_{name}.npz"))
    path = str(tmp_path / "ref_memory.dat")
    open(path, "wb").write(z[which].tobytes())
    mem = RLPriorityReplayBuffer(_rl_cfg(z))
    mem.load(path)
    np.testing.assert_array_equal(mem.memory.tree_array(), z["final_tree"])
    assert mem.length() == int(z["final_size"]) and mem.memory.max_priority == float(z["final_max_priority"])
    random.seed(int(z["after_seed"]))
    batches, w, args = mem.sample(step=int(z["after_step"]))
    assert list(args) == z["after_indices"].tolist()
    np.testing.assert_allclose(w, z["after_weights"].astype(np.float32), rtol=1e-6)
    assert [b[0] for b in batches] == z["after_first_field"].tolist()
    # ... and the other direction: what this build writes is the same container with the same list layout
    out = str(tmp_path / "ours.dat")
    mem.save(out, compress=True)
    raw = open(out, "rb").read()
    assert raw[:6] == bytes.fromhex("fd377a585a00")
    theirs = pickle.loads(lzma.decompress(z["backup_file"].tobytes()))
    ours = pickle.loads(lzma.decompress(raw))
    assert len(ours) == len(theirs) == 2 and ours[1] is None and theirs[1] is None
    assert [type(x) for x in ours[0]] == [type(x) for x in theirs[0]]
    assert ours[0][:4] == theirs[0][:4] and ours[0][4] == theirs[0][4]
    assert [pickle.dumps(a) == pickle.dumps(b) for a, b in zip(ours[0][5], theirs[0][5])].count(False) == 0


def test_parameter_file_written_by_the_reference_loads(tmp_path):
    """f1: `parameter.save()` of the reference (pickled state_dict, srl/base/rl/parameter.py:38-51) -> `runner.load_parameter()`."""
    import torch

    import simple_distributed_rl_amd as srl
    from simple_distributed_rl_amd.algorithms import dqn

    z = np.load(os.path.join(GOLDEN, "f1_parameter_dqn.npz"))
    path = str(tmp_path / "ref_param.dat")
    open(path, "wb").write(z["parameter_file"].tobytes())
    cfg = dqn.Config()
    cfg.hidden_block.set((16, 8))
    runner = srl.Runner("Grid", cfg)
    runner.set_device("cuda:0")
    runner.load_parameter(path)
    p = runner.parameter
    assert list(p.q_online.state_dict().keys()) == z["keys"].tolist()
    with torch.no_grad():
        q = p.q_online(torch.as_tensor(z["probe"], device=p.device)).cpu().numpy()
    np.testing.assert_allclose(q, z["q"], rtol=1e-5, atol=1e-6)
