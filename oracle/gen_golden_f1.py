#!/usr/bin/env python3
"""oracle/gen_golden_f1.py -- TEST INFRASTRUCTURE ONLY.  Wire-format fixtures written BY THE REFERENCE:

    f1_memory_<name>.npz   a Rainbow-configured RLPriorityReplayBuffer (Python ProportionalMemory inside) after a scripted
                           add / sample / update history:
                             backup_file   the bytes of `memory.save(path, compress=True)`  (srl/base/rl/memory.py:137-143 ->
                                           srl/utils/common.py:117-134: lzma container of a pickle of call_backup())
                             backup_plain  the bytes of `memory.save(path, compress=False)` (plain pickle)
                             final_tree / final_size / final_write / final_max_priority: the tree the file describes
                             after_*       what the reference's memory returns for `sample()` under random.seed(77) right after
                                           it was restored from that file (indices, weights, the first field of every item)
    f1_parameter_dqn.npz   the bytes of `parameter.save(path)` of a small DQN network + its output on a probe batch

Only data is stored: file bytes the reference wrote and arrays it returned.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_f1.py
"""
import os
import random
import sys
import tempfile

import numpy as np

REF = os.environ.get("SRL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _rainbow_memory(compress, capacity=300, warmup=16, batch=8):
    from srl.algorithms import rainbow

    cfg = rainbow.Config(batch_size=batch)
    cfg.memory.capacity, cfg.memory.warmup_size, cfg.memory.compress = capacity, warmup, compress
    cfg.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1000)
    return cfg


class _Cfg:
    """What RLMemory needs of an RLConfig (tests/quick/rl/memories/test_rl_memories.py uses a dummy config the same way)."""

    def __init__(self, base):
        self.memory, self.batch_size = base.memory, base.batch_size

    def get_dtype(self, fw):
        return np.float32


def gen_memory(name, compress):
    from srl.rl.memories.priority_replay_buffer import RLPriorityReplayBuffer

    base = _rainbow_memory(compress)
    mem = RLPriorityReplayBuffer(_Cfg(base))
    rng = np.random.default_rng(4)
    random.seed(5)
    for i in range(420):  # wraps the 300-slot ring
        pri = None if i % 3 == 0 else float(rng.random() * 2)
        mem.add((i, float(rng.standard_normal()), [i % 7, i % 5]), pri)
        if i >= 16 and i % 5 == 0:
            batches, w, args = mem.sample()
            mem.update(args, np.abs(rng.standard_normal(len(args))).astype(np.float32), i)
    tree = np.asarray(mem.memory.tree.tree, np.float64)
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "m.dat"), os.path.join(d, "m_plain.dat")
        mem.save(p1, compress=True)
        mem.save(p2, compress=False)
        blob, plain = open(p1, "rb").read(), open(p2, "rb").read()
        other = RLPriorityReplayBuffer(_Cfg(_rainbow_memory(compress)))
        other.load(p1)
    random.seed(77)
    batches, w, args = other.sample(step=123)
    np.savez_compressed(
        os.path.join(OUT, f"f1_memory_{name}.npz"),
        backup_file=np.frombuffer(blob, np.uint8), backup_plain=np.frombuffer(plain, np.uint8), compress=np.bool_(compress),
        capacity=np.int64(300), warmup=np.int64(16), batch=np.int64(8), alpha=0.6, beta_initial=0.4, beta_steps=1000.0,
        final_tree=tree, final_size=np.int64(mem.memory.size), final_write=np.int64(mem.memory.tree.write), final_max_priority=np.float64(mem.memory.max_priority),
        after_indices=np.asarray(args, np.int64), after_weights=np.asarray(w, np.float64), after_first_field=np.asarray([b[0] for b in batches], np.int64),
        after_step=np.int64(123), after_seed=np.int64(77), memory_step=np.int64(mem.step),
    )
    print(name, "ok:", len(blob), "bytes,", "size", mem.memory.size)


def gen_parameter():
    import torch

    import srl
    from srl.algorithms import dqn

    cfg = dqn.Config()
    cfg.hidden_block.set((16, 8))
    cfg.set_torch()
    runner = srl.Runner("Grid", cfg)
    runner.set_device("CPU")
    runner.set_seed(2)
    param = runner.make_parameter()
    probe = np.arange(2 * 2, dtype=np.float32).reshape(2, 2) / 4
    with torch.no_grad():
        q = param.q_online(torch.from_numpy(probe)).numpy()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "p.dat")
        param.save(p)
        blob = open(p, "rb").read()
    np.savez_compressed(os.path.join(OUT, "f1_parameter_dqn.npz"), parameter_file=np.frombuffer(blob, np.uint8), probe=probe, q=q,
                        keys=np.array(list(param.q_online.state_dict().keys())))
    print("parameter ok:", len(blob), "bytes")


if __name__ == "__main__":
    gen_memory("plain_items", False)
    gen_memory("compressed_items", True)
    gen_parameter()
