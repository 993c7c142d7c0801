"""oracle/gen_golden_uniform.py -- TEST INFRASTRUCTURE ONLY.  A seeded add / sample trace of the reference's uniform ReplayBuffer
(srl/rl/memories/priority_memories/replay_buffer.py:10-55: ring list + `random.sample(self.memory, batch_size)`, without replacement).

Run here, where /root/reference is importable:  PYTHONPATH=/root/reference python oracle/gen_golden_uniform.py
Items are their own insertion numbers, so the recorded batches say WHICH stored items the reference drew; `probe` is one `random.random()` after
every sample call (the position of Python's generator: a replacement must consume the stream exactly like the reference)."""
import os
import random

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    from srl.rl.memories.priority_memories.replay_buffer import ReplayBuffer

    capacity, B, seed = 300, 32, 20260929
    mem = ReplayBuffer(capacity)
    random.seed(seed)
    added, batches, at, probes, weights_ok = 0, [], [], [], True
    for upto, n_samples in ((40, 2), (299, 3), (300, 2), (451, 4), (1000, 3)):  # partly filled, one short of full, full, wrapped, wrapped several times
        while added < upto:
            mem.add(added, None)
            added += 1
        for _ in range(n_samples):
            items, w, upd = mem.sample(B, step=added)
            weights_ok &= (w == [1.0] * B) and upd == []
            batches.append(items)
            at.append(added)
            probes.append(random.random())
    assert weights_ok
    np.savez_compressed(os.path.join(OUT, "uniform_replay_trace.npz"), capacity=np.array(capacity), batch_size=np.array(B), seed=np.array(seed),
                        added_before_sample=np.array(at), sampled_items=np.array(batches, dtype=np.int64), probe=np.array(probes))
    print("uniform replay trace:", len(batches), "batches; first", batches[0][:6])


if __name__ == "__main__":
    main()
