"""Host time of the two graph launches of a fast lock-step, GPU idle at the start of each (perf_counter around the enqueueing call, synchronised before it),
and when the first kernel behind each launch runs on the GPU (a wall-clock stamp kernel enqueued right after the call returns)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=200_000, seed=0)
eng = RainbowEngine(cfg, 0, 200, overlap=True)
assert eng.fast
eng.prefill()
for _ in range(8):
    eng.step(1)
torch.cuda.synchronize()
eng.capture_graphs()
for _ in range(50):
    eng.step(1)
torch.cuda.synchronize()
ST = torch.zeros(8, dtype=torch.int64, device="cuda")
mark = lambda i: N.check(N.lib().srlx_debug_stamp(N.tptr(ST), i, N.torch_stream_ptr()))  # noqa: E731
rows = []
for k in range(200):
    torch.cuda.synchronize()
    mark(0)
    t0 = time.perf_counter()
    eng.fork_learner(1)
    t1 = time.perf_counter()
    mark(1)
    eng.actor_front()
    t2 = time.perf_counter()
    mark(2)
    eng.actor_commit_ring()
    eng.join_learner()
    eng.actor_commit_tree()
    eng.refresh_actor_copy()
    torch.cuda.synchronize()
    s = ST.cpu().tolist()
    rows.append((1e6 * (t1 - t0), 1e6 * (t2 - t1), (s[1] - s[0]) / 100.0, (s[2] - s[0]) / 100.0))
rows = rows[20:]
med = lambda j: sorted(r[j] for r in rows)[len(rows) // 2]  # noqa: E731
print(f"host: update graph launch {med(0):6.1f} us | actors' launches {med(1):6.1f} us || GPU: first kernel behind the update launch at {med(2):6.1f} us, behind the actors' pass at {med(3):6.1f} us")
