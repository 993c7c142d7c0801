"""Host-side cost of one fast lock-step, call by call (perf_counter around each enqueueing call, GPU kept busy), and the same loop's GPU period."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

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
names = ["fork_learner (graph launch)", "actor_front (4 launches)", "commit_ring", "join", "commit_tree (add)", "refresh (select)"]
acc = [0.0] * len(names)
n = 400
t_all = time.perf_counter()
for _ in range(n):
    t = [time.perf_counter()]
    eng.fork_learner(1); t.append(time.perf_counter())
    eng.actor_front(); t.append(time.perf_counter())
    eng.actor_commit_ring(); t.append(time.perf_counter())
    eng.join_learner(); t.append(time.perf_counter())
    eng.actor_commit_tree(); t.append(time.perf_counter())
    eng.refresh_actor_copy(); t.append(time.perf_counter())
    for k in range(len(names)):
        acc[k] += t[k + 1] - t[k]
host_total = time.perf_counter() - t_all
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
for nm, a in zip(names, acc):
    print(f"{nm:32s} {1e6 * a / n:7.1f} us")
print(f"host issue time per lock-step {1e6 * host_total / n:7.1f} us; wall per lock-step {1e6 * wall / n:7.1f} us")
