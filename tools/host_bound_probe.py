"""Is the Rainbow lock-step loop bound by the host (launch calls) or by the GPU?  Times the host's enqueue loop and the same loop including the final
synchronisation, and the host cost of each piece of a lock-step (learner graph launch, actors' eager pass, select / commit graphs, weight refresh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=200_000, seed=0)
eng = RainbowEngine(cfg, 0, 200, overlap=True)
eng.prefill()
for _ in range(8):
    eng.step(1)
torch.cuda.synchronize()
eng.capture_graphs()
for _ in range(20):
    eng.step(1)
torch.cuda.synchronize()
n = 400
t0 = time.perf_counter()
for _ in range(n):
    eng.step(1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue loop {1e6 * (t1 - t0) / n:.1f} us per lock-step; with the final synchronisation {1e6 * (t2 - t0) / n:.1f} us per lock-step")
# host cost of the pieces (the GPU is drained before and after each batch of calls, so nothing blocks on a full queue)
def host_cost(fn, reps=200):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return 1e6 * dt / reps

print(f"  fork_learner (graph launch, 3 streams): {host_cost(lambda: (eng.fork_learner(1), eng.join_learner()), 100):.1f} us")
print(f"  actor network pass (eager: frame table, pack, conv, fc1, head): {host_cost(lambda: eng._actor_net(None, None)):.1f} us")
print(f"  select graph: {host_cost(lambda: eng._select_graph.replay()):.1f} us")
print(f"  commit graph: {host_cost(lambda: eng._commit_graph.replay()):.1f} us")
print(f"  refresh_actor_copy: {host_cost(eng.refresh_actor_copy):.1f} us")
