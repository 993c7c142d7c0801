"""Rainbow: the captured update alone (run_updates, nothing beside it) -- under rocprofv3 --kernel-trace for tools/a57_trace_read.py-style timelines, or timed."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

cfg = RainbowDeviceConfig(n_envs=1024, memory_capacity=1_000_000, batch_size=32, seed=0)
eng = RainbowEngine(cfg, 0, episode_len=200, overlap=True)
eng.prefill()
for _ in range(4):
    eng.step(1)
eng.capture_graphs()
for _ in range(4):
    eng.step(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    eng.run_updates(1)
torch.cuda.synchronize()
print("update alone: %.4f ms" % (1e3 * (time.perf_counter() - t0) / n))
