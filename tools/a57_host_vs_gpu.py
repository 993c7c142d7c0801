"""Agent57_light lock-steps: when the HOST issued each step's first launch against when the GPU reached it (is the host ahead, or does a launch block it?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import agent57_light
from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine

rl = agent57_light.Config(batch_size=32)
rl.window_length = 4
rl.memory.capacity, rl.memory.warmup_size = 100_000, 40_000
rl.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1_000_000)
rl.input_block.image.set_dqn_block()
rl.hidden_block.set_dueling_network((512,))
rl.setup(srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=200))))
eng = Agent57LightFastEngine(rl, 1024, 0, episode_len=200, seed=0)
eng.prefill()
for _ in range(4):
    eng.step(1)
eng.capture_graphs()
for _ in range(8):
    eng.step(1)
torch.cuda.synchronize()
n = 24
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 1)]
host = []
ev[0].record()
t0 = time.perf_counter()
for k in range(n):
    a = time.perf_counter()
    eng.fork_learner(1)
    b = time.perf_counter()
    ev[2 * k + 1].record()
    eng.actor_step()
    ev[2 * k + 2].record()
    host.append((1e3 * (a - t0), 1e3 * (b - a), 1e3 * (time.perf_counter() - b)))
torch.cuda.synchronize()
for k in range(n):
    print("step %2d: host at %7.3f ms (graph launch %.3f, actor enqueue %.3f) | GPU: actors begin %7.3f end %7.3f" % (
        k, host[k][0], host[k][1], host[k][2], ev[0].elapsed_time(ev[2 * k + 1]), ev[0].elapsed_time(ev[2 * k + 2])))
