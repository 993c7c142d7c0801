"""Agent57_light at E = 1024: the lock-step, its actors alone, its update alone (fork + join, nothing beside it), and the update's kernels by section (HIP events)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import agent57_light
from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rl = agent57_light.Config(batch_size=32)
rl.window_length = 4
rl.memory.capacity, rl.memory.warmup_size = 200_000, 80_000
rl.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1_000_000)
rl.input_block.image.set_dqn_block()
rl.hidden_block.set_dueling_network((512,))
rl.setup(srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=200))))
eng = Agent57LightFastEngine(rl, E, 0, episode_len=200, seed=0, fc1_neighbour=int(os.environ.get("FC1N", "4")))
eng.multi_trunk = os.environ.get("MULTI", "1") == "1"
eng.prefill()
for _ in range(16):
    eng.step(1)
eng.capture_graphs()
for _ in range(8):
    eng.step(1)


def timed(fn, reps=64):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def upd():
    eng.fork_learner(1)
    eng.join_learner()
    eng._flip()


print("lock-step %.3f ms | actors alone %.3f ms | update alone %.3f ms" % (timed(lambda: eng.step(1)), timed(lambda: eng.step(0)), timed(upd)))
# host time of one lock-step's launches (is the host the bound?)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(32):
    eng.step(1)
host = 1e3 * (time.perf_counter() - t0) / 32
torch.cuda.synchronize()
print("host time per lock-step (enqueue only) %.3f ms; graph nodes: %s" % (host, {k: None for k in eng._graphs}))
