"""Where an Agent57_light engine lock-step goes: actor_step and learner_step timed separately (eager torch networks)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import agent57_light
from simple_distributed_rl_amd.device.agent57_light import Agent57LightEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rl = agent57_light.Config(batch_size=32)
rl.window_length = 4
rl.memory.capacity, rl.memory.warmup_size = 100_000, 1000
rl.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1_000_000)
rl.input_block.image.set_dqn_block()
rl.hidden_block.set_dueling_network((512,))
rl.setup(srl.make_env("SyntheticAtari-v0"))
eng = Agent57LightEngine(rl, E, 0)
for _ in range(30):
    eng.step(1)
torch.cuda.synchronize()


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


eager = timed(eng.learner_step, 20)
eng.capture_graphs()
print(f"learner_step eager {eager:.3f} ms -> graph {timed(eng.learner_step, 50):.3f} ms")
print(f"E={E}: actor_step {timed(eng.actor_step, 30):.3f} ms   learner_step {timed(eng.learner_step, 30):.3f} ms   policy_q {timed(eng.policy_q, 30):.3f} ms   "
      f"stack {timed(eng._stack, 30):.3f} ms")
