"""Agent57_light, E = 1024, captured graphs: two engine instances over N lock-steps -- finite losses throughout, identical trajectories (losses, train count, every
parameter bit for bit) at the end."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import agent57_light
from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400


def run():
    rl = agent57_light.Config(batch_size=32)
    rl.window_length = 4
    rl.memory.capacity, rl.memory.warmup_size = 100_000, 40_000
    rl.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1_000_000)
    rl.input_block.image.set_dqn_block()
    rl.hidden_block.set_dueling_network((512,))
    rl.setup(srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=200))))
    torch.manual_seed(0)
    eng = Agent57LightFastEngine(rl, 1024, 0, episode_len=200, seed=0)
    eng.prefill()
    for k in range(n):
        if k == 8:
            eng.capture_graphs()
        eng.step(1)
        if k % 100 == 99:
            info = eng.info()
            assert all(v == v and abs(v) < 1e6 for v in info.values() if isinstance(v, float)), info
    info = eng.info()
    flat = torch.cat([p.detach().reshape(-1) for net in eng.nets.values() for p in net.module.parameters()]).clone()
    eng.close()
    return info, flat


a, b = run(), run()
print(a[0])
assert a[0] == b[0], (a[0], b[0])
assert torch.equal(a[1], b[1])
print("two instances, %d lock-steps: identical" % n)
