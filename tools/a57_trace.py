"""Agent57_light lock-steps under rocprofv3 --kernel-trace: a short run whose kernel timeline tools/a57_trace_read.py turns into per-queue busy time and overlap."""
import os
import sys

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
eng.multi_trunk = os.environ.get("MULTI", "1") == "1"
eng.prefill()
for _ in range(4):
    eng.step(1)
eng.capture_graphs()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    eng.step(1)
torch.cuda.synchronize()
