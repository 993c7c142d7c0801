import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine
eng = PPOEngine(PPODeviceConfig(n_envs=4096, seed=0), 0)
eng.rollout(); torch.cuda.synchronize()
import contextlib, io
eng.cfg.epochs = 1; eng.cfg.minibatches = 4
eng.update(); torch.cuda.synchronize()
