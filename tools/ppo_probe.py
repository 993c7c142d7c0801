"""Probe: PPO engine learning curve / throughput on the Pendulum-shaped workload (eager vs HIP graphs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine
for name, E, graphs in [("eager-1024", 1024, False), ("graphs-1024", 1024, True), ("graphs-4096", 4096, True)]:
    cfg = PPODeviceConfig(n_envs=E, horizon=50, seed=1)
    eng = PPOEngine(cfg, 0)
    for _ in range(3):
        eng.step()
    if graphs:
        eng.capture_graphs()
    eng.pop_mean_episode_return()
    torch.cuda.synchronize(); t0 = time.time()
    curve = []
    iters = 200
    for it in range(iters):
        eng.step()
        if (it + 1) % 40 == 0:
            curve.append(round(eng.pop_mean_episode_return()))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(name, curve, f"{iters*cfg.n_envs*cfg.horizon/dt/1e6:.2f} M env-steps/s", eng.info())
