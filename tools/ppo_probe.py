"""Probe: PPO engine learning curve / throughput on the Pendulum-shaped workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine
for name, kw in [
    ("ref-defaults", dict()),
    ("return-noVclip", dict(v_target="return", enable_value_clip=False, discount=0.95, gae_discount=0.9, lr=1e-3, entropy_weight=0.0)),
    ("return-noVclip-g99", dict(v_target="return", enable_value_clip=False, discount=0.99, gae_discount=0.95, lr=1e-3, entropy_weight=0.0, epochs=8)),
]:
    cfg = PPODeviceConfig(n_envs=1024, horizon=50, seed=1, **kw)
    eng = PPOEngine(cfg, 0)
    torch.cuda.synchronize(); t0 = time.time()
    curve = []
    for it in range(200):
        eng.step()
        if (it + 1) % 20 == 0:
            curve.append(round(eng.pop_mean_episode_return()))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(name, curve, f"{200*cfg.n_envs*cfg.horizon/dt/1e6:.2f} M env-steps/s", eng.info())
