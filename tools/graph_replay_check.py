"""Do captured-graph replays give the same training whether or not the host synchronises between them?  (ROCm 7.2 / torch 2.10: the PPO
engine's two large torch-captured graphs did not -- device/ppo.py:step -- so the other engines are checked the same way.)
Rainbow: bit-equal parameters / loss / priorities after 3000 lock-steps; Agent57_light: EIGHT engine instances (synchronised and not, alternating) land on ONE
trajectory -- with the learner's image blocks in libsrlx (round 3) nothing on the update path picks an algorithm by timing any more.  `python
tools/graph_replay_check.py a57` runs only that part."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import agent57_light
from simple_distributed_rl_amd.device.agent57_light import Agent57LightEngine
from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine


def rainbow(sync, n):
    cfg = RainbowDeviceConfig(n_envs=256, batch_size=32, memory_capacity=50_000, memory_warmup_size=1000, seed=0)
    eng = RainbowEngine(cfg, 0, 200, overlap=True)
    eng.prefill()
    for _ in range(5):
        eng.step(1)
    torch.cuda.synchronize()
    eng.capture_graphs()
    for _ in range(n):
        eng.step(1)
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (round(sum(float(p.detach().double().sum()) for p in eng.q_online.parameters()), 9), float(eng.loss), int(eng.train_count_dev),
            float(eng.priorities.double().sum()))


def a57(sync, n):
    torch.manual_seed(0)
    cfg = agent57_light.Config(batch_size=16, actor_num=4, target_model_update_interval=50, episodic_memory_capacity=64, ucb_window_size=6)
    cfg.window_length = 4
    cfg.memory.capacity, cfg.memory.warmup_size = 64 * 60, 256
    cfg.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1000)
    cfg.input_block.image.set_dqn_block()
    cfg.hidden_block.set_dueling_network((64,))
    cfg.setup(srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=30))))
    eng = Agent57LightEngine(cfg, 64, 0, episode_len=30, seed=3)
    eng.prefill()
    for _ in range(4):
        eng.step(1)
    torch.cuda.synchronize()
    eng.capture_graphs()
    for _ in range(n):
        eng.step(1)
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return eng.train_count, {k: round(v, 5) for k, v in eng.learner.losses().items()}


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] != "a57":
        a, b = rainbow(True, 3000), rainbow(False, 3000)
        print("rainbow, 3000 lock-steps:  synced", a, "\n                       unsynced", b, "\n   bit-equal:", a == b, flush=True)
    runs = []
    for k in range(8):
        sync = k % 2 == 0
        runs.append(a57(sync, 40))
        print("agent57_light, 40 lock-steps, instance %d:" % k, "  synced" if sync else "unsynced", runs[-1], flush=True)
    print("agent57_light: distinct trajectories over 8 engine instances:", len({repr(r) for r in runs}), flush=True)
