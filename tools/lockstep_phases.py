"""Where a Rainbow lock-step's time goes on the GPU, untraced: events on the main stream (actors) and on the learner's stream, averaged over many lock-steps.
  t0 fork point (start) | actor_front end (network pass + selection + environments) | learner end | join + commit end | refresh end (= next t0)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=1_000_000, seed=0)
eng = RainbowEngine(cfg, 0, 200, overlap=True)
eng.prefill()
for _ in range(8):
    eng.step(1)
torch.cuda.synchronize()
eng.capture_graphs()
for _ in range(50):
    eng.step(1)
torch.cuda.synchronize()
n = 300
E = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
if eng.fast:  # the round-4 lock-step: network pass + selection | environments | ring commit (before the join) | join + PER add
    ev = [[E() for _ in range(7)] for _ in range(n)]
    main = torch.cuda.current_stream()
    for k in range(n):
        e = ev[k]
        e[0].record(main)
        eng.fork_learner(1)
        e[5].record(eng.s_learner)
        e[6].record(main)  # reached as soon as the host is back from the graph launch (the main stream is idle here)
        off = eng.replay.frame_table_current()
        eng.inf_actor.forward_u8_policy(eng.replay.obs_base, off, eng.eps, cfg.seed ^ 0xAC7, eng.policy_counter, eng.actions)
        e[1].record(main)
        eng.env.step(eng.actions)
        e[2].record(main)
        eng.actor_commit_ring()
        e[3].record(main)
        eng.join_learner()
        eng.actor_commit_tree()
        eng.refresh_actor_copy()
        e[4].record(main)
    torch.cuda.synchronize()
    for nm, i in zip(["policy pass START", "policy pass end", "environments end", "ring commit end", "add end (after join)", "learner end"], [6, 1, 2, 3, 4, 5]):
        v = sorted(ev[k][0].elapsed_time(ev[k][i]) for k in range(20, n))
        print(f"{nm:26s} median {1e3 * v[len(v) // 2]:7.1f} us   (10 % {1e3 * v[len(v) // 10]:7.1f}, 90 % {1e3 * v[9 * len(v) // 10]:7.1f})")
    sys.exit(0)
ev = [[E() for _ in range(6)] for _ in range(n)]
main = torch.cuda.current_stream()
for k in range(n):
    e = ev[k]
    e[0].record(main)
    eng.fork_learner(1)
    e[5].record(eng.s_learner)  # learner end (recorded on its stream after the update)
    q = eng._actor_net(None, None)
    e[1].record(main)  # network pass end
    eng._select_graph.replay()
    e[2].record(main)  # actor front end
    eng.join_learner()
    eng.actor_commit()
    e[3].record(main)
    eng.refresh_actor_copy()
    e[4].record(main)
torch.cuda.synchronize()
names = ["network pass end", "selection + env end", "commit end (after join)", "refresh end", "learner end"]
idx = [1, 2, 3, 4, 5]
for nm, i in zip(names, idx):
    v = sorted(ev[k][0].elapsed_time(ev[k][i]) for k in range(20, n))
    print(f"{nm:26s} median {1e3 * v[len(v) // 2]:7.1f} us   (10 % {1e3 * v[len(v) // 10]:7.1f}, 90 % {1e3 * v[9 * len(v) // 10]:7.1f})")
