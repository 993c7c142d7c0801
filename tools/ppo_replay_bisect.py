"""Which ingredient makes back-to-back replays of the PPO engine's two captured graphs diverge (device/ppo.py:step)?  Every variant runs in its own
process: 2 eager iterations, capture, then N replayed iterations at E environments; prints the final losses and whether the parameters are finite.
  python tools/ppo_replay_bisect.py            # the matrix
  python tools/ppo_replay_bisect.py one <E> <N>  # one run with the environment's SRLX_PPO_* switches"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch

    from simple_distributed_rl_amd.device.ppo import PPODeviceConfig, PPOEngine

    E, n = int(sys.argv[2]), int(sys.argv[3])
    eng = PPOEngine(PPODeviceConfig(n_envs=E, seed=0), 0)
    for _ in range(2):
        eng.step()
    if os.environ.get("SRLX_PPO_EAGER", "0") != "1":
        eng.capture_graphs()
    eng.step()
    torch.cuda.synchronize()
    for _ in range(n):
        eng.step()
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(p).all()) for p in eng.net.parameters())
    print(json.dumps({"finite": finite, **{k: (round(v, 6) if v == v else None) for k, v in eng.info().items()}}))
else:
    E, n = 4096, 40
    variants = [("srlx permutation kernel inside the graph, no host wait (shipped)", {}),
                ("srlx permutation kernel inside the graph, stream sync per iteration", {"SRLX_PPO_SYNC": "stream"}),
                ("torch.randperm drawn eagerly between replays, no host wait", {"SRLX_PPO_PERM": "eager"}),
                ("torch.randperm drawn eagerly between replays, stream sync per iteration", {"SRLX_PPO_PERM": "eager", "SRLX_PPO_SYNC": "stream"}),
                ("randperm inside the graph, stream sync per iteration", {"SRLX_PPO_PERM": "in_graph", "SRLX_PPO_SYNC": "stream"}),
                ("randperm inside the graph, no host wait", {"SRLX_PPO_PERM": "in_graph"}),
                ("randperm inside the graph, event.synchronize per iteration", {"SRLX_PPO_PERM": "in_graph", "SRLX_PPO_SYNC": "event"}),
                ("randperm inside the graph, stream sync every 4th", {"SRLX_PPO_PERM": "in_graph", "SRLX_PPO_SYNC": "every4"}),
                ("fixed permutations, no host wait", {"SRLX_PPO_PERM": "fixed"}),
                ("fixed permutations, stream sync per iteration", {"SRLX_PPO_PERM": "fixed", "SRLX_PPO_SYNC": "stream"}),
                ("no graphs at all (eager), permutations drawn eagerly", {"SRLX_PPO_EAGER": "1"})]
    for label, env in variants:
        r = subprocess.run([sys.executable, __file__, "one", str(E), str(n)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print("%-74s %s" % (label, line[-1] if line else "FAILED: " + r.stderr[-300:].replace("\n", " | ")), flush=True)
