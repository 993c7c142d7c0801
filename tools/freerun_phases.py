"""Phase times of the FREE-RUNNING fast lock-step (no synchronisation between lock-steps, as in bench.py): wall-clock stamp kernels on the actors' stream only
(nothing inside the update's graph), one slot per lock-step and phase, read back at the end."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=200_000, seed=0)
eng = RainbowEngine(cfg, 0, 200, overlap=True, actor_stream=os.environ.get("SRLX_ACTOR_STREAM") or None)  # (the tool's own command-line knob)
assert eng.fast
eng.prefill()
for _ in range(8):
    eng.step(1)
torch.cuda.synchronize()
BW = os.environ.get("SRLX_BACKWARD_STAMPS", "0") == "1"  # also stamps INSIDE the update's graph (srlx_qnet_set_stamp_buffer): they hold the last lock-step's times
GST = torch.zeros(32, dtype=torch.int64, device="cuda")
if BW:
    N.check(N.lib().srlx_qnet_set_stamp_buffer(eng.inf_online.h, N.tptr(GST)))
eng.capture_graphs()
for _ in range(50):
    eng.step(1)
torch.cuda.synchronize()
n, P = 400, 6
ST = torch.zeros(n * P, dtype=torch.int64, device="cuda")
lib = N.lib()
mark = lambda i: N.check(lib.srlx_debug_stamp(N.tptr(ST), i, N.torch_stream_ptr()))  # noqa: E731
first = False  # (the "actors first" issue order was measured and dropped in round 4)
bw_rows = []
for k in range(n):
    b = k * P
    if BW and k >= 50 and k % 25 == 0:  # read the graph's stamps of lock-step k - 1 against ITS start stamp
        torch.cuda.synchronize()
        g_ = GST.cpu().tolist()
        t0_ = int(ST[(k - 1) * P].item())
        bw_rows.append([(g_[i] - t0_) / 100.0 for i in range(16, 25)] + [(int(ST[(k - 1) * P + j].item()) - t0_) / 100.0 for j in (1, 2, 3, 4)] + [(g_[i] - t0_) / 100.0 for i in (10, 11, 12)])
    mark(b)
    if first:
        eng.fork_point()
        eng.actor_front()
        mark(b + 1)
        eng.fork_learner(1, marked=True)
    else:
        eng.fork_learner(1)
        mark(b + 5)  # the first launch on the actors' stream behind the update graph's launch
        eng.actor_front()
        mark(b + 1)
    eng.actor_commit_ring()
    mark(b + 2)
    eng.join_learner()
    mark(b + 3)
    eng.actor_commit_tree()
    eng.refresh_actor_copy()
    mark(b + 4)
torch.cuda.synchronize()
s = ST.cpu().view(n, P).tolist()
names = ["policy pass + environments end", "ring commit end", "join passed (update done)", "add end", "first launch behind the graph launch"]
for j, nm in enumerate(names, start=1):
    v = sorted((s[k][j] - s[k][0]) / 100.0 for k in range(50, n) if s[k][j])
    if v:
        print(f"{nm:40s} median {v[len(v) // 2]:7.1f} us   (10 % {v[len(v) // 10]:7.1f}, 90 % {v[9 * len(v) // 10]:7.1f})")
per = sorted((s[k + 1][0] - s[k][0]) / 100.0 for k in range(50, n - 1))
print(f"{'lock-step period':40s} median {per[len(per) // 2]:7.1f} us   (10 % {per[len(per) // 10]:7.1f}, 90 % {per[9 * len(per) // 10]:7.1f})")

if BW:
    nm = ["head backward (TD) end", "fc1 data gradient end", "conv3 data gradient + fold end", "conv2 data gradient + fold end", "conv1 weight gradient end",
          "[branch] priority write-back end", "[branch] conv3 weight gradient end", "[branch] conv2 weight gradient end", "[branch] fc1 weight gradient + Adam end",
          "ACTORS policy pass + environments end", "ACTORS ring commit end", "ACTORS join passed", "ACTORS add end",
          "online pass: convolutions end", "online pass: first dense layer end", "online pass: head end"]
    for j, x in enumerate(nm):
        v = sorted(r[j] for r in bw_rows)
        print(f"  in the free-running loop: {x:44s} median {v[len(v) // 2]:7.1f} us  ({v[0]:7.1f} .. {v[-1]:7.1f})")
