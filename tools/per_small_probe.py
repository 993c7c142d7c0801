"""Latency of the learner-sized PER calls (device pointers, back to back on one stream): sample / update at 32 and 64, add at 1024 and 7168."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd import _native as N
lib = N.lib(); dev = torch.device("cuda:0"); cap = 1_000_000
h = N.c_p(); N.check(lib.srlx_per_create(ctypes.byref(h), cap, 0.5, 0.4, 1e6, 1, 1e-4, 0))
g = torch.Generator(device="cuda").manual_seed(1)
N.check(lib.srlx_per_add(h, cap, N.tptr(torch.rand(cap, dtype=torch.float64, device=dev, generator=g)), N.PRIO_F64, 1, None))
def timed(fn, reps=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
out = []
for n in (32, 64):
    idx = torch.randint(0, cap, (n,), device=dev, generator=g) + cap - 1
    pri = torch.rand(n, dtype=torch.float32, device=dev, generator=g)
    out.append(f"update_{n} {timed(lambda: N.check(lib.srlx_per_update(h, n, N.tptr(idx), N.tptr(pri), N.PRIO_F32, 1, None))):.2f} us")
for n in (1024, 7168):
    m = torch.ones(n, dtype=torch.uint8, device=dev)
    out.append(f"add_{n} {timed(lambda: N.check(lib.srlx_per_add(h, n, N.tptr(m), N.PRIO_NONE_MASKED, 1, None)), 100):.2f} us")
print("  ".join(out))
