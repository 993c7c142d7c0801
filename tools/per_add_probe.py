"""Probe: PER add latency for n = 1024..8192 consecutive slots (the learner rank of an N-GPU run commits N*E per step)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd import _native as N
lib = N.lib(); dev = torch.device("cuda:0")
cap = 1_000_448
h = N.c_p(); N.check(lib.srlx_per_create(ctypes.byref(h), cap, 0.5, 0.4, 1e6, 1, 1e-4, 0))
pri = torch.rand(cap, dtype=torch.float64, device=dev)
N.check(lib.srlx_per_add(h, cap, N.tptr(pri), N.PRIO_F64, 1, None))
for n in (1024, 2048, 4096, 8192):
    mask = (torch.rand(n, device=dev) < 0.995).to(torch.uint8)
    def run(): N.check(lib.srlx_per_add(h, n, N.tptr(mask), N.PRIO_NONE_MASKED, 1, None))
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    print(f"n={n}: {a.elapsed_time(b)/20*1e3:.1f} us per add")
