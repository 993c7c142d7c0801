"""add timings (n consecutive ring slots, masked max-priority adds as the engines issue them) for the library selected by SRLX_LIB."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd import _native as N
lib = N.lib()
cap = 1_003_520
h = N.c_p(); N.check(lib.srlx_per_create(ctypes.byref(h), cap, 0.5, 0.4, 1e6, 1, 1e-4, 0))
pri = torch.rand(cap, dtype=torch.float64, device="cuda")
N.check(lib.srlx_per_add(h, cap, N.tptr(pri), N.PRIO_F64, 1, None))
out = []
for n in (1024, 2048, 4096, 7168):
    mask = torch.ones(n, dtype=torch.uint8, device="cuda")
    for _ in range(10):
        N.check(lib.srlx_per_add(h, n, N.tptr(mask), N.PRIO_NONE_MASKED, 1, None))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        N.check(lib.srlx_per_add(h, n, N.tptr(mask), N.PRIO_NONE_MASKED, 1, None))
    b.record(); torch.cuda.synchronize()
    out.append(f"add_{n} {1e3 * a.elapsed_time(b) / 100:.1f} us")
print(os.path.basename(os.environ.get("SRLX_LIB", "in-tree")), " | ".join(out))
