"""update / sample / add timings of the small PER operations for the library selected by SRLX_LIB (A/B of kernel variants)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd import _native as N
lib = N.lib()
cap = 1_000_000
h = N.c_p(); N.check(lib.srlx_per_create(ctypes.byref(h), cap, 0.5, 0.4, 1e6, 1, 1e-4, 0))
pri = torch.rand(cap, dtype=torch.float64, device="cuda")
N.check(lib.srlx_per_add(h, cap, N.tptr(pri), N.PRIO_F64, 1, None))
out = []
for B in (32, 64, 128):
    idx = torch.randint(0, cap, (B,), device="cuda") + cap - 1
    p = torch.rand(B, dtype=torch.float32, device="cuda")
    for _ in range(20):
        N.check(lib.srlx_per_update(h, B, N.tptr(idx), N.tptr(p), N.PRIO_F32, 1, None))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300):
        N.check(lib.srlx_per_update(h, B, N.tptr(idx), N.tptr(p), N.PRIO_F32, 1, None))
    b.record(); torch.cuda.synchronize()
    out.append(f"update_{B} {1e3 * a.elapsed_time(b) / 300:.1f} us")
print(os.path.basename(os.environ.get("SRLX_LIB", "in-tree")), " | ".join(out))
