"""Probe: time the bulk PER sample pipeline stand-alone (used with rocprofv3 for per-kernel times).
The printed checksums pin the outputs across kernel variants."""
import sys, os, ctypes, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd import _native as N
lib = N.lib()
dev = torch.device("cuda:0")
QUICK = len(sys.argv) > 1 and sys.argv[1] in ("quick", "one20")
ONE20 = len(sys.argv) > 1 and sys.argv[1] == "one20"  # 2^20 draws only (the PMC passes of tools/_pmc_per.sh)
CAPS = [(int(c), 0.0) for c in sys.argv[2:]]
for cap, zero_frac in (CAPS if CAPS else ((1_000_000, 0.0),) if QUICK else ((1_000_000, 0.0), (1_000_000, 0.01), (300_001, 0.0))):
    h = N.c_p(); N.check(lib.srlx_per_create(ctypes.byref(h), cap, 0.5, 0.4, 1e6, 1, 1e-4, 0))
    g = torch.Generator(device="cuda").manual_seed(1)
    pri = torch.rand(cap, dtype=torch.float64, device=dev, generator=g)
    if zero_frac:
        pri[torch.rand(cap, device=dev, generator=g) < zero_frac] = 0.0
        node = 0
        while 2 * node + 1 < 2 * cap - 1: node = 2 * node + 1
        pri[node - (cap - 1)] = 0.0  # the leftmost leaf: a uniform of exactly 0.0 lands on it and is rejected
        N.check(lib.srlx_per_add(h, cap, N.tptr(pri), N.PRIO_RAW if hasattr(N, "PRIO_RAW") else 3, 1, None))
    else:
        N.check(lib.srlx_per_add(h, cap, N.tptr(pri), N.PRIO_F64, 1, None))
    for draws in ((1 << 20,) if ONE20 else (1 << 20, 1 << 22) if QUICK else (1 << 20, 1 << 22, 1 << 24)):
        u = torch.rand(draws, dtype=torch.float64, device=dev, generator=g)
        B = draws if not zero_frac else draws // 2
        if zero_frac: u[torch.rand(draws, device=dev, generator=g) < zero_frac] = 0.0  # forces the in-order rejection path
        idx = torch.empty(B, dtype=torch.int64, device=dev); w = torch.empty(B, dtype=torch.float32, device=dev)
        used = torch.zeros(1, dtype=torch.int64, device=dev); step = torch.zeros(1, dtype=torch.int64, device=dev)
        def run():
            N.check(lib.srlx_per_sample(h, B, 0, N.tptr(step), N.tptr(u), draws, N.tptr(idx), None, N.tptr(w), N.tptr(used), 1, None))
        for _ in range(3): run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): run()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        cs = hashlib.sha1(idx.cpu().numpy().tobytes() + w.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"cap {cap} zeros {zero_frac} draws {draws}: {ms*1e3:.1f} us/call  {draws/ms/1e6:.2f} Gdraws/s  alg {draws*188/ms/1e6/1e3:.0f} GB/s "
              f"({draws*188/ms/1e6/8e3*100:.1f}% of 8TB/s) used {int(used.item())} checksum {cs}")
    lib.srlx_per_destroy(h)
