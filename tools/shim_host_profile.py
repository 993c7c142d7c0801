"""Where the b1 shim's host time goes: the speedtest loop (add 1, sample 64, update 64 on a 1 000 000-leaf memory) with the C calls timed apart from the Python
around them (tests/quick/rl/memories/speedtest.py:15-58 is the loop)."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.rl.memories.priority_memories.proportional_memory import ProportionalMemory

m = ProportionalMemory(1_000_000, alpha=0.5)
random.seed(0)
for i in range(20000):
    m.add((i, i, i), random.random())
lib = m._lib
acc = {}
for name in ("srlx_per_sample_after_adds_mt", "srlx_per_update"):
    f = getattr(lib, name)

    def wrap(*a, _f=f, _n=name):
        t0 = time.perf_counter()
        r = _f(*a)
        acc[_n] = acc.get(_n, 0.0) + time.perf_counter() - t0
        return r
    setattr(lib, name, wrap)
n = 3000
ts = te = tu = 0.0
for k in range(n):
    t0 = time.perf_counter()
    m.add((k, k, k), None)
    t1 = time.perf_counter()
    b, w, idx = m.sample(64, k)
    t2 = time.perf_counter()
    m.update(idx, np.random.rand(64).astype(np.float32))
    t3 = time.perf_counter()
    ts += t2 - t1
    tu += t3 - t2
    te += t1 - t0
print("add %.2f us | sample %.2f us (C call %.2f) | update %.2f us (C call %.2f) [incl. ~0.1 us of timing wrapper each]" % (
    1e6 * te / n, 1e6 * ts / n, 1e6 * acc["srlx_per_sample_after_adds_mt"] / n, 1e6 * tu / n, 1e6 * acc["srlx_per_update"] / n))
