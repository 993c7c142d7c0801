"""Probe: NGU episodic-novelty kernel at BASELINE size (E = 1024 memories x 30000 x 32 floats = 3.9 GB) as a function of
the number of live entries per memory (timing only: the append counters are set directly, the buffer content is arbitrary)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.algorithms._device_ops import NguOps

dev = torch.device("cuda:0")
E, D, cap = 1024, 32, 30000
ngu = NguOps(dev, E, D, cap, 10, 0.001, 0.008, 0.1)
cnt_ptr = ctypes.c_void_p()
N.check(N.lib().srlx_ngu_counts(ngu.h, ctypes.byref(cnt_ptr)))
hip = ctypes.CDLL("libamdhip64.so")
x = torch.rand((E, D), device=dev)
for live in (200, 2000, 30000):
    counts = torch.from_numpy(np.full(E, live, np.int64)).to(dev)

    def set_counts():
        hip.hipMemcpy(ctypes.c_void_p(cnt_ptr.value), ctypes.c_void_p(counts.data_ptr()), ctypes.c_size_t(E * 8), ctypes.c_int(3))  # device to device

    set_counts()
    ngu.episodic(x)
    torch.cuda.synchronize()
    set_counts()
    reps = 10
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ngu.episodic(x)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    gb = E * live * D * 4 / 1e9
    print(f"live {live}: {ms*1e3:.0f} us per step, {gb/ms*1e3:.0f} GB/s of {gb*1e3:.0f} MB ({gb/ms*1e3/8000*100:.1f}% of 8 TB/s)")
