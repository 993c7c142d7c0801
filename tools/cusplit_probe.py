"""Experiment: actors and learner on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask on every stream involved) instead of sharing the
chip through stream priorities.  Prints ms per lock-step (E = 1024, one update per lock-step) for several splits; learner eager (a
multi-stream graph's inner branches would not carry the mask), actor select/commit graphs as usual."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

hip = ctypes.CDLL("libamdhip64.so")


def masked(bits, total=256):
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask) == 0
    return torch.cuda.ExternalStream(s.value)


def run(n_learner, layout, graphs_learner=False):
    cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=200_000, seed=0)
    eng = RainbowEngine(cfg, 0, 200, overlap=True)
    eng.prefill()
    if n_learner:
        if layout == "low":
            lbits = list(range(n_learner))
        else:  # "stride": every (256 / n)-th CU
            lbits = list(range(0, 256, 256 // n_learner))[:n_learner]
        abits = [i for i in range(256) if i not in set(lbits)]
        s_act, eng.s_learner, eng.s_target = masked(abits), masked(lbits), masked(lbits)
        side = masked(lbits)
        N.check(eng.lib.srlx_qnet_set_side_stream(eng.inf_online.h, N.c_p(side.cuda_stream)))
        eng._keep_streams = (s_act, side)
    else:
        s_act = torch.cuda.current_stream()
    with torch.cuda.stream(s_act):
        for _ in range(20):
            eng.step(1)
        torch.cuda.synchronize()
        eng.capture_graphs(actor=True, learner=graphs_learner or not n_learner)
        for _ in range(50):
            eng.step(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            eng.step(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"learner CUs {n_learner:3d} layout {layout:6s} learner graph {graphs_learner or not n_learner}: {dt * 1e3:.4f} ms per lock-step, {1024 / dt:,.0f} env-steps/s", flush=True)


if __name__ == "__main__":
    run(0, "-")
    for n, lay in ((32, "low"), (64, "low"), (64, "stride"), (96, "low"), (128, "stride")):
        run(n, lay)
    run(64, "low", graphs_learner=True)
