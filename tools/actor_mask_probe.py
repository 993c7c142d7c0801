"""Experiment: the ACTORS' stream confined to all but k compute units (hipExtStreamCreateWithCUMask), the learner (its captured multi-stream graph, unmasked) as shipped:
do k CUs that a convolution workgroup can never occupy let the learner's small dependent kernels start without waiting for one to retire?  ms per lock-step, E = 1024."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

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


def run(k, layout):
    cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=200_000, seed=0)
    eng = RainbowEngine(cfg, 0, 200, overlap=True)
    eng.prefill()
    if k:
        free = set(range(0, 256, 256 // k)) if layout == "stride" else set(range(k))
        s_act = masked([i for i in range(256) if i not in free])
    else:
        s_act = torch.cuda.current_stream()
    with torch.cuda.stream(s_act):
        for _ in range(20):
            eng.step(1)
        torch.cuda.synchronize()
        eng.capture_graphs()
        for _ in range(50):
            eng.step(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            eng.step(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"CUs kept free of the actors' kernels {k:3d} ({layout}): {dt * 1e3:.4f} ms per lock-step, {1024 / dt:,.0f} env-steps/s", flush=True)
    eng._keep = s_act


if __name__ == "__main__":
    for k, lay in ((0, "-"), (16, "stride"), (32, "stride"), (64, "stride"), (32, "low"), (0, "-")):
        run(k, lay)
