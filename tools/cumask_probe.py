"""Probe: does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) confine the actor's network pass, eagerly and
under HIP-graph replay?  Prints the isolated forward time on the default stream and on masked streams."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(n_cus, total=256):
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cus):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=100_000, seed=0)
eng = RainbowEngine(cfg, 0, 200, overlap=True)
eng.prefill()
torch.cuda.synchronize()
off = eng.replay.frame_table_current()

def timeit(stream, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            eng.inf_actor.forward_u8(eng.replay.obs_base, off)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            eng.inf_actor.forward_u8(eng.replay.obs_base, off)
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

print("default stream: %.1f us" % (timeit(torch.cuda.current_stream()) * 1e3))
for n in (256, 224, 192, 128):
    s = masked_stream(n)
    print("masked %d CUs eager: %.1f us" % (n, timeit(s) * 1e3))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            eng.inf_actor.forward_u8(eng.replay.obs_base, off)
        for _ in range(3): g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): g.replay()
        b.record()
    torch.cuda.synchronize()
    print("masked %d CUs graph replay on the masked stream: %.1f us" % (n, a.elapsed_time(b) / 20 * 1e3))
