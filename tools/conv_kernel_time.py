"""k_convnet_fused alone (the actors' PLANES variant at E rows): mean of probed launches (HIP events recorded by the library right around the kernel) and the
phase timeline of workgroup 0.  Usage: python tools/conv_kernel_time.py [E]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
net = EngineQNet(6).cuda()
qn = QNetInference(net, E)
qn.enable_fc1_planes(private_weights=True)
F = 84 * 84
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randint(0, 256, (4096 * F,), dtype=torch.uint8, device="cuda", generator=g)
off = torch.randint(0, 4096, (E, 4), device="cuda", generator=g) * F
for _ in range(5):
    q = qn.forward_u8(ring.data_ptr(), off)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
for a, b in ev:
    a.record(), b.record()
for a, b in ev:
    qn.set_probe(a, b)
    q = qn.forward_u8(ring.data_ptr(), off)
torch.cuda.synchronize()
v = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
print(f"E={E} k_convnet_fused: min {v[0]:.1f} median {v[len(v) // 2]:.1f} mean {sum(v) / len(v):.1f} us; q checksum {float(q.double().sum()):.6f}")
dbg = torch.zeros((8, 8), dtype=torch.int64, device="cuda")
N.check(qn.lib.srlx_qnet_set_debug(qn.h, N.tptr(dbg)))
qn.forward_u8(ring.data_ptr(), off)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype("int64")
t0 = d[:, 0].min()
names = ["start", "loads+filters issued", "frames staged (barrier)", "conv1 done", "barrier", "conv2 done", "barrier", "conv3 stored"]
for k, n in enumerate(names):
    print("%-26s" % n, " ".join("%7d" % (x - t0) for x in d[:, k]))
print("phase lengths of the slowest wave: stage %d  conv1 %d  conv2 %d  conv3 %d  total %d clocks" % (
    (d[:, 2] - d[:, 0]).max(), (d[:, 3] - d[:, 2]).max(), (d[:, 5] - d[:, 4]).max(), (d[:, 7] - d[:, 6]).max(), d[:, 7].max() - t0))
