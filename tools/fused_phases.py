"""Phase timeline of the fused convolution kernel (workgroup 0, all 8 waves) from its shader-clock stamps (srlx_qnet_set_debug)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
net = EngineQNet(6).cuda()
qn = QNetInference(net, E)
F = 84 * 84
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randint(0, 256, (4096 * F,), dtype=torch.uint8, device="cuda", generator=g)
off = torch.randint(0, 4096, (E, 4), device="cuda", generator=g) * F
for _ in range(5):
    qn.forward_u8(ring.data_ptr(), off)
dbg = torch.zeros((8, 8), dtype=torch.int64, device="cuda")
N.check(qn.lib.srlx_qnet_set_debug(qn.h, N.tptr(dbg)))
qn.forward_u8(ring.data_ptr(), off)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype("int64")
t0 = d[:, 0].min()
names = ["start", "loads+filters issued", "frames staged (barrier)", "conv1 done", "barrier", "conv2 done", "barrier", "conv3 stored"]
print("E =", E, "  shader clocks relative to the first wave's start; per wave:")
for k, n in enumerate(names):
    print("%-26s" % n, " ".join("%7d" % (x - t0) for x in d[:, k]))
print("phase lengths of the slowest wave: stage %d  conv1 %d  conv2 %d  conv3 %d  total %d clocks" % (
    (d[:, 2] - d[:, 0]).max(), (d[:, 3] - d[:, 2]).max(), (d[:, 5] - d[:, 4]).max(), (d[:, 7] - d[:, 6]).max(), d[:, 7].max() - t0))
