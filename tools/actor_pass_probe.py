"""The actors' network pass exactly as the engine launches it (E rows, fused convolution kernel, first dense layer on operand planes refreshed
from an online network, head) in a loop -- the target of the rocprofv3 kernel-trace / PMC passes of tools/r3_measure.sh.  Prints the pass time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.manual_seed(0)
online, actor = EngineQNet(6).cuda(), EngineQNet(6).cuda()
qn = QNetInference(actor, E)
if os.environ.get("SRLX_FC1_PLANES", "0") == "1":  # default: the configuration the engine ships beside a learner (operands split while staging)
    qn.enable_fc1_planes(private_weights=True)
F = 84 * 84
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randint(0, 256, (4096 * F,), dtype=torch.uint8, device="cuda", generator=g)
off = torch.randint(0, 4096, (E, 4), device="cuda", generator=g) * F
for _ in range(5):
    qn.refresh_from(online)
    q = qn.forward_u8(ring.data_ptr(), off)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    qn.refresh_from(online)
    q = qn.forward_u8(ring.data_ptr(), off)
b.record()
torch.cuda.synchronize()
print(f"E={E}: {1e3 * a.elapsed_time(b) / reps:.1f} us per (weight refresh + pass); checksum {float(q.double().sum()):.9f}")
