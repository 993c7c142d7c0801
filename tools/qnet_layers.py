"""Probe: per-layer kernel times of srlx_qnet at a fixed batch (run under rocprofv3 --kernel-trace --stats).
argv: batch [u8|f32].  u8: frames come from a 4 GB uint8 ring through a random frame-offset table (HBM resident)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "u8"
net = EngineQNet(6).cuda()
qn = QNetInference(net, max_batch=4096)
if mode == "u8":
    F, n_frames = 7056, 600_000
    ring = torch.randint(0, 256, (n_frames * F,), dtype=torch.uint8, device="cuda")
    for _ in range(30):
        off = torch.randint(0, n_frames, (B, 4), device="cuda", dtype=torch.int64) * F
        qn.forward_u8(ring.data_ptr(), off)
else:
    x = torch.rand(B, 4, 84, 84, device="cuda")
    for _ in range(30):
        qn.forward_f32(x)
torch.cuda.synchronize()
