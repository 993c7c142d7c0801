"""Probe: per-layer kernel times of srlx_qnet at a fixed batch (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
net = EngineQNet(6).cuda()
qn = QNetInference(net, max_batch=4096)
x = torch.rand(B, 4, 84, 84, device="cuda")
for _ in range(30):
    qn.forward_f32(x)
torch.cuda.synchronize()
