"""Times srlx_adam_step alone on the network's small tensors and on all of them (HIP events, 200 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.qnet import DeviceAdam, EngineQNet, QNetInference

dev = torch.device("cuda:0")
net = EngineQNet(6).to(dev)
inf = QNetInference(net, 128, 0)
inf.enable_training(32)
steps = torch.zeros(1, dtype=torch.int64, device=dev)
for fused in (False, True):
    opt = DeviceAdam(inf._params(), lr=1e-4)
    if fused:
        opt.fuse_first_dense(inf, steps)
    for _ in range(10):
        opt.step(steps)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        opt.step(steps)
    b.record()
    torch.cuda.synchronize()
    print("fused" if fused else "all tensors", "numel", sum(opt.params[i].numel() for i in opt._idx), f"{a.elapsed_time(b) / 200 * 1e3:.1f} us per launch", flush=True)
