import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference
torch.backends.cudnn.benchmark = True
net = EngineQNet(6).cuda()
qn = QNetInference(net, max_batch=4096)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (32, 96, 1024, 4096):
    x = torch.rand(B, 4, 84, 84, device="cuda")
    with torch.no_grad():
        t_torch = timeit(lambda: net(x, channels_first=True))
    t_mine = timeit(lambda: qn.forward_f32(x))
    gf = B * 39.9e6 / 1e9
    print(f"B={B}: torch {t_torch:.0f}us  srlx_qnet {t_mine:.0f}us  ({gf/t_mine*1e6/1e3:.1f} TFLOP/s fp32)")
x = torch.rand(32, 4, 84, 84, device="cuda"); g = torch.rand(32, 6, device="cuda")
def fb():
    net.zero_grad(set_to_none=True)
    net(x).backward(g)
print("EngineQNet torch fwd+bwd B=32: %.0fus" % timeit(fb))
