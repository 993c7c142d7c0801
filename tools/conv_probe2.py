import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
net = atari_qnetwork(6).to(dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
x = torch.rand(32, 4, 84, 84, device=dev); g = torch.rand(32, 6, device=dev)
x96 = torch.rand(96, 4, 84, 84, device=dev); x1k = torch.rand(1024, 4, 84, 84, device=dev)
def fb():
    net.zero_grad(set_to_none=True)
    net(x, channels_first=True).backward(g)
with torch.no_grad():
    t96 = timeit(lambda: net(x96, channels_first=True)); t1k = timeit(lambda: net(x1k, channels_first=True))
print(os.environ.get("MIOPEN_FIND_MODE"), "fwd96 %.0fus fwd1024 %.0fus fwd+bwd32 %.0fus" % (t96, t1k, timeit(fb)))
