"""Probe: which formulation of the DQN image block is fastest on this GPU (fp32)?  Not part of the product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
net = atari_qnetwork(6).to(dev)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


class UnfoldConv(nn.Module):
    def __init__(self, conv):
        super().__init__()
        self.w, self.b = conv.weight, conv.bias
        self.k, self.s, self.p = conv.kernel_size[0], conv.stride[0], conv.padding[0]

    def forward(self, x):
        B, C, H, W = x.shape
        x = F.pad(x, (self.p,) * 4, mode="replicate")
        Ho = (H + 2 * self.p - self.k) // self.s + 1
        cols = F.unfold(x, self.k, stride=self.s)  # [B, C*k*k, Ho*Wo]
        out = torch.matmul(self.w.view(self.w.shape[0], -1), cols) + self.b.view(1, -1, 1)
        return out.view(B, -1, Ho, Ho)


def unfold_net(net):
    convs = [l for l in net.in_block.image_block.image_layers if isinstance(l, nn.Conv2d)]
    ucs = [UnfoldConv(c) for c in convs]

    def fwd(x):
        for u in ucs:
            x = F.relu(u(x))
        return net.hidden_block(x.flatten(1))

    return fwd


uf = unfold_net(net)
for B in (32, 96, 1024):
    x = torch.rand(B, 4, 84, 84, device=dev)
    with torch.no_grad():
        ref = net(x, channels_first=True)
        alt = uf(x)
        print("B", B, "max diff unfold vs conv", float((ref - alt).abs().max()))
        t1 = timeit(lambda: net(x, channels_first=True))
        t2 = timeit(lambda: uf(x))
        xcl = x.contiguous(memory_format=torch.channels_last)
        netcl = atari_qnetwork(6).to(dev).to(memory_format=torch.channels_last)
        t3 = timeit(lambda: netcl(xcl, channels_first=True))
    print(f"  fwd no-grad: conv2d {t1:.0f}us  unfold+matmul {t2:.0f}us  channels_last {t3:.0f}us")
    # per-layer
    with torch.no_grad():
        h = x
        for i, l in enumerate(net.in_block.image_block.image_layers):
            if isinstance(l, nn.Conv2d):
                tl = timeit(lambda: l(h))
                u = UnfoldConv(l)
                tu = timeit(lambda: u(h))
                print(f"    conv{i//2+1} in{tuple(h.shape)}: conv2d {tl:.0f}us unfold {tu:.0f}us")
            h = l(h)
        flat = h.flatten(1)
        th = timeit(lambda: net.hidden_block(flat))
        print(f"    dueling head: {th:.0f}us")

x = torch.rand(32, 4, 84, 84, device=dev)
g = torch.rand(32, 6, device=dev)


def fb(f):
    def run():
        net.zero_grad(set_to_none=True)
        f(x).backward(g)
    return run


print("B=32 fwd+bwd: conv2d %.0fus  unfold %.0fus" % (timeit(fb(lambda x: net(x, channels_first=True))), timeit(fb(uf))))
opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True)
net(x, channels_first=True).backward(g)
print("adam step %.0fus" % timeit(lambda: opt.step()))
optf = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True, fused=True)
print("adam fused step %.0fus" % timeit(lambda: optf.step()))
