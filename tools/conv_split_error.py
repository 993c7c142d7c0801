"""Forward error of the fused convolution kernel's split against a float64 evaluation of the same network (torch, CPU): max |q - q64| / max |q64| over a batch of
random frames, for the split in force (default: two float16 parts; SRLX_CONV_BF16X3=1: three bf16 parts; SRLX_CONV1_F32=1 SRLX_FC1_F32=1: the float32 pipe).
Usage: python tools/conv_split_error.py [scale]   (scale multiplies the convolution weights: activations further from / closer to float16's range limits)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import copy

import torch

from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
torch.manual_seed(0)
net = EngineQNet(6).cuda()
with torch.no_grad():
    for name, p in net.named_parameters():
        if "conv" in name and p.dim() == 4:
            p.mul_(scale)
B, F = 160, 84 * 84
qn = QNetInference(net, B)
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randint(0, 256, (600 * F,), dtype=torch.uint8, device="cuda", generator=g)
idx = torch.randint(0, 600, (B, 4), device="cuda", generator=g)
q = qn.forward_u8(ring.data_ptr(), idx * F).double().cpu()
net64 = copy.deepcopy(net).double().cpu()
frames = ring.view(600, 84, 84).cpu()[idx.cpu()]  # [B, 4, 84, 84]
with torch.no_grad():
    q64 = net64(frames.double() / 255.0)
err = float((q - q64).abs().max() / q64.abs().max())
mode = "f32 pipe" if os.environ.get("SRLX_CONV1_F32") == "1" else ("bf16x3" if os.environ.get("SRLX_CONV_BF16X3") == "1" else "f16x2")
print(f"split={mode} scale={scale}: max |q - q64| / max |q64| = {err:.3e}   (max |q64| = {float(q64.abs().max()):.4g})")
