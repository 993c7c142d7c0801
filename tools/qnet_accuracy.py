"""Error of the device forward against the reference network evaluated in float64 (tests/golden/qnet84_*.npz), per pipe selection.
Run once per environment (the SRLX_*_F32 switches are read once per process):
  python tools/qnet_accuracy.py                        # split-bf16 pipe everywhere (the default)
  SRLX_CONV1_F32=1 SRLX_CONV23_F32=1 SRLX_FC1_F32=1 python tools/qnet_accuracy.py   # float32 pipe everywhere"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from test_qnet_pinned import _golden
from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

env = {k: v for k, v in os.environ.items() if k.startswith("SRLX_")}
for kind in ("init", "wide"):
    z, sd = _golden(kind)
    net = EngineQNet(6).cuda().load_reference_state_dict(sd)
    B = z["frames"].shape[0]
    qn = QNetInference(net, max_batch=1024)
    ring = torch.tensor(z["frames"]).permute(0, 3, 1, 2).contiguous().view(B * 4, 84 * 84).cuda()
    off = (torch.arange(B * 4, device="cuda", dtype=torch.int64) * (84 * 84)).view(B, 4).clone()
    x = torch.tensor(z["frames"].astype(np.float32) / 255).permute(0, 3, 1, 2).contiguous().cuda()
    ref32, ref64 = z["q_ref_f32"].astype(np.float64), z["q_ref_f64"]
    scale = np.abs(ref64).max()
    for label, q in (("forward_f32", qn.forward_f32(x).cpu().numpy().astype(np.float64)), ("forward_u8 ", qn.forward_u8(ring.data_ptr(), off).cpu().numpy().astype(np.float64))):
        big = np.abs(ref32) > 1e-3 * scale
        print(f"{kind:5s} {label} env={env}: max err vs f64 / max|q| = {np.abs(q - ref64).max() / scale:.3e} (reference f32: {np.abs(ref32 - ref64).max() / scale:.3e});"
              f" worst rel vs ref32 on |q|>1e-3 max: {(np.abs(q - ref32)[big] / np.abs(ref32)[big]).max():.3e}; rms err/scale {np.sqrt(((q - ref64) ** 2).mean()) / scale:.3e}")
