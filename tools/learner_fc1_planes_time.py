"""A learner's 128-row forward pass with its first dense layer on borrowed operand planes (srlx_qnet_set_planes_small: k_fc1_planes_h, 128 rows = one row tile) against the
staging-split GEMM (k_gemm_s16<APlain, .., H16>): HIP-event time of the whole pass and of the dense layer's bracket, 200 launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
net = EngineQNet(6).cuda()
F = 84 * 84
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randint(0, 256, (600 * F,), dtype=torch.uint8, device="cuda", generator=g)
off = torch.randint(0, 600, (rows, 4), device="cuda", generator=g) * F
ref = QNetInference(net, rows, 0)
actor = QNetInference(net, 512, 0)
actor.enable_fc1_planes(private_weights=True)
actor.enable_actor_sets()
ref.publish_to(actor, 1, with_fc1=True)
for name, use in (("staging split", False), ("operand planes", True)):
    h = QNetInference(net, rows, 0)
    h.enable_training(32)
    if use:
        h.enable_fc1_planes(private_weights=False)
        h.set_planes_small(True, actor.set_planes_ptr(1))
    for _ in range(10):
        q = h.forward_u8(ring.data_ptr(), off)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        q = h.forward_u8(ring.data_ptr(), off)
    b.record()
    torch.cuda.synchronize()
    print(f"{name:16s} rows={rows}: {a.elapsed_time(b) / 200 * 1e3:7.1f} us per forward pass (back to back), checksum {float(q.double().sum()):.6f}")
