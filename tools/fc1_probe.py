"""The actors' first dense layer (1024 rows x 7744 -> 1024 units) two ways on one handle: operands split while staging (k_gemm_s16) and the
conversion-free GEMM on pre-split operand planes (k_fc1_planes, srlx_fc1_planes.hip).  Prints the FC1 launch time of each (HIP events recorded by
srlx_qnet_set_probe_fc1 right around the GEMM launch), the whole pass, and whether the Q-values are bit-identical."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
net = EngineQNet(6).cuda()
F = 84 * 84
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randint(0, 256, (4096 * F,), dtype=torch.uint8, device="cuda", generator=g)
off = torch.randint(0, 4096, (E, 4), device="cuda", generator=g) * F


def run(qn, label, reps=30):
    for _ in range(5):
        q = qn.forward_u8(ring.data_ptr(), off)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(), b.record()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for a, b in ev:
        qn.set_probe_fc1(a, b)
        q = qn.forward_u8(ring.data_ptr(), off)
    t1.record()
    torch.cuda.synchronize()
    fc = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{label:28s} FC1 {1e3 * fc[len(fc) // 2]:7.1f} us (min {1e3 * fc[0]:.1f})   pass {1e3 * t0.elapsed_time(t1) / reps:7.1f} us   checksum {float(q.double().sum()):.9f}")
    return q.clone()


plain = QNetInference(net, E)
qa = run(plain, "split while staging")
pl = QNetInference(net, E)
pl.enable_fc1_planes(private_weights=True)
qb = run(pl, "pre-split operand planes")
print("bit-identical:", bool(torch.equal(qa, qb)), " max |diff| =", float((qa - qb).abs().max()), " max |q| =", float(qa.abs().max()))
pl2 = QNetInference(net, E)
pl2.enable_fc1_planes(private_weights=False)
run(pl2, "planes, re-split per forward")
