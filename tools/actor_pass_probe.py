"""The actors' policy pass exactly as the round-4 engine launches it beside a learner (E rows: fused convolution kernel reading a PUBLISHED parameter set, first
dense layer on operand planes with half-CU workgroups, head kernel with the epsilon-greedy selection in its epilogue) in a loop -- the target of the rocprofv3
kernel-trace / PMC passes of tools/r4_measure.sh.  SRLX_FC1_NEIGHBOUR=0: the CU-filling first-dense-layer kernel; SRLX_PROBE_S16=1: the staging-split GEMM."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.manual_seed(0)
online = EngineQNet(6).cuda()
src = QNetInference(online, 128)
qn = QNetInference(online, E)
if os.environ.get("SRLX_PROBE_S16", "0") != "1":
    qn.enable_fc1_planes(private_weights=True)
    qn.enable_actor_sets()
    qn.set_fc1_neighbour(int(os.environ.get("SRLX_FC1_NEIGHBOUR", "4")))
    src.publish_to(qn, 0, with_fc1=True)
    qn.select_set(0)
F = 84 * 84
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randint(0, 256, (4096 * F,), dtype=torch.uint8, device="cuda", generator=g)
off = torch.randint(0, 4096, (E, 4), device="cuda", generator=g) * F
eps = torch.full((E,), 0.1, device="cuda")
counter = torch.zeros(1, dtype=torch.int64, device="cuda")
actions = torch.zeros(E, dtype=torch.int32, device="cuda")
for _ in range(5):
    q = qn.forward_u8_policy(ring.data_ptr(), off, eps, 0xAC7, counter, actions)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    q = qn.forward_u8_policy(ring.data_ptr(), off, eps, 0xAC7, counter, actions)
b.record()
torch.cuda.synchronize()
print(f"E={E}: {1e3 * a.elapsed_time(b) / reps:.1f} us per policy pass; checksum {float(q.double().sum()):.9f} actions {int(actions.sum())}")
