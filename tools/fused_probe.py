"""Times srlx_qnet_forward_u8 over E samples with and without the fused conv1->conv2->conv3 kernel (SRLX_NO_FUSED_CONV=1 selects
the three-launch path; the switch is read once per process, so each arm is its own process), checks both against each other."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "arm":
    import torch

    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    E = int(sys.argv[2])
    torch.manual_seed(0)
    net = EngineQNet(6).cuda()
    qn = QNetInference(net, E)
    F = 84 * 84
    g = torch.Generator(device="cuda").manual_seed(1)
    ring = torch.randint(0, 256, (4096 * F,), dtype=torch.uint8, device="cuda", generator=g)
    off = torch.randint(0, 4096, (E, 4), device="cuda", generator=g) * F
    for _ in range(5):
        q = qn.forward_u8(ring.data_ptr(), off)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    a.record()
    for _ in range(reps):
        q = qn.forward_u8(ring.data_ptr(), off)
    b.record()
    torch.cuda.synchronize()
    print(f"E={E} fused={os.environ.get('SRLX_NO_FUSED_CONV', '0') != '1'}: {a.elapsed_time(b) / reps * 1e3:.1f} us per pass; q checksum {float(q.double().sum()):.6f} "
          f"absmax {float(q.abs().max()):.6f}")
    torch.save(q.cpu(), f"/tmp/q_{E}_{os.environ.get('SRLX_NO_FUSED_CONV', '0')}.pt")
else:
    import torch

    for E in (1024, 128, 96):
        for flag in ("0", "1"):
            env = dict(os.environ, SRLX_NO_FUSED_CONV=flag)
            subprocess.check_call([sys.executable, __file__, "arm", str(E)], env=env)
        qa, qb = torch.load(f"/tmp/q_{E}_0.pt"), torch.load(f"/tmp/q_{E}_1.pt")
        print(f"  max |fused - unfused| = {float((qa - qb).abs().max()):.3e} (relative to max |q| {float(qb.abs().max()):.3e})")
