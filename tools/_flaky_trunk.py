"""Repeats the body of tests/test_agent57_engine_gpu.py::test_trainable_trunk_gradients_equal_autograd on ONE engine and reports which of the six gradients differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from simple_distributed_rl_amd import _native as N
import test_agent57_engine_gpu as T

eng, cfg = T._engine84()
for _ in range(8):
    eng.step(learner_updates=0)
rp = eng.replay
rp.sample_items(eng.train_count_dev, all_states=True)
B, W = rp.B, eng.Wn
off01 = rp.frame_off_all.view(2 * B, W)
obs = torch.zeros((B, 2, W, 84 * 84), dtype=torch.float32, device="cuda")
N.check(rp.lib.srlx_store_gather_nstep(rp.h_store, B, N.tptr(rp.batch.indices), N.tptr(obs), N.tptr(rp.batch.actions), N.tptr(rp.batch.rewards), N.tptr(rp.batch.terminated), N.torch_stream_ptr()))
stack = obs.view(2 * B, W, 84, 84)
net = eng.parameter.emb_network
trunk = eng._ltrunks["emb"]
params = [t for c in trunk.convs for t in (c.weight, c.bias)]
names = ["c1.w", "c1.b", "c2.w", "c2.b", "c3.w", "c3.b"]
g = torch.Generator(device="cuda").manual_seed(5)
bad = 0
import copy
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    for stride in (1, 2):
        R = torch.randn((2 * B, trunk.channels * trunk.pixels), device="cuda", generator=g)
        if stride == 2:
            R[1::2] = 0
        blk64 = copy.deepcopy(net.in_block).double().cpu()
        for q in blk64.parameters():
            q.grad = None
        f64 = blk64(stack.double().cpu(), channels_first=True)
        (f64 * R.double().cpu()).sum().backward()
        ref = [t for c in [m for m in blk64.modules() if isinstance(m, torch.nn.Conv2d)] for t in (c.weight.grad, c.bias.grad)]
        for p in params:
            p.grad = None
        got_f = trunk.features(rp.obs_base, off01, stride)
        (got_f * R).sum().backward()
        torch.cuda.synchronize()
        ef = float((got_f.double().cpu() - f64.detach()).abs().max()) / float(f64.abs().max())
        errs = [float((p.grad.double().cpu() - r).abs().max()) / float(r.abs().max()) for p, r in zip(params, ref)]
        cnt = [int(((p.grad.double().cpu() - r).abs() > 1e-5 * r.abs().max()).sum()) for p, r in zip(params, ref)]
        if ef > 1e-5 or max(errs) > 1e-5:
            bad += 1
            print("iteration", it, "stride", stride, "features %.1e" % ef, " ".join(f"{n}:{e:.1e}:{c}" for n, e, c in zip(names, errs, cnt)))
            # once more, same inputs: is it reproducible?
            for p in params:
                p.grad = None
            got_f2 = trunk.features(rp.obs_base, off01, stride)
            (got_f2 * R).sum().backward()
            torch.cuda.synchronize()
            errs2 = [float((p.grad.double().cpu() - r).abs().max()) / float(r.abs().max()) for p, r in zip(params, ref)]
            print("   second try on the same inputs:", " ".join(f"{n}:{e:.1e}" for n, e in zip(names, errs2)))
print("mismatching passes:", bad)
