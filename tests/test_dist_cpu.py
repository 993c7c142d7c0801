"""world_size-2 gloo test of the actor->learner transition gather and the learner->actor parameter
broadcast (simple_distributed_rl_amd/device/dist.py:TransitionBus).  CPU only."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_distributed_rl_amd.device.dist import TransitionBus, flatten_parameters

        E, F = 5, 48
        dev = torch.device("cpu")
        bus = TransitionBus(E, F, torch.uint8, dev)
        ok = True
        for step in range(3):
            rng = np.random.default_rng(1000 * step + rank)
            a = torch.tensor(rng.integers(0, 6, E), dtype=torch.int32)
            r = torch.tensor(rng.standard_normal(E), dtype=torch.float32)
            t = torch.tensor(rng.integers(0, 2, E), dtype=torch.uint8)
            d = torch.tensor(rng.integers(0, 2, E), dtype=torch.uint8)
            o = torch.tensor(rng.integers(0, 256, (E, F)), dtype=torch.uint8)
            if step == 1:  # the split form the engines use: the exchange is in flight between the two calls
                bus.push_begin(a, r, t, d, o)
                got = bus.push_end()
            else:
                got = bus.push(a, r, t, d, o)
            if rank == 0:
                for src in range(world):
                    g = np.random.default_rng(1000 * step + src)
                    ea, er = g.integers(0, 6, E), g.standard_normal(E).astype(np.float32)
                    et, ed = g.integers(0, 2, E), g.integers(0, 2, E)
                    eo = g.integers(0, 256, (E, F))
                    sl = slice(src * E, (src + 1) * E)
                    ok &= bool((got[0][sl].numpy() == ea).all() and (got[1][sl].numpy() == er).all())
                    ok &= bool((got[2][sl].numpy() == et).all() and (got[3][sl].numpy() == ed).all() and (got[4][sl].numpy() == eo).all())
            else:
                ok &= got is None
        # the packed record buffer also carries per-environment float fields (Agent57_light: intrinsic reward, arm, previous action / rewards)
        bus5 = TransitionBus(E, F, torch.uint8, dev, extra_floats=5)
        x = torch.arange(E * 5, dtype=torch.float32).view(E, 5) + 1000 * rank
        got = bus5.push(a, r, t, d, o, x)
        if rank == 0:
            ok &= len(got) == 6 and tuple(got[5].shape) == (world * E, 5)
            for src in range(world):
                ok &= bool(torch.equal(got[5][src * E : (src + 1) * E], torch.arange(E * 5, dtype=torch.float32).view(E, 5) + 1000 * src))
        # parameter fan-out: the actor's network aliases the flat buffer
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        flat = flatten_parameters(net)
        ok &= flat.numel() == sum(-(-p.numel() // 64) * 64 for p in net.parameters())  # every parameter starts on a 256-byte boundary
        ok &= all(p.data_ptr() % 256 == flat.data_ptr() % 256 for p in net.parameters())
        x = torch.ones(2, 7)
        before = net(x).detach().clone()
        bus.broadcast_params(flat)
        after = net(x).detach()
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        ok &= bool(torch.equal(after, ref(x).detach()))
        if rank != 0:
            ok &= not torch.equal(before, after)
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_transition_bus_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _learner_only_worker(rank, world, port, ret):
    """BASELINE configs[3] in small: rank 0 only learns, ranks 1.. act.  The exchange is one group of point-to-point transfers."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_distributed_rl_amd.device.dist import TransitionBus

        E, F, K = 4, 32, 5
        bus = TransitionBus(E, F, torch.uint8, torch.device("cpu"), extra_floats=K, actor_ranks=range(1, world))
        gather = TransitionBus(E, F, torch.uint8, torch.device("cpu"), extra_floats=K, p2p=False)  # the collective form, every rank an equal part
        assert bus.p2p and not gather.p2p
        ok = True
        for step in range(3):
            rng = np.random.default_rng(77 * step + rank)
            a = torch.tensor(rng.integers(0, 6, E), dtype=torch.int32)
            r = torch.tensor(rng.standard_normal(E), dtype=torch.float32)
            t = torch.tensor(rng.integers(0, 2, E), dtype=torch.uint8)
            d = torch.tensor(rng.integers(0, 2, E), dtype=torch.uint8)
            o = torch.tensor(rng.integers(0, 256, (E, F)), dtype=torch.uint8)
            x = torch.tensor(rng.standard_normal((E, K)), dtype=torch.float32)
            bus.push_begin(a, r, t, d, o, x)
            got = bus.push_end()
            want = gather.push(a, r, t, d, o, x)
            if rank == 0:
                for g, w in zip(got, want):  # the actor ranks' rows agree with what the gather delivers; row block 0 (nobody acts there) is never written
                    ok &= bool(torch.equal(g[E:], w[E:])) and not bool(g[:E].any())
            else:
                ok &= got is None
        per_step = (10 + 4 * K) * E + E * F
        if rank == 0:
            ok &= bus.sent_bytes == 0 and bus.recv_bytes == 3 * per_step * (world - 1)
        else:
            ok &= bus.sent_bytes == 3 * per_step and bus.recv_bytes == 0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_a_learner_only_rank_sends_nothing_gloo_world3():
    world = 3
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_learner_only_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True, 2: True}


def _slot_worker(rank, world, port, ret, learner_acts):
    """The slot form of the exchange (device/dist.py:DistributedRainbow): packed records + frames, staging slots that rotate on the learner rank, one group of
    point-to-point transfers per lock-step; the learner's own transitions (when it acts) are copied, never sent."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_distributed_rl_amd.device.dist import TransitionBus

        E, F, K, S = 8, 16, 1, 3
        first = 0 if learner_acts else 1
        bus = TransitionBus(E, F, torch.uint8, torch.device("cpu"), extra_floats=K, actor_ranks=range(first, world))
        bus.enable_slots(S)
        ok = True

        def slab(r, step):
            rng = np.random.default_rng(1000 * step + r)
            return (torch.tensor(rng.integers(0, 6, E), dtype=torch.int32), torch.tensor(rng.standard_normal(E), dtype=torch.float32),
                    torch.tensor(rng.integers(0, 2, E), dtype=torch.uint8), torch.tensor(rng.integers(0, 2, E), dtype=torch.uint8),
                    torch.tensor(rng.integers(0, 256, (E, F)), dtype=torch.uint8), torch.tensor(rng.standard_normal((E, K)), dtype=torch.float32))

        for step in range(5):
            a, r, t, d, o, x = slab(rank, step)
            if rank == 0:
                bus.recv_begin(step % S)
                if learner_acts:
                    bus.put_own(step % S, bus.pack(a, r, t, d, x), o)
                bus.recv_end()
                scal, obs = bus.slot_scal[step % S], bus.slot_obs[step % S]
                for src in range(first, world):  # row block of actor rank `src`: exactly what that rank packed this lock-step
                    i = bus.row_of[src]
                    wa, wr, wt, wd, wo, wx = slab(src, step)
                    ok &= bool(torch.equal(scal[i], bus.pack(wa, wr, wt, wd, wx))) and bool(torch.equal(obs[i * E:(i + 1) * E], wo))
                    rec = scal[i]
                    ok &= bool(torch.equal(rec[: 4 * E].view(torch.int32), wa)) and bool(torch.equal(rec[10 * E:].view(torch.float32).view(E, K), wx))
                if step >= 1:  # the slot of the previous lock-step still holds that slab (it is committed one or two lock-steps after it arrived)
                    i = bus.row_of[world - 1]
                    ok &= bool(torch.equal(bus.slot_obs[(step - 1) % S][i * E:(i + 1) * E], slab(world - 1, step - 1)[4]))
            else:
                bus.send_end()  # (the previous slab has left)
                bus.send_begin(bus.pack(a, r, t, d, x), o)
        bus.send_end()
        per_step = (10 + 4 * K) * E + E * F
        if rank == 0:
            ok &= bus.sent_bytes == 0 and bus.recv_bytes == 5 * per_step * (world - 1)
        else:
            ok &= bus.sent_bytes == 5 * per_step and bus.recv_bytes == 0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,learner_acts", [(2, True), (3, False)])
def test_slot_exchange_gloo(world, learner_acts):
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_slot_worker, args=(world, port, ret, learner_acts), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _grad_avg_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from simple_distributed_rl_amd.device.ppo import flat_grad_all_reduce

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    x = torch.full((4, 3), float(rank + 1))
    net(x).sum().backward()
    mine = [p.grad.clone() for p in net.parameters()]
    flat_grad_all_reduce(net)
    q.put((rank, [g.numpy() for g in mine], [p.grad.numpy().copy() for p in net.parameters()]))
    dist.destroy_process_group()


def test_ppo_gradient_average_world2():
    """Data-parallel PPO (BASELINE config 5): one flat all-reduce leaves every rank with the mean gradient."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_grad_avg_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in range(2)])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, g0, a0), (_, g1, a1) = res
    for x0, x1, y0, y1 in zip(g0, g1, a0, a1):
        np.testing.assert_allclose(y0, (x0 + x1) / 2, rtol=1e-6)
        np.testing.assert_array_equal(y0, y1)


def _ppo_grad_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_distributed_rl_amd.device.ppo import flat_vector_all_reduce

        flat = torch.arange(12931, dtype=torch.float32) * (rank + 1)
        scale = flat_vector_all_reduce(flat)
        want = torch.arange(12931, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ret[rank] = bool(scale == 1.0 / world and torch.equal(flat, want))
    finally:
        dist.destroy_process_group()


def test_ppo_flat_gradient_all_reduce_gloo_world2():
    """DistributedPPO's exchange (device/ppo.py:flat_vector_all_reduce): ONE in-place sum of the flat gradient vector; the factor the optimiser launch applies is
    1 / world size (the mean of the ranks' gradients, as the reference's data-parallel learner would average them)."""
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_ppo_grad_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)
