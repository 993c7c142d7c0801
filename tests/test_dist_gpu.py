"""End-to-end check of the multi-rank actor/learner topology (device/dist.py:DistributedRainbow) with two
ranks SHARING the one GPU of the test box (gloo rendezvous, tensors staged through the host): the global
replay on rank 0 receives every rank's transitions, the learner trains, and the broadcast leaves the
actor rank with the learner's weights.  The RCCL transport (backend "nccl") cannot be given two ranks on one GPU, so its
calls -- the uint8 gathers into views of the staging buffers, the parameter broadcast, the barrier and the MAX
all-reduce bench.py uses -- are run at world size 1 on the real backend; N > 1 over xGMI is bench.py --gpus N."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret, learner_acts=True):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_distributed_rl_amd.device.dist import DistributedRainbow
        from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

        cfg = RainbowDeviceConfig(n_envs=16, batch_size=8, memory_capacity=16 * 2 * 40, memory_warmup_size=64, obs_hw=(20, 20), hidden_units=32,
                                  n_actions=4, seed=rank, target_model_update_interval=3)
        eng = DistributedRainbow(cfg, 0, episode_len=7, sync_interval=2, learner_acts=learner_acts)
        for _ in range(8):
            eng.step(learner_updates=2)
        eng.capture_graphs()  # HIP graphs mid-run: must not step the local environments without pushing
        for _ in range(4):
            eng.step(learner_updates=2)
        eng.flush()  # the exchange of the last lock-step is still in flight (software-pipelined push): commit it
        torch.cuda.synchronize()
        out = {"flat_sum": float(eng.flat.double().sum().item()), "flat_abs": float(eng.flat.double().abs().sum().item())}
        if rank == 0:
            info = eng.info()
            out.update(info)
            out["per"] = eng.replay.per_state()
            out["global_envs"] = eng.replay.E
        out["env_steps_local"] = int(eng.env_steps_local)
        ret[rank] = out
    except Exception:  # the peer's "connection closed" would otherwise hide the first failure
        import traceback

        ret[f"error{rank}"] = traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


def _spawn(world, *args):
    mgr = mp.get_context("spawn").Manager()  # never fork a process that has initialised HIP
    ret = mgr.dict()
    try:
        mp.spawn(_worker, args=(world, _free_port(), ret) + args, nprocs=world, join=True)
    except Exception:
        errs = [v for k, v in sorted(ret.items(), key=lambda kv: str(kv[0])) if str(k).startswith("error")]
        raise AssertionError("worker failed:\n" + "\n".join(errs))
    return ret


def test_distributed_rainbow_two_ranks_one_gpu():
    ret = _spawn(2)
    r0, r1 = ret[0], ret[1]
    assert r0["global_envs"] == 32
    assert r0["memory"] == 12 * 32  # every lock-step added one item per env of BOTH ranks
    assert r0["per"]["size"] == 12 * 32
    assert r0["train_count"] > 0 and r0["loss"] == r0["loss"]  # trained, loss not NaN
    # step 12 ended with a broadcast (sync_interval=2): the actor rank holds the learner's exact weights
    assert r0["flat_sum"] == r1["flat_sum"] and r0["flat_abs"] == r1["flat_abs"]


def test_distributed_rainbow_dedicated_learner_rank():
    """learner_acts=False (the default from 4 ranks, BASELINE.json configs[3]): rank 0 only learns -- the global replay holds
    the actor rank's environments only, rank 0 steps no environment, and the broadcast still hands its weights over."""
    ret = _spawn(2, False)
    r0, r1 = ret[0], ret[1]
    assert r0["global_envs"] == 16
    assert r0["memory"] == 12 * 16 and r0["per"]["size"] == 12 * 16
    assert r0["train_count"] > 0 and r0["loss"] == r0["loss"]
    assert r0["env_steps_local"] == 0 and r1["env_steps_local"] > 0
    assert r0["train_count"] <= 2 * 12 + 1  # learner_updates=2 per step (+1 for the graph capture), not twice that
    assert r0["flat_sum"] == r1["flat_sum"] and r0["flat_abs"] == r1["flat_abs"]


def _rccl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from simple_distributed_rl_amd.device.dist import TransitionBus

        E, F = 64, 84 * 84
        bus = TransitionBus(E, F, torch.uint8, dev, always_collective=True)
        g = torch.Generator(device="cuda").manual_seed(3)
        ok = True
        for _ in range(3):
            actions = torch.randint(0, 6, (E,), dtype=torch.int32, device=dev, generator=g)
            rewards = torch.randn(E, device=dev, generator=g)
            term = (torch.rand(E, device=dev, generator=g) < 0.1).to(torch.uint8)
            done = (torch.rand(E, device=dev, generator=g) < 0.2).to(torch.uint8)
            frames = torch.randint(0, 256, (E, F), dtype=torch.uint8, device=dev, generator=g)
            bus.push_begin(actions, rewards, term, done, frames)  # async_op gathers on the communicator's stream
            busy = torch.randn(256, 256, device=dev, generator=g) @ torch.randn(256, 256, device=dev, generator=g)  # work that overlaps the exchange
            got = bus.push_end()
            torch.cuda.synchronize()
            assert busy.shape == (256, 256)
            ok = ok and all(torch.equal(a, b) for a, b in zip(got, (actions, rewards, term, done, frames)))
        flat = torch.randn(1 << 20, device=dev, generator=g)
        ref = flat.clone()
        bus.broadcast_params(flat)
        dist.barrier()
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([7], dtype=torch.int64, device=dev)
        dist.broadcast(n, src=0)
        torch.cuda.synchronize()
        ret[rank] = {"push_ok": bool(ok), "bcast_ok": bool(torch.equal(flat, ref)), "max": float(t.item()), "n": int(n.item()),
                     "backend": dist.get_backend()}
    finally:
        dist.destroy_process_group()


def test_rccl_transport_calls_at_world_size_one():
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret[0] == {"push_ok": True, "bcast_ok": True, "max": 1.25, "n": 7, "backend": "nccl"}


@pytest.mark.parametrize("n,actor_gpus,launcher", [pytest.param(2, 2, True, marks=pytest.mark.slow), (4, 3, True), pytest.param(2, 2, False, marks=pytest.mark.slow)])
def test_bench_multi_rank_rehearsal(n, actor_gpus, launcher):
    """`bench.py --gpus N` with N ranks sharing the test GPU (`--backend gloo`), under the driver's launcher and WITHOUT one
    (bench.py then starts the ranks itself): the whole N>1 bench path -- rendezvous, prefill, warm-up, graph capture, timed loop,
    barriers, MAX all-reduces, the JSON line and its env-step accounting for both topologies (2: rank 0 acts and learns; 4:
    dedicated learner rank) -- minus the RCCL transport."""
    import json
    import subprocess

    steps, warmup, inner = 3, 1, 4
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--steps", str(steps), "--warmup", str(warmup), "--inner", str(inner),
             "--envs", "128", "--capacity", "100000", "--batch-size", "16", "--no-per-micro"]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port())] + bench
    else:
        cmd = [sys.executable] + bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    assert r.stdout.rstrip().splitlines()[-1] == lines[0]  # the JSON line is the last thing on the merged stdout of all ranks
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["actor_gpus"] == actor_gpus and d["scaling"] == "weak" and d["rccl_ranks"] == 0  # gloo rehearsal: no RCCL ranks
    per_step = inner * 128 * actor_gpus
    assert d["config"]["transitions_per_step"] == per_step
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - per_step) < 1e-6 * per_step  # value = all actor ranks' env-steps / time
    # one update per lock-step: eager warm-up (warmup x min(inner, 8)), 2 after the switch to graphs (captured lazily: no extra update), warm-up, timed region
    assert d["final"]["train_count"] == warmup * min(inner, 8) + 2 + warmup * inner + steps * inner
    assert "cpu_baseline" not in d and d["roofline"]["avg_launch_ms"] > 0 and d["roofline"]["pass"]["avg_launch_group_ms"] > 0
    # the same total environment count on ONE GPU, timed by rank 0 in a process of its own behind the distributed region: the line's own strong-scaling ratio
    assert d["strong_ref"]["envs"] == 128 * actor_gpus and d["strong_ref"]["value"] > 0 and abs(d["strong_ratio"] - d["value"] / d["strong_ref"]["value"]) < 1e-9


@pytest.mark.parametrize("n,actor_gpus", [pytest.param(2, 2, marks=pytest.mark.slow), (4, 3)])
def test_bench_agent57_light_multi_rank_rehearsal(n, actor_gpus):
    """BASELINE configs[3] as a bench line: `bench.py --algo agent57_light --gpus N` (here N ranks sharing the test GPU over gloo): DistributedAgent57Light,
    the grouped send/recv transition push, the flat five-network broadcast, barriers and MAX all-reduce, env-step accounting over the ACTOR ranks."""
    import json
    import subprocess

    steps, inner = 2, 4
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--algo", "agent57_light", "--gpus", str(n), "--backend", "gloo", "--steps", str(steps), "--warmup", "1", "--inner", str(inner),
           "--envs", "32", "--capacity", "4000", "--batch-size", "8", "--sync-interval", "4"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["actor_gpus"] == actor_gpus and d["rccl_ranks"] == 0 and d["config"]["envs_total"] == 32 * actor_gpus
    per_step = inner * 32 * actor_gpus
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - per_step) < 1e-6 * per_step
    assert d["final"]["train_count"] >= steps * inner and d["final"]["memory"] > 0


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """`python bench.py --gpus 8` on a smaller node over RCCL must fail loudly instead of timing one GPU."""
    import subprocess

    if torch.cuda.device_count() >= 8:
        pytest.skip("this node really has 8 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _a57_worker(rank, world, port, ret, learner_acts):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import simple_distributed_rl_amd as srl
        from simple_distributed_rl_amd.algorithms import agent57_light
        from simple_distributed_rl_amd.device.dist import DistributedAgent57Light

        cfg = agent57_light.Config(batch_size=8, actor_num=4, target_model_update_interval=5, episodic_memory_capacity=64, ucb_window_size=6)
        cfg.window_length = 4
        cfg.memory.capacity, cfg.memory.warmup_size = 2 * 8 * 30, 32
        cfg.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1000)
        cfg.hidden_block.set_dueling_network((64,))
        env = srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(84, 84), n_actions=3, episode_len=7)))
        cfg.setup(env)
        torch.manual_seed(100 + rank)  # ranks start from DIFFERENT weights: the first broadcast must make them equal
        eng = DistributedAgent57Light(cfg, 8, 0, episode_len=7, sync_interval=2, learner_acts=learner_acts, seed=5)
        for k in range(12):
            if k == 8:
                eng.capture_graphs()  # (the learner rank's update variants, captured the first time each runs)
            eng.step(learner_updates=1)
        eng.flush()
        out = {"flat_sum": float(eng.flat.double().sum().item()), "env_steps_local": eng.env_steps_local}
        if eng.acts:
            out["arm_max"] = int(eng.local.ucb.arm.max().item())
        if rank == 0:
            out.update(eng.info())
            out["global_envs"] = eng.replay.E
            lx = eng.local.lx
            out["x_nonzero"] = float(lx["r_int"].abs().sum().item())
            out["arms_seen"] = sorted(set(lx["actor"][: eng.replay._steps_committed % eng.replay.L].flatten().long().tolist()))
            out["graphs"] = len(eng.local._graphs)
        ret[rank] = out
    except Exception:
        import traceback

        ret[f"error{rank}"] = traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("learner_acts", [True, False])
def test_distributed_agent57_light_two_ranks_one_gpu(learner_acts):
    """BASELINE.json configs[3] topology with Agent57_light (dedicated learner rank for learner_acts=False), two ranks time-sharing the
    test GPU, every rank on the all-libsrlx engine and the slot exchange (round 6): transitions AND the five UVFA / intrinsic fields reach the learner's global
    replay as packed records committed inside its update (eagerly, then from lazily captured graphs), it trains all five networks, and the flat broadcast leaves
    the actor rank with the learner's weights."""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    try:
        mp.spawn(_a57_worker, args=(2, _free_port(), ret, learner_acts), nprocs=2, join=True)
    except Exception:
        errs = [v for k, v in sorted(ret.items(), key=lambda kv: str(kv[0])) if str(k).startswith("error")]
        raise AssertionError("worker failed:\n" + "\n".join(errs))
    r0, r1 = ret[0], ret[1]
    actor_ranks = 2 if learner_acts else 1
    assert r0["global_envs"] == 8 * actor_ranks and r0["memory"] == 12 * 8 * actor_ranks
    assert r0["train_count"] > 0 and all(r0[k] == r0[k] for k in ("ext_loss", "int_loss", "emb_loss", "lifelong_loss"))
    assert r0["x_nonzero"] > 0 and set(r0["arms_seen"]) <= {0, 1, 2, 3} and len(r0["arms_seen"]) > 1  # intrinsic rewards and arms arrived
    assert (r0["env_steps_local"] > 0) == learner_acts and r1["env_steps_local"] == 12 * 8
    assert r0["flat_sum"] == r1["flat_sum"]  # step 12 ended with a broadcast (sync_interval = 2)
    assert r0["graphs"] >= 2  # one captured update per staging slot


def _a57_world1_worker(rank, world, port, ret, backend, always_collective):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy

        import simple_distributed_rl_amd as srl
        from simple_distributed_rl_amd import _native as N
        from simple_distributed_rl_amd.algorithms import agent57_light
        from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine
        from simple_distributed_rl_amd.device.dist import DistributedAgent57Light

        cfg = agent57_light.Config(batch_size=16, actor_num=4, target_model_update_interval=5, episodic_memory_capacity=64, ucb_window_size=6)
        cfg.window_length = 4
        cfg.memory.capacity, cfg.memory.warmup_size = 8 * 30, 32
        cfg.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1000)
        cfg.hidden_block.set_dueling_network((64,))
        env = srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(84, 84), n_actions=3, episode_len=7)))
        cfg.setup(env)
        torch.manual_seed(7)
        ref = Agent57LightFastEngine(copy.deepcopy(cfg), 8, 0, episode_len=7, seed=5)
        torch.manual_seed(7)
        eng = DistributedAgent57Light(copy.deepcopy(cfg), 8, 0, episode_len=7, sync_interval=4, seed=5, always_collective=always_collective)
        steps = 17
        for _ in range(steps):
            ref.actor_step()
            eng.step(learner_updates=0)
        eng.flush()
        torch.cuda.synchronize()
        zero = torch.zeros(1, dtype=torch.int64, device="cuda:0")
        out = {"committed": (ref.replay._steps_committed, eng.replay._steps_committed), "len": (ref.replay.length(), eng.replay.length())}
        same = True
        for _ in range(4):  # four seeded draws from both replays: same tree, same generator -> the same items, field by field
            b1, b2 = ref.replay.sample(zero), eng.replay.sample(zero)
            same = same and all(torch.equal(getattr(b1, k), getattr(b2, k)) for k in ("indices", "obs", "actions", "rewards", "terminated", "weights"))
            loc = []
            for rp in (ref.replay, eng.replay):
                e, s = torch.zeros(rp.B, dtype=torch.int64, device="cuda:0"), torch.zeros(rp.B, dtype=torch.int64, device="cuda:0")
                N.check(rp.lib.srlx_store_locate(rp.h_store, rp.B, N.tptr(b1.indices), N.tptr(e), N.tptr(s), None, N.torch_stream_ptr()))
                loc.append((e, s))
            (e1, s1), (e2, s2) = loc
            same = same and torch.equal(e1, e2) and torch.equal(s1, s2)
            lx = eng.local.lx
            for t, k in ((ref.x_r_int, "r_int"), (ref.x_actor, "actor"), (ref.x_prev_action, "prev_action"), (ref.x_prev_r_ext, "prev_r_ext"), (ref.x_prev_r_int, "prev_r_int")):
                same = same and torch.equal(t[s1, e1], lx[k][s2, e2])
        out["same"] = bool(same)
        out["r_int_nonzero"] = float(eng.local.lx["r_int"].abs().sum().item())
        ret[rank] = out
    except Exception:
        import traceback

        ret[f"error{rank}"] = traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend,always_collective", [("gloo", False), ("nccl", True)])
def test_distributed_agent57_light_world_one_equals_the_single_engine(backend, always_collective):
    """The slot exchange must hand the learner the transition of lock-step t with the UVFA / intrinsic fields of lock-step t: at world size 1 (the rank's own
    rows are device copies into the staging slot; with RCCL initialised the group calls run too) the global replay + field arrays equal, item by item, what a plain
    Agent57LightFastEngine with the same seed stored -- packed, staged, committed one lock-step late and unpacked: nothing torn, nothing shifted by one slot."""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    try:
        mp.spawn(_a57_world1_worker, args=(1, _free_port(), ret, backend, always_collective), nprocs=1, join=True)
    except Exception:
        errs = [v for k, v in sorted(ret.items(), key=lambda kv: str(kv[0])) if str(k).startswith("error")]
        raise AssertionError("worker failed:\n" + "\n".join(errs))
    r = ret[0]
    assert r["committed"][0] == r["committed"][1] == 17 and r["len"][0] == r["len"][1]
    assert r["same"] and r["r_int_nonzero"] > 0


def _priority_worker(rank, world, port, ret, learner_acts):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes

        import numpy as np

        import hot_path_oracle as H
        from simple_distributed_rl_amd import _native as N
        from simple_distributed_rl_amd.device.dist import DistributedRainbow
        from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

        n, A = 3, 4
        cfg = RainbowDeviceConfig(n_envs=8, batch_size=8, memory_capacity=8 * 2 * 60, memory_warmup_size=1 << 40, n_actions=A, seed=3, epsilon=0.3,
                                  actor_initial_priority=True)
        eng = DistributedRainbow(cfg, 0, episode_len=19, sync_interval=4, learner_acts=learner_acts)
        steps = 40
        for _ in range(steps):
            eng.step(learner_updates=0)  # the weights stand still: the actors' cached rows are what the learner's network gives on the stored states
        eng.flush()
        torch.cuda.synchronize()
        out = {}
        if rank == 0:
            r = eng.replay
            cap, E = r.capacity, r.E
            mp_, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
            tree = np.empty(2 * cap - 1)
            N.check(r.lib.srlx_per_backup(r.h_per, ctypes.byref(mp_), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
            added = write.value  # every add appends E leaves; the last commit's add is still deferred
            slots = np.arange(added)
            idx = torch.tensor(slots + cap - 1, dtype=torch.int64, device="cuda")
            B = len(slots)
            obs = torch.zeros((B, n + 1, cfg.window_length, 84 * 84), dtype=torch.float32, device="cuda")
            act = torch.zeros((B, n), dtype=torch.int32, device="cuda")
            rew = torch.zeros((B, n), dtype=torch.float32, device="cuda")
            ter = torch.zeros((B, n), dtype=torch.float32, device="cuda")
            N.check(r.lib.srlx_store_gather_nstep(r.h_store, B, N.tptr(idx), N.tptr(obs), N.tptr(act), N.tptr(rew), N.tptr(ter), None))
            net = eng.local.q_online
            with torch.no_grad():
                q = torch.cat([net(obs[k:k + 64].view(-1, cfg.window_length, 84, 84), channels_first=True) for k in range(0, B, 64)]).view(B, n + 1, A).cpu().numpy()
            tgt = H.nstep_target(q[:, 1:], q[:, 1:], act.cpu().numpy(), rew.cpu().numpy(), ter.cpu().numpy(), None, cfg.discount, cfg.retrace_h, True, False)
            td = np.abs(tgt - q[np.arange(B), 0, act[:, 0].cpu().numpy()])
            want = (td.astype(np.float64) + cfg.memory_epsilon) ** cfg.memory_alpha
            got = tree[cap - 1:][slots]
            est = (got != 0.0) & (got != 1.0)
            out = dict(added=int(added), per_step=int(E), estimated=int(est.sum()), at_max=int((got == 1.0).sum()), empty=int((got == 0.0).sum()),
                       max_rel=float(np.max(np.abs(got[est] - want[est]) / np.maximum(want[est], 1e-6))), max_priority=mp_.value)
        ret[rank] = out
    except Exception:
        import traceback

        ret[f"error{rank}"] = traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("learner_acts", [True, False])
def test_actor_ranks_ship_initial_priorities(learner_acts):
    """cfg.actor_initial_priority over two ranks: every actor rank estimates |n-step target - Q(s_0, a_0)| of its items from its cached Q rows and ships the estimates
    in the packed record one lock-step behind the frames; the learner rank's tree add (one lock-step behind its ring commit) takes them.  The leaves of the GLOBAL
    tree equal the oracle's n-step arithmetic on the learner's own Q-values of the stored states (the weights stand still)."""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    try:
        mp.spawn(_priority_worker, args=(2, _free_port(), ret, learner_acts), nprocs=2, join=True)
    except Exception:
        errs = [v for k, v in sorted(ret.items(), key=lambda kv: str(kv[0])) if str(k).startswith("error")]
        raise AssertionError("worker failed:\n" + "\n".join(errs))
    d = ret[0]
    assert d["per_step"] == (16 if learner_acts else 8) and d["added"] > 30 * d["per_step"] and d["max_priority"] == 1.0
    assert d["estimated"] > 0.5 * d["added"] and d["at_max"] > 0 and d["empty"] > 0
    assert d["max_rel"] < 2e-4, d


# ---- round 5: the distributed job on the round-4 lock-step; the learner rank commits arrived slabs INSIDE its update -------------------------------------------------
def _fast_worker(rank, world, port, ret, learner_acts, actor_priority):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_distributed_rl_amd.device.dist import DistributedRainbow
        from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

        E = 512
        n_act = world if learner_acts else world - 1
        cfg = RainbowDeviceConfig(n_envs=E, batch_size=32, memory_capacity=E * n_act * 12, memory_warmup_size=E * n_act * 2, seed=5, target_model_update_interval=4,
                                  actor_initial_priority=actor_priority)
        eng = DistributedRainbow(cfg, 0, episode_len=9, sync_interval=3, learner_acts=learner_acts)
        assert eng.local.fast, "every rank of the job runs the round-4 lock-step"
        steps = 12
        for k in range(steps):
            if k == 6:
                eng.capture_graphs()
            eng.step(learner_updates=1)
        eng.flush()
        torch.cuda.synchronize()
        out = {"flat_sum": float(eng.flat.double().sum().item()), "role": eng.local.role, "env_steps_local": int(eng.env_steps_local)}
        if rank == 0:
            out.update(eng.info())
            out["per"] = eng.replay.per_state()
            out["graphs"] = sorted(str(k) for k in eng.local._learner_graphs)
            out["global_envs"] = eng.replay.E
        else:
            out["set_reads"] = int(eng.local._set)
        ret[rank] = out
    except Exception:
        import traceback

        ret[f"error{rank}"] = traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("learner_acts,actor_priority", [(True, False), (False, False), (False, True)])
def test_distributed_rainbow_on_the_fast_lockstep(learner_acts, actor_priority):
    """Two ranks, 512 environments each at the benchmark geometry: every rank runs the round-4 lock-step (`fast`) -- an actor rank the fused policy pass on
    parameter sets it republishes after each broadcast, the learner rank its update with the arrived slab's ring commit + tree add on a side branch of the update
    (captured per staging slot once graphs are switched on).  The global replay ends up with every lock-step of every actor rank; the actor rank holds the
    learner's weights after the last broadcast."""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    try:
        mp.spawn(_fast_worker, args=(2, _free_port(), ret, learner_acts, actor_priority), nprocs=2, join=True)
    except Exception:
        errs = [v for k, v in sorted(ret.items(), key=lambda kv: str(kv[0])) if str(k).startswith("error")]
        raise AssertionError("worker failed:\n" + "\n".join(errs))
    r0, r1 = ret[0], ret[1]
    n_act = 2 if learner_acts else 1
    assert r0["role"] == ("both" if learner_acts else "learner") and r1["role"] == "actor"
    assert r0["global_envs"] == 512 * n_act
    assert r0["memory"] == min(12 * 512 * n_act, r0["per"]["size"]) and r0["per"]["size"] == 12 * 512 * n_act  # every slab of every actor rank was committed
    assert r0["train_count"] >= 8 and r0["loss"] == r0["loss"]
    assert len(r0["graphs"]) >= 2  # one captured update per staging slot (x published set where the learner rank acts)
    assert r0["flat_sum"] == r1["flat_sum"]  # lock-step 12 ended with a broadcast (sync_interval 3)
    assert r1["env_steps_local"] == 12 * 512 and r0["env_steps_local"] == (12 * 512 if learner_acts else 0)


def _order_worker(rank, world, port, ret, fast):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes

        import numpy as np

        import hot_path_oracle as H
        from oracle_bindings import ADD_RAW, OraclePER
        from simple_distributed_rl_amd import _native as N
        from simple_distributed_rl_amd.device.dist import DistributedRainbow
        from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

        if fast:
            cfg = RainbowDeviceConfig(n_envs=512, batch_size=32, memory_capacity=512 * 9, memory_warmup_size=512 * 4, seed=11, target_model_update_interval=5)
        else:
            cfg = RainbowDeviceConfig(n_envs=16, batch_size=8, memory_capacity=16 * 2 * 20, memory_warmup_size=48, obs_hw=(20, 20), hidden_units=32, n_actions=4, seed=11,
                                      target_model_update_interval=5)
        eng = DistributedRainbow(cfg, 0, episode_len=7, sync_interval=4)
        assert eng.local.fast == fast
        rp = eng.replay
        E, B = rp.E, cfg.batch_size
        o = OraclePER(rp.capacity, cfg.memory_alpha, cfg.memory_beta_initial, cfg.memory_beta_steps, True, cfg.memory_epsilon)
        checked = 0
        for k in range(26):
            if k == 14:
                eng.capture_graphs()
            counter0, trained0, ingested0 = int(rp.rng_counter.item()), eng.local.train_count, eng._next_ingest
            eng.step(learner_updates=1)
            torch.cuda.synchronize()
            # the learner's side of this lock-step, replayed on the oracle in the order the tree must have seen: draw -> add (the slab committed inside the update) -> write-back
            if eng.local.train_count > trained0:
                u = H.rng_uniform(cfg.seed ^ 0x5EED, trained0, rp.u.numel())
                used, idx, w, _ = o.sample(B, trained0, u)
                assert used == int(rp.used.item()) and used > 0
                np.testing.assert_array_equal(rp.batch.indices.cpu().numpy(), idx)
                np.testing.assert_allclose(rp.batch.weights.cpu().numpy(), (w / 1.0).astype(np.float32), rtol=1e-6)
                checked += 1
            if eng._next_ingest > ingested0:
                for m in rp.item_mask.cpu().numpy():
                    if m:
                        o.add(None)
                    else:
                        o.add(0.0, mode=ADD_RAW)
            if eng.local.train_count > trained0:
                o.update(idx, eng.local.priorities.cpu().numpy())
        eng.flush()
        torch.cuda.synchronize()
        for m in rp.item_mask.cpu().numpy():  # (the flush committed the last slab)
            o.add(None) if m else o.add(0.0, mode=ADD_RAW)
        mp_, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
        tree = np.empty(2 * rp.capacity - 1)
        N.check(rp.lib.srlx_per_backup(rp.h_per, ctypes.byref(mp_), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
        ret[rank] = dict(checked=checked, tree_equal=bool((tree == o.tree()).all()), max_priority=(mp_.value, o.max_priority), write=(write.value, o.write),
                         graphs=len(eng.local._learner_graphs))
    except Exception:
        import traceback

        ret[f"error{rank}"] = traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fast", [False, True])
def test_learner_rank_tree_order_is_draw_add_writeback(fast):
    """The learner rank commits a slab on a side branch of its update: the update's draw samples the tree BEFORE that slab's add, the priority write-back lands AFTER
    it.  Replayed on the CPU oracle in exactly that order (draw with the keyed uniforms, E adds at max_priority / 0, update with the priorities the update produced):
    every sampled index and the final tree are bit-equal -- eagerly and from the captured graphs."""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    try:
        mp.spawn(_order_worker, args=(1, _free_port(), ret, fast), nprocs=1, join=True)
    except Exception:
        errs = [v for k, v in sorted(ret.items(), key=lambda kv: str(kv[0])) if str(k).startswith("error")]
        raise AssertionError("worker failed:\n" + "\n".join(errs))
    d = ret[0]
    assert d["checked"] >= 15 and d["graphs"] >= 2, d
    assert d["tree_equal"] and d["max_priority"][0] == d["max_priority"][1] and d["write"][0] == d["write"][1], d
