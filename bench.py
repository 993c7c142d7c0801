#!/usr/bin/env python3
"""bench.py -- Rainbow 84x84x4 actor/learner throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input on every rank:
    E lock-stepped synthetic Atari-shaped environments advance one step
        (frame-stack -> Q-network -> epsilon-greedy -> env -> ring commit -> PER add), and
    U full Rainbow learner updates run (PER sample B=32 -> n-step gather -> forwards -> fused
        TD/Huber/priority kernel -> backward -> Adam -> PER update) against a 1M-transition PER.
`value` = whole-job env-steps/s; learner updates/s is reported next to it.  All inputs (frame ring,
sum-tree, networks) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# two HIP hardware queues (see simple_distributed_rl_amd/_native.py): must be in the environment before the runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=1024, help="environments per GPU (E)")
    ap.add_argument("--updates", type=int, default=1, help="learner updates per step on the learner rank (U); 1 balances actor and learner time at E=1024")
    ap.add_argument("--capacity", type=int, default=1_000_000)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--episode-len", type=int, default=200)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP graphs")
    ap.add_argument("--no-overlap", action="store_true", help="run actor and learner back to back on one stream (N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--sync-interval", type=int, default=16, help="learner->actor weight broadcast every k steps (N>1)")
    ap.add_argument("--per-micro", action="store_true", help="also time the bulk PER sample kernel (extra field)")
    ap.add_argument("--learner-acts", choices=("auto", "yes", "no"), default="auto",
                    help="N>1: does the learner rank run actors too? auto = yes below 4 GPUs, no (dedicated learner GPU) from 4")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for rehearsing the N>1 code path with several ranks on ONE GPU)")
    ap.add_argument("--dist-selftest", action="store_true", help="run the N>1 code path (DistributedRainbow, RCCL gathers/broadcasts) at world size 1")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and rank != 0:
        # the launcher merges every rank's stdout: only rank 0 may write there (RCCL prints a version banner through C stdio,
        # flushed when a process exits -- it must not land behind rank 0's JSON line)
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    import torch

    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev_index = local_rank % torch.cuda.device_count() if args.backend != "nccl" else local_rank  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    if world > 1 or args.dist_selftest:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        dist = None

    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    cfg = RainbowDeviceConfig(n_envs=args.envs, batch_size=args.batch_size, memory_capacity=args.capacity, seed=rank)

    if dist is not None:
        from simple_distributed_rl_amd.device.dist import DistributedRainbow

        eng = DistributedRainbow(cfg, dev_index, args.episode_len, sync_interval=args.sync_interval, always_collective=args.dist_selftest,
                                 learner_acts={"auto": None, "yes": True, "no": False}[args.learner_acts])
    else:
        eng = RainbowEngine(cfg, dev_index, args.episode_len, overlap=not args.no_overlap)
    is_learner = rank == 0

    # ---- fill the replay (untimed): random-policy rollout until the ring is full, then |delta| ~ U(0,1)
    #      priorities like tests/quick/rl/memories/speedtest.py:40-41
    eng.prefill()
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(args.updates)
    torch.cuda.synchronize()
    if not args.no_graph:
        eng.capture_graphs()
        for _ in range(2):  # untimed: the first replay of a freshly captured graph instantiates it (tens of ms), whatever --warmup was
            eng.step(args.updates)
        torch.cuda.synchronize()

    # HIP events around the actors' network pass (the dominant kernel group) and, inside it, around the two launches of the
    # dominant single kernel (k_gemm<AConv>: conv2 + conv3), all on the stream they are launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    local = getattr(eng, "local", eng)
    probing = bool(local.mfma) and getattr(eng, "acts", True)
    pr = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if probing else []
    for a, b in pr:  # torch creates the underlying hipEvent_t at the first record
        a.record()
        b.record()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if probing:
            local.inf_actor.set_probe(*pr[k])
        eng.step(args.updates, events=ev[k])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ev_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    conv_ms = sum(a.elapsed_time(b) for a, b in pr) / len(pr) if probing else 0.0
    if dist is not None:  # a learner-only rank 0 runs no actor pass: report the slowest actor rank's
        t = torch.tensor([ev_ms, conv_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ev_ms, conv_ms = float(t[0].item()), float(t[1].item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    actor_gpus = world if dist is None else eng.n_actor_ranks
    env_steps = args.steps * args.envs * actor_gpus
    updates = args.steps * args.updates
    info = eng.info()
    out = {
        "metric": "env-steps/sec + learner updates/sec, Rainbow 84x84x4",
        "value": env_steps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "learner_updates_per_s": updates / elapsed,
        "config": {
            "workload": "Rainbow on synthetic 84x84x4 Atari frames, PER 1M transitions, n-step=3 (BASELINE.json configs[2])",
            "envs_per_gpu": args.envs,
            "learner_updates_per_step": args.updates,
            "batch_size": args.batch_size,
            "per_capacity": eng.replay.capacity,
            "n_step": cfg.multisteps,
            "window": cfg.window_length,
            "n_actions": cfg.n_actions,
            "noisy_dense": cfg.enable_noisy_dense,
            "epsilon": cfg.epsilon,
            "hip_graphs": not args.no_graph,
            "untimed_steps_after_graph_capture": 0 if args.no_graph else 2,
            "qnet": ("libsrlx: fp32 MFMA forward, hand-written backward (no autograd)" if getattr(eng, "mfma_train", False) else "libsrlx fp32 MFMA forward, torch autograd backward") if eng.mfma else "torch",
            "actor_learner_overlap": (not args.no_overlap) if dist is None else True,
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "topology": "1 GPU: actor+learner" if dist is None else (f"{world} GPUs: rank0 learner+actor, {world - 1} actor ranks, RCCL gather/broadcast" if eng.learner_acts
                         else f"{world} GPUs: rank0 learner + replay, {world - 1} actor ranks (BASELINE.json configs[3] topology), RCCL gather/broadcast"),
            "actor_gpus": actor_gpus,
        },
        "roofline": roofline(eng, ev_ms, conv_ms),
        "final": {"loss": info["loss"], "train_count": info["train_count"], "memory": info["memory"]},
    }
    if args.per_micro:
        out["per_micro"] = per_micro(eng)
    if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only: the other runs would only repeat it
        out["cpu_baseline"] = cpu_baseline(args, cfg)
    if dist is not None:
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: first drain what native libraries (RCCL's banner) left in C stdio buffers
    import ctypes

    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD (= the fp32 vector peak)


def _isolated_forward_ms(eng, reps=20):
    """The same kernel group with nothing else on the GPU (in the timed region the learner's streams share the chip)."""
    import torch

    local = getattr(eng, "local", eng)
    for _ in range(3):
        local._actor_net(None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    off = local.replay.frame_table_current()
    a.record()
    for _ in range(reps):
        local.inf_actor.forward_u8(local.replay.obs_base, off)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def roofline(eng, ev_ms, conv_ms=0.0):
    """The dominant hand-written kernel, timed live with events on its launch stream: k_gemm<AConv> (the implicit-GEMM
    convolutions conv2 + conv3 of the actors' network pass, two launches per step).  `pass` = the whole pass (5 kernels)."""
    if eng.mfma:
        flops = eng.actor_forward_flops()
        tf = flops / (ev_ms * 1e-3) / 1e12
        iso_ms = _isolated_forward_ms(eng)
        group = {
            "kernel": "srlx_qnet_forward_u8 over E envs: k_conv1_u8 (conv1 from the uint8 ring) + k_gemm<AConv> x2 + k_gemm<APlain,splitK> (FC1) + k_head",
            "achieved": tf,
            "frac": tf / MFMA_F32_PEAK_TFLOPS,
            "flops_per_launch_group": flops,
            "avg_launch_group_ms": ev_ms,
            "isolated": {"avg_launch_group_ms": iso_ms, "achieved": flops / (iso_ms * 1e-3) / 1e12, "frac": flops / (iso_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS},
        }
        if conv_ms > 0.0:
            cf = eng.conv_gemm_flops()
            ctf = cf / (conv_ms * 1e-3) / 1e12
            return {
                "kernel": "k_gemm<AConv, 64, true, false, 128>: implicit-GEMM convolutions conv2 + conv3 of the actors' pass (2 launches per step, "
                          "v_mfma_f32_32x32x2_f32); rocprofv3 check: 2 x this kernel's AverageNs in profiles/*_kernel_stats.csv = avg_launch_pair_ms",
                "bound": "mfma",
                "achieved": ctf,
                "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ctf / MFMA_F32_PEAK_TFLOPS,
                "traffic": None,
                "flops_per_launch_pair": cf,
                "avg_launch_pair_ms": conv_ms,
                "note": "timed inside the step loop, where the learner's streams run beside it; `pass` = the whole network pass of the actors, "
                        "`pass.isolated` = that pass alone on an idle GPU",
                "pass": group,
                "dtype": "f32 in / f32 accumulate",
            }
        group.update({"bound": "mfma", "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None, "dtype": "f32 in / f32 accumulate"})
        return group
    nbytes = eng.stack_bytes_per_launch()
    gbs = nbytes / (ev_ms * 1e-3) / 1e9
    return {
        "kernel": "k_stack_current_u8 (uint8 frame ring -> float32 [E,4,84,84] policy input)",
        "bound": "hbm",
        "achieved": gbs,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": gbs / HBM_PEAK_GBS,
        "traffic": None,
        "bytes_per_launch": nbytes,
        "avg_launch_ms": ev_ms,
    }


def per_micro(eng, draws=1 << 20, reps=20):
    """Bulk PER sampling (LDS-staged multi-workgroup descent) on the benchmark's own 1M-leaf tree; the headline entry is
    2^20 draws per call, `by_draws` shows how the fixed per-call cost amortises at 2^22 and 2^24."""
    import torch

    from simple_distributed_rl_amd import _native as N

    r = eng.replay
    d = r.dev
    depth = (2 * r.capacity - 1).bit_length() - 1
    bytes_per_draw = (depth + 1) * 8 + 8 + 12  # tree reads + uniform in + index/weight out

    def one(n, reps):
        u = torch.rand(n, dtype=torch.float64, device=d)
        idx = torch.empty(n, dtype=torch.int64, device=d)
        w = torch.empty(n, dtype=torch.float32, device=d)
        used = torch.zeros(1, dtype=torch.int64, device=d)
        step = torch.zeros(1, dtype=torch.int64, device=d)

        def run():
            N.check(r.lib.srlx_per_sample(r.h_per, n, 0, N.tptr(step), N.tptr(u), n, N.tptr(idx), None, N.tptr(w), N.tptr(used), 1, N.torch_stream_ptr()))

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        times = []
        for _ in range(3):  # median of three timed batches: one disturbed batch does not become the reported number
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                run()
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / reps)
        ms = sorted(times)[1]
        gbs = n * bytes_per_draw / (ms * 1e-3) / 1e9
        return {"draws_per_call": n, "ms_per_call": ms, "draws_per_s": n / (ms * 1e-3), "algorithmic_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}

    out = {"kernel": "per_sample bulk (k_descend_bulk + k_compact_bulk: normalising pass, or device-wide ordered compaction when a draw is rejected)", "bytes_per_draw": bytes_per_draw}
    out.update(one(draws, reps))
    out["by_draws"] = [one(1 << 22, 10), one(1 << 24, 5)]
    return out


def cpu_baseline(args, cfg):
    """The reference-shaped CPU path (sequential, one environment, batch-1 policy inference, PER in the
    C oracle, numpy target, torch-CPU network) timed on this box's host cores on a bounded sample of the
    same workload at the same env-steps : learner-updates ratio.  kind = "port" (oracle/ restatement)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch

    import hot_path_oracle as H
    from oracle_bindings import OraclePER
    from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

    host_cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    A, n, W, B = cfg.n_actions, cfg.multisteps, cfg.window_length, cfg.batch_size
    q_on = atari_qnetwork(A)
    q_tg = atari_qnetwork(A)
    q_tg.load_state_dict(q_on.state_dict())
    opt = torch.optim.Adam(q_on.parameters(), lr=cfg.lr)
    # pick the intra-op thread count that makes the reference-shaped batch-1 inference fastest on this
    # host (all-cores oversubscribes badly on many-core boxes); `cores` reports what was used
    probe = torch.rand(1, W, 84, 84)
    best, cores = None, 1
    for th in sorted({1, 4, 8, 16, 32, min(64, host_cores), host_cores}):
        if th > host_cores:
            continue
        torch.set_num_threads(th)
        with torch.no_grad():
            for _ in range(3):
                q_on(probe, channels_first=True)
            t = time.perf_counter()
            for _ in range(10):
                q_on(probe, channels_first=True)
            t = time.perf_counter() - t
        if best is None or t < best:
            best, cores = t, th
    torch.set_num_threads(cores)
    cap = 100_000  # bounded: tree depth 17 instead of 20; the network dominates the CPU time anyway
    per = OraclePER(cap, cfg.memory_alpha, cfg.memory_beta_initial, cfg.memory_beta_steps, True, cfg.memory_epsilon)
    for x in rng.random(cap):
        per.add(float(np.sqrt(x + 1e-4)), mode=2)
    store = H.StoreOracle(1, 4096, 84 * 84, W, n, A, True, 0)
    store.reset_all(rng.integers(0, 256, (1, 84 * 84), dtype=np.uint8))
    ratio = max(1, args.envs // max(1, args.updates))  # env steps per learner update, same as the GPU run
    env_steps = updates = 0
    t0 = time.perf_counter()
    deadline = t0 + args.cpu_seconds
    while time.perf_counter() < deadline:
        for _ in range(ratio):
            s = store.stack_current().reshape(1, W, 84, 84)
            with torch.no_grad():
                q = q_on(torch.from_numpy(s), channels_first=True).numpy()
            a = H.epsilon_greedy(q, [cfg.epsilon], rng.random((1, 2)))
            nxt, rew, term, done = H.synth_env_step(store, args.episode_len)
            store.commit_step(a, rew, term, done, nxt)
            per.add(None)
            env_steps += 1
            if time.perf_counter() >= deadline:
                break
        # one learner update
        _, idx, w, _ = per.sample(B, updates, rng.random(B + 8))
        valid_q = rng.integers(8, max(9, store.pos - n - 1), B)
        items = [store.gather_item(0, int(q_)) for q_ in valid_q]
        obs = np.stack([it[0] for it in items]).reshape(B, n + 1, W, 84, 84)
        act = np.stack([it[1] for it in items])
        rew = np.stack([it[2] for it in items])
        ter = np.stack([it[3] for it in items])
        nxt_t = torch.from_numpy(obs[:, 1:].reshape(B * n, W, 84, 84))
        with torch.no_grad():
            qo = q_on(nxt_t, channels_first=True).numpy().reshape(B, n, A)
            qt = q_tg(nxt_t, channels_first=True).numpy().reshape(B, n, A)
        target = H.nstep_target(qo, qt, act, rew, ter, None, cfg.discount, cfg.retrace_h, True, False)
        q0 = q_on(torch.from_numpy(obs[:, 0]), channels_first=True)
        qsel = q0[torch.arange(B), torch.from_numpy(act[:, 0]).long()]
        wt = torch.from_numpy(w.astype(np.float32))
        tt = torch.from_numpy(target)
        loss = torch.nn.functional.huber_loss(tt * wt, qsel * wt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        per.update(idx, np.abs(target - qsel.detach().numpy()).astype(np.float32))
        updates += 1
    el = time.perf_counter() - t0
    return {
        "value": env_steps / el,
        "unit": "env-steps/s",
        "learner_updates_per_s": updates / el,
        "cores": cores,
        "kind": "port",
        "sample": f"{env_steps} sequential env-steps (1 env, batch-1 inference) + {updates} learner updates (B={B}, n={n}) in {el:.1f}s, "
        f"{ratio} env-steps per update as in the GPU run; PER capacity bounded to {cap}",
    }


if __name__ == "__main__":
    main()
