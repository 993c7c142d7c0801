"""Where a Rainbow lock-step's time goes on the GPU, untraced: events on the main stream (actors) and on the learner's stream, averaged over many lock-steps.
  t0 fork point (start) | actor_front end (network pass + selection + environments) | learner end | join + commit end | refresh end (= next t0)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

# SRLX_ACTOR_STREAM=high | low | normal: run the ACTORS' side (torch's current stream for the whole script) on a stream of that priority instead of the null stream --
# HIP keeps a pool of hardware queues per priority level, and a HIP graph's internal branch streams are normal-priority streams
_as = os.environ.get("SRLX_ACTOR_STREAM", "")
if _as:
    import ctypes

    _hip = ctypes.CDLL("libamdhip64.so")
    _st = ctypes.c_void_p()
    assert _hip.hipStreamCreateWithPriority(ctypes.byref(_st), 1, {"high": -1, "normal": 0, "low": 1}[_as]) == 0  # 1 = hipStreamNonBlocking
    torch.cuda.set_stream(torch.cuda.ExternalStream(_st.value))
cfg = RainbowDeviceConfig(n_envs=1024, batch_size=32, memory_capacity=1_000_000, seed=0)
eng = RainbowEngine(cfg, 0, 200, overlap=True)
eng.prefill()
for _ in range(8):
    eng.step(1)
torch.cuda.synchronize()
PH = os.environ.get("SRLX_LEARNER_PHASES", "0") == "1"  # also: the update's own phases (events recorded inside its graph), one synchronised lock-step at a time
if PH:
    # device wall-clock stamps written by one-thread launches inside the captured update (this HIP runtime refuses external event records during capture);
    # each stamp is one more launch on the update's chain (~3-5 us): read the phases as differences, the total runs ~20 us long
    from simple_distributed_rl_amd import _native as N

    STAMPS = torch.zeros(32, dtype=torch.int64, device="cuda")
    MARK = lambda i: N.check(N.lib().srlx_debug_stamp(N.tptr(STAMPS), i, N.torch_stream_ptr()))  # noqa: E731
    if os.environ.get("SRLX_PY_MARKS", "1") == "1":  # stamps between the update's host-level calls (they are nodes of its graph: they can change its queue placement)
        eng._phase_mark = MARK
    if os.environ.get("SRLX_BACKWARD_STAMPS", "1") == "1":
        N.check(N.lib().srlx_qnet_set_stamp_buffer(eng.inf_online.h, N.tptr(STAMPS)))
eng.capture_graphs()
for _ in range(50):
    eng.step(1)
torch.cuda.synchronize()
n = 300
E = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
if eng.fast:  # the round-4 lock-step: network pass + selection | environments | ring commit (before the join) | join + PER add
    ev = [[E() for _ in range(7)] for _ in range(n)]
    main = torch.cuda.current_stream()
    for k in range(n):
        e = ev[k]
        e[0].record(main)
        eng.fork_learner(1)
        e[5].record(eng.s_learner)
        e[6].record(main)  # reached as soon as the host is back from the graph launch (the main stream is idle here)
        off = eng.replay.frame_table_current()
        eng.inf_actor.forward_u8_policy(eng.replay.obs_base, off, eng.eps, cfg.seed ^ 0xAC7, eng.policy_counter, eng.actions)
        e[1].record(main)
        eng.env.step(eng.actions)
        e[2].record(main)
        eng.actor_commit_ring()
        e[3].record(main)
        eng.join_learner()
        eng.actor_commit_tree()
        eng.refresh_actor_copy()
        e[4].record(main)
    torch.cuda.synchronize()
    if PH:
        names = ["update start", "draw + gather end", "online pass end", None, "gradients end (branches joined)", "Adam end", "publish end", None, None, None, "ACTORS: stream start", "ACTORS: policy pass + environments end", "ACTORS: ring commit end", "ACTORS: join passed"] + [None] * 2 + [
            "  head backward (TD) end", "  fc1 data gradient end", "  conv3 data gradient + fold end", "  conv2 data gradient + fold end", "  conv1 weight gradient end",
            "  [branch] priority write-back end", "  [branch] conv3 weight gradient end", "  [branch] conv2 weight gradient end", "  [branch] fc1 weight gradient + Adam end"]
        rec = [[] for _ in names]
        t_end = []
        for k in range(120):
            MARK(8)
            eng.fork_learner(1)
            MARK(10)
            eng.actor_front(None)
            MARK(11)
            eng.actor_commit_ring()
            MARK(12)
            eng.join_learner()
            MARK(13)
            eng.actor_commit_tree()
            eng.refresh_actor_copy()
            MARK(9)
            torch.cuda.synchronize()
            st_ = STAMPS.cpu().tolist()
            for i in range(len(names)):
                rec[i].append((st_[i] - st_[8]) / 100.0)  # 100 MHz -> us
            t_end.append((st_[9] - st_[8]) / 100.0)
        for nm, v in zip(names + ["lock-step end"], rec + [t_end]):
            if nm is None:
                continue
            v = sorted(v[20:])
            print(f"  learner: {nm:42s} median {v[len(v) // 2]:7.1f} us   (10 % {v[len(v) // 10]:7.1f}, 90 % {v[9 * len(v) // 10]:7.1f})")
        rec = [[] for _ in names]
        for k in range(60):  # the same update with nothing beside it
            MARK(8)
            eng.fork_learner(1)
            eng.join_learner()
            torch.cuda.synchronize()
            st_ = STAMPS.cpu().tolist()
            for i in range(len(names)):
                rec[i].append((st_[i] - st_[8]) / 100.0)
        for nm, v in zip(names, rec):
            if nm is None:
                continue
            v = sorted(v[10:])
            print(f"  learner ALONE: {nm:42s} median {v[len(v) // 2]:7.1f} us")
        # ... alone, but right BEHIND an actors' lock-step without update (sequential: is it what ran before, not what runs beside?)
        rec = [[] for _ in names]
        for k in range(60):
            eng.step(0)
            MARK(8)
            eng.fork_learner(1)
            eng.join_learner()
            torch.cuda.synchronize()
            st_ = STAMPS.cpu().tolist()
            for i in range(len(names)):
                rec[i].append((st_[i] - st_[8]) / 100.0)
        for nm, v in zip(names, rec):
            if nm is None:
                continue
            v = sorted(v[10:])
            print(f"  learner BEHIND an actors' pass: {nm:42s} median {v[len(v) // 2]:7.1f} us")
        # ... and alone but COLD: 1 GiB streamed through the caches (L2 + the 256 MB Infinity Cache) before every update
        big = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
        rec = [[] for _ in names]
        for k in range(40):
            big.add_(1.0)
            MARK(8)
            eng.fork_learner(1)
            eng.join_learner()
            torch.cuda.synchronize()
            st_ = STAMPS.cpu().tolist()
            for i in range(len(names)):
                rec[i].append((st_[i] - st_[8]) / 100.0)
        for nm, v in zip(names, rec):
            if nm is None:
                continue
            v = sorted(v[5:])
            print(f"  learner ALONE, COLD caches: {nm:42s} median {v[len(v) // 2]:7.1f} us")
    for nm, i in zip(["policy pass START", "policy pass end", "environments end", "ring commit end", "add end (after join)", "learner end"], [6, 1, 2, 3, 4, 5]):
        v = sorted(ev[k][0].elapsed_time(ev[k][i]) for k in range(20, n))
        print(f"{nm:26s} median {1e3 * v[len(v) // 2]:7.1f} us   (10 % {1e3 * v[len(v) // 10]:7.1f}, 90 % {1e3 * v[9 * len(v) // 10]:7.1f})")
    sys.exit(0)
ev = [[E() for _ in range(6)] for _ in range(n)]
main = torch.cuda.current_stream()
for k in range(n):
    e = ev[k]
    e[0].record(main)
    eng.fork_learner(1)
    e[5].record(eng.s_learner)  # learner end (recorded on its stream after the update)
    q = eng._actor_net(None, None)
    e[1].record(main)  # network pass end
    eng._select_graph.replay()
    e[2].record(main)  # actor front end
    eng.join_learner()
    eng.actor_commit()
    e[3].record(main)
    eng.refresh_actor_copy()
    e[4].record(main)
torch.cuda.synchronize()
names = ["network pass end", "selection + env end", "commit end (after join)", "refresh end", "learner end"]
idx = [1, 2, 3, 4, 5]
for nm, i in zip(names, idx):
    v = sorted(ev[k][0].elapsed_time(ev[k][i]) for k in range(20, n))
    print(f"{nm:26s} median {1e3 * v[len(v) // 2]:7.1f} us   (10 % {1e3 * v[len(v) // 10]:7.1f}, 90 % {1e3 * v[9 * len(v) // 10]:7.1f})")
