"""The multi-GPU exchange under RCCL's STREAM semantics, on one GPU.

`torch.distributed` over RCCL returns from `batch_isend_irecv` / `broadcast` at once: the transfer runs on the communicator's own stream, behind what the caller
had enqueued on its current stream, and `work.wait()` is a stream-level wait -- the host never blocks.  gloo (every other N > 1 test of this repository) blocks the
host until the data has arrived, which hides a missing dependency between a transfer and the kernels around it; and RCCL refuses two ranks on one GPU
("Duplicate GPU detected"), so the real transport cannot be exercised on a one-GPU box either.

This test puts a fabric with exactly those semantics under `device/dist.py`: two ranks as two THREADS of one process (a learner-only rank and an actor rank, the
BASELINE configs[3] topology), point-to-point transfers and the parameter broadcast as device copies on a communicator stream of their own, issued when both sides
have posted, behind events of both posters' streams -- and DELAYED there by a multi-millisecond spin kernel, an order of magnitude longer than a lock-step, so
that any consumer that does not wait for its transfer reads stale staging buffers and any producer that overwrites a buffer before its send has left corrupts the
slab.  The learner's replay (ring, tree), its weights and its counters after 30 lock-steps must equal, bit for bit, those of the same job on a fabric whose every
transfer is synchronous (device idle before and after the copy).  What it covers: `TransitionBus.send_begin / send_end / recv_begin / recv_end / put_own`,
`DistributedRainbow.step / prefill / flush`, the engine's `before_env` hook, the captured update's ingest of staging slots, the broadcast / republish order
(reference: the queue / board hand-over of srl/base/run/play_mp.py:76-118,121-165,248-318)."""
import collections
import os
import sys
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TIMEOUT = 120.0


class _Work:
    def __init__(self, fabric, slot):
        self.fabric, self.slot = fabric, slot

    def wait(self):
        if not self.slot["matched"].wait(TIMEOUT):
            raise RuntimeError("fabric: a transfer was waited for and its peer never posted")
        if self.fabric.sync:
            return
        torch.cuda.current_stream().wait_event(self.slot["done"])  # what ProcessGroupNCCL's work.wait() is: the host goes on


class _Fabric:
    """Matches sends and receives per (source, destination) in posting order and runs each transfer on `comm`."""

    def __init__(self, dev, sync: bool, delay_cycles: int):
        self.dev, self.sync, self.delay = dev, sync, delay_cycles
        # the communicator stream sits on a priority level of its own (low): HIP keeps one pool of hardware queues per level, so its spin kernel can never share an
        # in-order queue with a stream that should have waited for it (with GPU_MAX_HW_QUEUES = 2 and a normal-priority communicator stream that happened in about
        # one run of five: the unguarded consumer then waited anyway, by accident of queue placement, and the fabric's self-check below missed the bug)
        import ctypes

        from simple_distributed_rl_amd import _native as N

        raw = ctypes.c_void_p()
        N.check(N.lib().srlx_stream_create(1, ctypes.byref(raw)))
        self._raw_comm = raw
        self.comm = torch.cuda.ExternalStream(raw.value, device=dev)
        self.lock = threading.Lock()
        self.sends = collections.defaultdict(collections.deque)
        self.recvs = collections.defaultdict(collections.deque)
        self.bcast = collections.defaultdict(collections.deque)  # per receiving rank: posted sources
        self.bcast_cv = threading.Condition(self.lock)
        self.bytes = 0
        self.source_posted = threading.Event()  # rank 0 has built its networks (the first thing a job does on the fabric is its parameter broadcast)

    def _posted(self, tensor):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        if self.sync:  # (event / stream waits only: a device-wide synchronise from this thread would invalidate a stream capture running in the other rank's)
            ev.synchronize()
        return {"tensor": tensor, "posted": ev, "matched": threading.Event(), "done": torch.cuda.Event()}

    def _copy(self, src, dst):
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(src["posted"])
            self.comm.wait_event(dst["posted"])
            if self.delay and not self.sync:
                torch.cuda._sleep(self.delay)
            dst["tensor"].copy_(src["tensor"].view(dst["tensor"].dtype).view(dst["tensor"].shape))
            src["done"].record(self.comm)
            dst["done"].record(self.comm)
        self.bytes += dst["tensor"].numel() * dst["tensor"].element_size()
        if self.sync:
            dst["done"].synchronize()
        src["matched"].set()
        dst["matched"].set()

    def post(self, kind, me, peer, tensor):
        slot = self._posted(tensor)
        with self.lock:
            key = (me, peer) if kind == "send" else (peer, me)
            mine, theirs = (self.sends, self.recvs) if kind == "send" else (self.recvs, self.sends)
            if theirs[key]:
                other = theirs[key].popleft()
                self._copy(slot, other) if kind == "send" else self._copy(other, slot)
            else:
                mine[key].append(slot)
        return _Work(self, slot)

    def broadcast(self, me, src_rank, world, tensor, members=None):
        slot = self._posted(tensor)
        members = list(range(world)) if members is None else list(members)
        world = len(members)
        with self.bcast_cv:
            if me == src_rank:  # the source's stream, too, waits for the collective: its buffer is read until the last receiver has its copy
                slot["served"] = []
                for r in members:
                    if r != me:
                        self.bcast[r].append(slot)
                self.bcast_cv.notify_all()
                self.source_posted.set()
                if not self.bcast_cv.wait_for(lambda: len(slot["served"]) == world - 1, TIMEOUT):
                    raise RuntimeError("fabric: a broadcast's receivers never posted")
                served = list(slot["served"])
            else:
                if not self.bcast_cv.wait_for(lambda: len(self.bcast[me]) > 0, TIMEOUT):
                    raise RuntimeError("fabric: a broadcast's source never posted")
                src = self.bcast[me].popleft()
                self._copy({"tensor": src["tensor"], "posted": src["posted"], "matched": threading.Event(), "done": torch.cuda.Event()}, slot)
                src["served"].append(slot["done"])
                self.bcast_cv.notify_all()
                served = [slot["done"]]
        if not self.sync:
            for ev in served:
                torch.cuda.current_stream(self.dev).wait_event(ev)  # (a non-async collective: the caller's stream waits for it)


class _P2POp:
    def __init__(self, op, tensor, peer, group=None, tag=0):
        self.op, self.tensor, self.peer = op, tensor, peer


class _FakeDist:
    """The part of torch.distributed that device/dist.py uses, for ONE rank of the fabric."""

    P2POp = _P2POp

    def __init__(self, fabric, rank, world):
        self.fabric, self.rank, self.world = fabric, rank, world

    @staticmethod
    def isend(*a, **k):
        raise AssertionError("only through batch_isend_irecv")

    irecv = isend

    def is_initialized(self):
        return True

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world

    def get_backend(self, group=None):
        return "nccl"

    def batch_isend_irecv(self, ops):
        return [self.fabric.post("send" if op.op is _Switch.isend else "recv", self.rank, op.peer, op.tensor) for op in ops]

    def broadcast(self, tensor, src=0, group=None):
        self.fabric.broadcast(self.rank, src, self.world, tensor, members=group)

    def new_group(self, ranks):
        return list(ranks)  # (a group is its member list: only `broadcast` takes one here)

    def gather(self, *a, **k):
        raise AssertionError("the slot exchange does not gather")


class _Switch:
    """What `device.dist.dist` is replaced with: every thread sees its own rank's facade."""

    local = threading.local()

    @staticmethod
    def isend(*a, **k):
        raise AssertionError("only through batch_isend_irecv")

    @staticmethod
    def irecv(*a, **k):
        raise AssertionError("only through batch_isend_irecv")

    P2POp = _P2POp

    def __getattr__(self, name):
        return getattr(_Switch.local.facade, name)


class _DeviceBytes:
    """Zero-copy uint8 view of device memory for torch.as_tensor (the frame ring lives in libsrlx, not in a torch tensor)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _replay_snapshot(rp, batch: int):
    """Everything the exchange wrote into the learner's replay: the whole frame ring, the whole tree, and the scalars of a fixed spread of items."""
    import ctypes

    import numpy as np

    from simple_distributed_rl_amd import _native as N

    frames = torch.as_tensor(_DeviceBytes(rp.obs_base, rp.E * rp.L * rp.F), device="cuda").clone()
    tree = np.empty(2 * rp.capacity - 1)
    N.check(rp.lib.srlx_per_backup(rp.h_per, ctypes.byref(N.c_f64(0)), ctypes.byref(N.c_i64(0)), ctypes.byref(N.c_i64(0)), N.np_ptr(tree)))
    items = []
    for first in range(0, rp.capacity - batch, max(batch, (rp.capacity - batch) // 24)):
        rp.batch.indices.copy_(torch.arange(first, first + batch, dtype=torch.int64, device="cuda") + rp.capacity - 1)
        b = rp.gather_drawn(all_states=False)
        items += [b.actions.clone(), b.rewards.clone(), b.terminated.clone()]
    torch.cuda.synchronize()
    return [frames, torch.tensor(tree)] + items


def _job(sync: bool, delay_cycles: int, steps: int, actor_priority: bool, learner_acts: bool = False):
    import simple_distributed_rl_amd.device.dist as dmod
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

    dev = torch.device("cuda:0")
    fabric = _Fabric(dev, sync, delay_cycles)
    out, errors = {}, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            # (two ranks in ONE process: neither may sit on the legacy default stream, whose launches synchronise implicitly with the other rank's streams and
            # would invalidate a stream capture running there -- a rank of a real job has its process, and its default stream, to itself)
            torch.cuda.set_stream(torch.cuda.Stream(device=dev))
            _Switch.local.facade = _FakeDist(fabric, rank, 2)
            E = 512
            cfg = RainbowDeviceConfig(n_envs=E, batch_size=32, memory_capacity=E * 10, memory_warmup_size=E * 3, seed=23, target_model_update_interval=4,
                                      actor_initial_priority=actor_priority)
            job = dmod.DistributedRainbow(cfg, 0, episode_len=6, sync_interval=5, learner_acts=learner_acts)
            assert job.local.fast and job.local.role == (("both" if learner_acts else "learner") if rank == 0 else "actor")
            # (eager launches throughout: the update's captured graph is recorded from exactly these streams and events, and its launch is ordered behind the same
            # `wait_stream`; a stream capture in one thread while the other rank's thread drives the runtime is not something this HIP runtime survives reliably)
            for k in range(steps):
                job.step(learner_updates=1)
            job.flush()
            torch.cuda.synchronize()
            if rank == 0:
                rp = job.replay
                out["per"] = rp.per_state()
                out["info"] = {k_: v for k_, v in job.info().items() if k_ in ("train_count", "memory", "loss")}
                out["flat"] = job.flat.detach().clone()
                out["ring"] = _replay_snapshot(rp, cfg.batch_size)
                out["graphs"] = len(job.local._learner_graphs)
            else:
                out["actor_flat"] = job.flat.detach().clone()
                out["actor_steps"] = int(job.env_steps_local)
            job.local.close()
        except Exception:
            import traceback

            errors.append(f"rank {rank}:\n{traceback.format_exc()}")

    saved = dmod.dist
    dmod.dist = _Switch()
    try:
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in (0, 1)]
        # (two ranks of a real job are two processes with a random generator each; here they share torch's: rank 1 starts once rank 0 has initialised its networks --
        # rank 1's own initial weights are overwritten by the first broadcast)
        threads[0].start()
        assert fabric.source_posted.wait(TIMEOUT) or errors, "rank 0 never reached its first broadcast"
        threads[1].start()
        for t in threads:
            t.join(4 * TIMEOUT)
        assert not any(t.is_alive() for t in threads), "a rank hung"
        assert not errors, "\n".join(errors)
    finally:
        dmod.dist = saved
    out["bytes"] = fabric.bytes
    return out


@pytest.mark.parametrize("actor_priority,learner_acts", [(False, False), (True, False), (False, True)])
def test_exchange_under_stream_ordered_transfers_equals_synchronous_transfers(actor_priority, learner_acts):
    """learner_acts: the 2-GPU topology (rank 0 runs the single-GPU lock-step on its own environments and ingests both ranks' slab: `put_own` beside the receives)."""
    steps = 30
    want = _job(sync=True, delay_cycles=0, steps=steps, actor_priority=actor_priority, learner_acts=learner_acts)
    got = _job(sync=False, delay_cycles=6_000_000, steps=steps, actor_priority=actor_priority, learner_acts=learner_acts)  # ~2.5-3 ms per transfer: ten lock-steps' worth
    assert want["info"]["train_count"] >= 12 and want["actor_steps"] == steps * 512
    assert got["info"] == want["info"] and got["per"] == want["per"] and got["graphs"] == want["graphs"] and got["bytes"] == want["bytes"]
    for k, (a, b) in enumerate(zip(got["ring"], want["ring"])):
        assert torch.equal(a, b), f"ring tensor {k} differs: a consumer ran ahead of its transfer, or a producer overwrote a buffer in flight"
    assert torch.equal(got["flat"], want["flat"]), "the learner's weights differ"
    assert torch.equal(got["actor_flat"], want["actor_flat"]) and torch.equal(got["actor_flat"], got["flat"])  # lock-step 30 ended with a broadcast (interval 5)


@pytest.mark.parametrize("broken", ["recv_end", "send_end"])
def test_the_fabric_exposes_a_missing_stream_wait(broken, monkeypatch):
    """A check OF the check, in the default suite since round 6: the communicator stream has a hardware-queue pool of its own (see _Fabric), and the broken job is
    repeated up to three times until it shows a difference (round 5: opt-in, 8 of 10 runs detected `recv_end`).  The test above has teeth: with the stream-level wait of `TransitionBus.recv_end` (the learner's staging slot is read by the next update's ingest) or of
    `send_end` (the actor's environments overwrite the frames a send is still reading) taken out -- the host still learns that the peer has posted, as it would
    over RCCL -- the delayed fabric produces a different replay."""
    import simple_distributed_rl_amd.device.dist as dmod

    steps = 14
    want = _job(sync=True, delay_cycles=0, steps=steps, actor_priority=False)

    def no_stream_wait(self):
        for w in self._pending:
            assert w.slot["matched"].wait(TIMEOUT)
        self._pending = []
        if broken == "send_end":
            self._keep = None
        else:
            for host, view in self._staged_in:
                view.copy_(host.to(self.device))
            self._staged_in = []

    monkeypatch.setattr(dmod.TransitionBus, broken, no_stream_wait)
    # (every transfer ~25-30 ms late: several eager lock-steps of host time, so that the unguarded consumer / producer is certain to run first)
    for attempt in range(3):
        got = _job(sync=False, delay_cycles=60_000_000, steps=steps, actor_priority=False)
        if any(not torch.equal(a, b) for a, b in zip(got["ring"], want["ring"])) or not torch.equal(got["flat"], want["flat"]):
            return
    raise AssertionError(f"three runs without `{broken}`'s stream wait reproduced the synchronous job bit for bit: the fabric does not expose the missing dependency")


def _job_a57(sync: bool, delay_cycles: int, steps: int):
    """BASELINE configs[3] (DistributedAgent57Light: dedicated learner rank + one actor rank) on the fabric, every rank on the all-libsrlx engine and the slot
    exchange (round 6: packed records + frames into rotating staging slots, the slab's commit -- ring, item fields, tree add -- inside the learner's update behind its
    draw, the priority write-back behind the add).  No parameter broadcast after the first one (sync_interval beyond the run): what the actor rank plays depends on the
    initial weights only."""
    import simple_distributed_rl_amd as srl
    import simple_distributed_rl_amd.device.dist as dmod
    from simple_distributed_rl_amd.algorithms import agent57_light

    dev = torch.device("cuda:0")
    fabric = _Fabric(dev, sync, delay_cycles)
    out, errors = {}, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            torch.cuda.set_stream(torch.cuda.Stream(device=dev))
            _Switch.local.facade = _FakeDist(fabric, rank, 2)
            cfg = agent57_light.Config(batch_size=8, actor_num=4, target_model_update_interval=5, episodic_memory_capacity=64, ucb_window_size=6)
            cfg.window_length = 4
            cfg.memory.capacity, cfg.memory.warmup_size = 8 * 12, 32  # (a 17-slot ring: the run below writes every slot, so the whole ring can be compared)
            cfg.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1000)
            cfg.hidden_block.set_dueling_network((64,))
            env = srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(84, 84), n_actions=3, episode_len=7)))
            cfg.setup(env)
            if rank == 0:
                torch.manual_seed(100)
            job = dmod.DistributedAgent57Light(cfg, 8, 0, episode_len=7, sync_interval=10**6, learner_acts=False, seed=5)
            for k in range(steps):
                if k == steps // 2:
                    job.capture_graphs()  # (the second half replays the lazily captured update variants: one per staging slot)
                job.step(learner_updates=1)
            job.flush()
            torch.cuda.synchronize()
            if rank == 0:
                rp = job.replay
                lx = job.local.lx
                out["x"] = torch.stack([lx[k].float() for k in ("r_int", "actor", "prev_action", "prev_r_ext", "prev_r_int")]).clone()
                out["flat"] = job.flat.detach().clone()
                out["losses"] = job.local.losses()
                out["frames"] = torch.as_tensor(_DeviceBytes(rp.obs_base, rp.E * rp.L * rp.F), device="cuda").clone()
                out["size"] = rp.per_state()["size"]
                out["train_count"] = job.train_count
            else:
                out["actor_steps"] = int(job.env_steps_local)
        except Exception:
            import traceback

            errors.append(f"rank {rank}:\n{traceback.format_exc()}")

    saved = dmod.dist
    dmod.dist = _Switch()
    try:
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in (0, 1)]
        threads[0].start()
        assert fabric.source_posted.wait(TIMEOUT) or errors, "rank 0 never reached its first broadcast"
        threads[1].start()
        for t in threads:
            t.join(4 * TIMEOUT)
        assert not any(t.is_alive() for t in threads), "a rank hung"
        assert not errors, "\n".join(errors)
    finally:
        dmod.dist = saved
    return out


def test_agent57_light_exchange_under_stream_ordered_transfers():
    """The same check for the configs[3] job (round 6: on the slot exchange with the in-update ingest, like the Rainbow job above): frames and the five UVFA /
    intrinsic fields in the learner's global replay, the learner's weights (all five networks, one flat buffer) and its four losses -- bit for bit between
    synchronous transfers and asynchronous ones delayed by ~3 ms on a communicator stream of their own."""
    steps = 22
    want = _job_a57(sync=True, delay_cycles=0, steps=steps)
    got = _job_a57(sync=False, delay_cycles=6_000_000, steps=steps)
    assert want["actor_steps"] == steps * 8 and want["size"] == got["size"] > 0 and want["train_count"] > 0
    assert torch.equal(got["frames"], want["frames"]), "frames differ: a transfer was overtaken by the kernels around it"
    assert torch.equal(got["x"], want["x"]), "UVFA / intrinsic fields differ"
    assert torch.equal(got["flat"], want["flat"]) and got["losses"] == want["losses"], "the learner trained on something else"


def _job_replay_role(sync: bool, delay_cycles: int, steps: int, host_sync: bool):
    """The three-role topology (device/replay_role.py: learner <- replay GPU <- actor) as three threads over the fabric."""
    import simple_distributed_rl_amd.device.dist as dmod
    import simple_distributed_rl_amd.device.replay_role as rmod
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

    dev = torch.device("cuda:0")
    fabric = _Fabric(dev, sync, delay_cycles)
    out, errors = {}, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            torch.cuda.set_stream(torch.cuda.Stream(device=dev))
            _Switch.local.facade = _FakeDist(fabric, rank, 3)
            cfg = RainbowDeviceConfig(n_envs=64, batch_size=8, memory_capacity=64 * 40, memory_warmup_size=128, target_model_update_interval=5, seed=11)
            top = rmod.ReplayRoleRainbow(cfg, 0, episode_len=9, sync_interval=4, prefetch=2, updates=1)
            top.host_sync = host_sync
            for _ in range(steps):
                top.step()
            top.finish()
            if top.role == "learner":
                out["flat"], out["loss"], out["train_count"] = top.flat.detach().clone(), float(top.local.loss.item()), top.local.train_count
            elif top.role == "replay":
                import ctypes

                import numpy as np

                from simple_distributed_rl_amd import _native as N

                rp = top.replay
                tree = np.empty(2 * rp.capacity - 1, np.float64)
                mp_, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
                N.check(rp.lib.srlx_per_backup(rp.h_per, ctypes.byref(mp_), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
                out["per"] = (mp_.value, size.value, write.value, torch.tensor(tree))
                out["served"], out["write_backs"] = top.served, int(top.step_dev.item())
            else:
                out["actor_flat"] = top.flat.detach().clone()
        except Exception:
            import traceback

            errors.append(f"rank {rank}:\n{traceback.format_exc()}")

    saved = dmod.dist, rmod.dist
    dmod.dist = rmod.dist = _Switch()
    try:
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in (0, 1, 2)]
        threads[0].start()  # (the ranks share torch's generator: the others start once the learner has built its networks and posted its first broadcast)
        assert fabric.source_posted.wait(TIMEOUT) or errors, "the learner rank never reached its first broadcast"
        for t in threads[1:]:
            t.start()
        for t in threads:
            t.join(4 * TIMEOUT)
        assert not any(t.is_alive() for t in threads), "a rank hung"
        assert not errors, "\n".join(errors)
    finally:
        dmod.dist, rmod.dist = saved
    return out


def test_replay_gpu_role_under_stream_ordered_transfers():
    """The replay-GPU topology (device/replay_role.py) under RCCL's stream semantics: batch messages, priority write-backs and the actors' slabs as
    asynchronous, DELAYED transfers on the fabric's communicator stream, WITHOUT the per-lock-step host synchronisation the role used to put in front of its
    posts -- the learner's weights, loss and update count and the replay rank's tree (every leaf, max_priority, size, write position) equal those of the same job
    over synchronous transfers, bit for bit."""
    steps = 30
    a = _job_replay_role(sync=True, delay_cycles=0, steps=steps, host_sync=True)
    b = _job_replay_role(sync=False, delay_cycles=6_000_000, steps=steps, host_sync=False)
    assert a["train_count"] == b["train_count"] > 10 and a["served"] == b["served"] == steps and a["write_backs"] == b["write_backs"]
    assert a["loss"] == b["loss"] and torch.equal(a["flat"], b["flat"]) and torch.equal(a["actor_flat"], b["actor_flat"])
    for x, y in zip(a["per"], b["per"]):
        assert (torch.equal(x, y) if torch.is_tensor(x) else x == y)
    # the check has teeth: with the stream-level waits for the posted groups taken out, the same delayed fabric trains on batches that have not arrived
    import simple_distributed_rl_amd.device.replay_role as rmod

    orig = rmod.ReplayRoleRainbow.__dict__["_complete"]  # (the staticmethod object itself: the attribute access would hand back the bare function)
    rmod.ReplayRoleRainbow._complete = staticmethod(lambda works: None)
    try:
        c = _job_replay_role(sync=False, delay_cycles=6_000_000, steps=steps, host_sync=False)
    finally:
        rmod.ReplayRoleRainbow._complete = orig
    assert not torch.equal(a["flat"], c["flat"])
