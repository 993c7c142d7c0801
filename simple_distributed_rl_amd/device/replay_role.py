"""The reference's THREE-role topology on the device path: actors -> replay -> learner, one process per GPU.

Reference: srl/base/run/play_mp_memory.py -- a memory process between the actor processes and the trainer process (`_run_memory`, :253-351: it
ingests the actors' items, samples batches ahead of the trainer into a bounded queue (`mem_to_train`, depth 5, :595-621) and applies the priority
updates the trainer sends back through a second bounded queue (`train_to_mem`, :361-413)).  Here the memory process is a REPLAY GPU:

    rank 0   learner   trains on served batches, never touches a ring or a tree; weights -> actors every `sync_interval` lock-steps (one flat broadcast)
    rank 1   replay    owns the uint8 frame ring + the sum-tree for the environments of ALL actor ranks: ingests their slabs (round 6: the slot exchange of
                       device/dist.py:TransitionBus -- packed records + frames straight into rotating staging slots, one unpack-and-commit launch), samples, and SERVES every batch as one message -- the frames its n-step windows point at
                       packed by `srlx_pack_frames`, the frame-offset tables re-based onto the packed frames, indices / weights / n-step scalars
    rank 2.. actors    E lock-stepped environments each, exactly the actor ranks of device/dist.py:DistributedRainbow

The two bounded queues become a static software pipeline (deterministic, no polling): per lock-step the replay rank sends `updates` batch messages and
the learner returns as many priority write-backs; the learner trains on the batch it received `prefetch` lock-steps earlier (so `prefetch` batches are
always in flight: the reference's depth-5 prefetch queue), and the replay rank applies a write-back one lock-step after it was produced (its backlog is
bounded by construction; the reference bounds it at 100).  Per lock-step each of the two ranks issues ALL its transfers with the other as ONE group
(`dist.batch_isend_irecv`: on RCCL every send / receive of a rank pair shares one communicator stream, and ungrouped non-blocking calls posted in opposite
orders on the two sides wait for each other forever) and completes it at the start of its next lock-step.  Whether a batch is "warm" is a function of the
lock-step it was served in (the replay's fill level is lock-steps x environments), so both ranks compute it on the host: no header is read back from the
device, no `.item()` in the loop (the header still travels and SRLX_CHECK_HEADERS=1 compares it).  Every buffer is allocated once.  The network kernels take (base pointer, offset table): the learner evaluates the
packed frames exactly as it would a ring, through the same `RainbowEngine` update (online + target pass, fused TD / Huber / priorities, hand-written
backward, Adam) with a served batch standing in for its replay.
"""
import dataclasses
from typing import Optional

import torch
import torch.distributed as dist

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.dist import TransitionBus, flatten_parameters
from simple_distributed_rl_amd.device.qnet import DeviceAdam
from simple_distributed_rl_amd.device.replay import ReplayBatch

LEARNER, REPLAY, FIRST_ACTOR = 0, 1, 2


class BatchCodec:
    """Layout of one served batch as a flat uint8 message (both ends compute it from the configuration)."""

    def __init__(self, B: int, n: int, W: int, F: int):
        self.B, self.n, self.W, self.F = B, n, W, F
        rows = B * (n + 1) * W
        self.rows = rows
        fields = [("header", 16), ("indices", 8 * B), ("weights", 4 * B), ("actions", 4 * B * n), ("rewards", 4 * B * n), ("terminated", 4 * B * n),
                  ("rel_all", 8 * rows), ("rel_next", 8 * B * n * W), ("frames", rows * F)]
        self.off, at = {}, 0
        for name, nbytes in fields:
            at = -(-at // 256) * 256  # every field on a 256-byte boundary (vector loads of the frames, aligned int64 views)
            self.off[name] = (at, nbytes)
            at += nbytes
        self.nbytes = at

    def view(self, buf: torch.Tensor, name: str, dtype: torch.dtype) -> torch.Tensor:
        a, nb = self.off[name]
        return buf[a : a + nb].view(dtype)


class ServedBatch:
    """What `RainbowEngine._learner_body` asks of a replay, answered from one received message (learner rank)."""

    lagged = False

    def __init__(self, codec: BatchCodec, device: torch.device, train_count_dev: Optional[torch.Tensor] = None):
        c = self.codec = codec
        self.B = c.B
        self.stage = torch.zeros(c.nbytes, dtype=torch.uint8, device=device)  # fixed address: a captured update could read it
        v = lambda name, dt: c.view(self.stage, name, dt)  # noqa: E731
        self.batch = ReplayBatch(indices=v("indices", torch.int64), weights=v("weights", torch.float32), obs=None, actions=v("actions", torch.int32).view(c.B, c.n),
                                 rewards=v("rewards", torch.float32).view(c.B, c.n), terminated=v("terminated", torch.float32).view(c.B, c.n))
        self.frame_off_all = v("rel_all", torch.int64).view(c.B, c.n + 1, c.W)
        self.frame_off_next = v("rel_next", torch.int64).view(c.B, c.n, c.W)
        self.obs_base = self.stage.data_ptr() + c.off["frames"][0]
        self.out = torch.zeros(16 + 12 * c.B, dtype=torch.uint8, device=device)  # write-back: [valid int64 | pad | indices | priorities]
        self._train_count_dev = train_count_dev

    def sample_items(self, d_step, uniforms=None, all_states=False):
        return self.batch

    def update(self, indices, priorities):
        B = self.B
        self.out[16 : 16 + 8 * B].view(torch.int64).copy_(indices)
        self.out[16 + 8 * B :].view(torch.float32).copy_(priorities)
        self._train_count_dev.add_(1)

    def is_warmup_needed(self) -> bool:
        return False

    # ---- what RainbowEngine(role="learner", learner_replay=...) asks of its replay besides the batch ----
    def count_updates_in(self, counter: torch.Tensor):
        self._train_count_dev = counter  # (`update` advances it: the write-back is this replay's last word on an update)

    def check_draws(self):
        pass

    def length(self) -> int:
        return 0


class ReplayRoleRainbow:
    def __init__(self, cfg, device: int, episode_len: int = 200, sync_interval: int = 16, prefetch: int = 5, updates: int = 1, env=None):
        from simple_distributed_rl_amd.device.rainbow import RainbowEngine
        from simple_distributed_rl_amd.device.replay import DeviceReplay

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert self.world >= 3, "three roles: learner (rank 0), replay (rank 1), actors (ranks 2..)"
        assert 0 <= prefetch <= 5, "the reference's prefetch queue holds 5 batches (play_mp_memory.py:595-621)"
        self.cfg, self.dev = cfg, torch.device(f"cuda:{device}")
        self.role = "learner" if self.rank == LEARNER else ("replay" if self.rank == REPLAY else "actor")
        self.n_actor_ranks = self.world - FIRST_ACTOR
        self.sync_interval, self.prefetch, self.updates = int(sync_interval), int(prefetch), int(updates)
        self.staged = dist.get_backend() == "gloo"  # test rigs: ranks share one GPU, tensors travel through the host
        E = cfg.n_envs
        H, W_ = cfg.obs_hw
        F, W, n, B = H * W_, cfg.window_length, cfg.multisteps, cfg.batch_size
        self.codec = BatchCodec(B, n, W, F)
        self.step_count, self._in_flight, self.env_steps_local, self.served, self.trained = 0, False, 0, 0, 0
        self._graphs = False
        self.weights_group = dist.new_group([LEARNER] + list(range(FIRST_ACTOR, self.world)))  # (every rank calls new_group)
        self.bus = TransitionBus(E, F, torch.uint8, self.dev, learner_rank=REPLAY, actor_ranks=range(FIRST_ACTOR, self.world), p2p=True)
        self.local = self.replay = None
        if self.role != "replay":
            pad = n + W
            small = dataclasses.replace(cfg, memory_capacity=E * 4, memory_warmup_size=1 << 62, seed=cfg.seed + 1_000_003 * self.rank,
                                        n_envs=E if self.role == "actor" else 8)
            if self.role == "learner":
                # the fast learner (round 6): one captured update per served batch -- online | target pass, TD / Huber / priorities in the backward's head kernel, Adam
                # inside the gradient launches -- on a replay that is a received message (`ServedBatch`); its priorities leave in the write-back message
                self.served_batch = ServedBatch(self.codec, self.dev)
                self.local = RainbowEngine(small, device, episode_len, ring_len=pad + 4, role="learner", learner_replay=self.served_batch, overlap=False)
            else:
                self.local = RainbowEngine(small, device, episode_len, ring_len=pad + 4, env=env, overlap=False)
            self.flat = flatten_parameters(self.local.q_online)
            for inf in (self.local.inf_actor, self.local.inf_online, self.local.inf_target):
                if inf is not None and inf.net is self.local.q_online:
                    inf.bind()
            if isinstance(self.local.optimizer, DeviceAdam):
                self.local.optimizer.bind()
            self._broadcast_weights()
            if self.local.fast:
                self.local._publish_out_of_band()  # packed filters / operand planes of the parameters' new home
        if self.role == "replay":
            total = self.n_actor_ranks * E
            self.replay = DeviceReplay(total, -(-cfg.memory_capacity // total) + n + W, F, W, n, cfg.n_actions, B, True, cfg.enable_reward_clip, cfg.memory_alpha,
                                       cfg.memory_beta_initial, cfg.memory_beta_steps, cfg.memory_epsilon, cfg.memory_warmup_size, cfg.seed, device,
                                       has_duplicate=cfg.memory_has_duplicate)
            xdev = "cpu" if self.staged else self.dev
            self.msg = [torch.zeros(self.codec.nbytes, dtype=torch.uint8, device=self.dev) for _ in range(2 * self.updates)]  # batches: built here, two lock-steps deep
            self.msg_tx = [torch.zeros(self.codec.nbytes, dtype=torch.uint8, device=xdev) for _ in range(2 * self.updates)] if self.staged else self.msg
            self.wb_rx = [torch.zeros(16 + 12 * B, dtype=torch.uint8, device=xdev) for _ in range(2 * self.updates)]
            self.wb_dev = torch.zeros(16 + 12 * B, dtype=torch.uint8, device=self.dev)
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)  # the learner's train count as far as this rank knows (beta schedule)
            self._works = []  # the group posted in the previous lock-step
        if self.role == "learner":
            xdev = "cpu" if self.staged else self.dev
            self._rx = []  # (serve lock-step, buffer) of received batches, in arrival order
            self._rx_bufs = [torch.zeros(self.codec.nbytes, dtype=torch.uint8, device=xdev) for _ in range((self.prefetch + 2) * self.updates)]
            self._rx_n = 0
            self.wb_tx = [torch.zeros(16 + 12 * B, dtype=torch.uint8, device=xdev) for _ in range(2 * self.updates)]
            self._posted = []  # (serve lock-step, buffer) of the receives in the group posted in the previous lock-step
            self._works = []
        self.check_headers = __import__("os").environ.get("SRLX_CHECK_HEADERS", "0") == "1"
        # a host synchronisation in front of every posted group (replay and learner rank).  Not needed: RCCL runs a group behind everything enqueued on the posting
        # thread's current stream, and tests/test_dist_stream_semantics_gpu.py::test_replay_gpu_role_under_stream_ordered_transfers runs the role without it
        self.host_sync = False
        self._total_envs = self.n_actor_ranks * E
        # first observations of every actor environment -> the replay rank's ring position 0 (a one-off exchange through staging slot 0; the records are not used)
        self.bus.enable_slots(2)
        if self.role == "actor":
            eng = self.local
            self.bus.send_begin(self.bus.pack(eng.actions, eng.env.rewards, eng.env.terminated, eng.env.done), eng.first_obs)
            self.bus.send_end()
        elif self.role == "replay":
            self.bus.recv_begin(0)
            self.bus.recv_end()
            self.replay.reset_all(self.bus.slot_obs[0])
        torch.cuda.synchronize(self.dev)

    # ---- helpers --------------------------------------------------------------------------------------------------------------
    def _broadcast_weights(self):
        if self.staged:
            host = self.flat.cpu()
            dist.broadcast(host, src=LEARNER, group=self.weights_group)
            if self.role == "actor":
                self.flat.copy_(host)
        else:
            dist.broadcast(self.flat, src=LEARNER, group=self.weights_group)

    def _warm_at(self, serve_step: int) -> bool:
        """Was the replay past its warm-up gate when it served the batches of lock-step `serve_step`?  (it has committed `serve_step` lock-steps by then)"""
        return serve_step >= 0 and min(self.cfg.memory_capacity, serve_step * self._total_envs) >= self.cfg.memory_warmup_size

    def _trained_from(self, learner_step: int) -> int:
        """The serve lock-step of the batch the learner trains on in its lock-step `learner_step` (negative: nothing has arrived yet)."""
        return learner_step - 1 - self.prefetch

    @staticmethod
    def _complete(works):
        for w in works:
            w.wait()

    # ---- the driver interface of device/mp_runner.py (what DistributedRainbow offers) ------------------------------------------------------
    @property
    def global_envs(self) -> int:
        return self.n_actor_ranks * self.cfg.n_envs

    def broadcast_weights(self):
        """learner -> actors now (mp_runner: after the learner rank has loaded the Runner's parameter)."""
        if self.role != "replay":
            self._broadcast_weights()

    def flush(self):
        self.finish()

    # ---- one lock-step ----------------------------------------------------------------------------------------------------------
    def step(self, updates=None):
        getattr(self, "_step_" + self.role)()
        self.step_count += 1
        if self.step_count % self.sync_interval == 0 and self.role != "replay":
            self._broadcast_weights()

    def _step_actor(self):
        eng = self.local
        q = eng._actor_net(None, None)
        if self._in_flight:
            self.bus.send_end()  # the previous slab's frames have left before the environments overwrite them
        eng._actor_select(q)
        eng.actor_commit()
        self.env_steps_local += self.cfg.n_envs
        env = eng.env
        self.bus.send_begin(self.bus.pack(eng.actions, env.rewards, env.terminated, env.done), env.next_obs)  # one record + the frames: ONE group of two sends
        self._in_flight = True

    def _commit_slab(self, slot: int):
        """The slab in staging slot `slot` -> ring + tree: one launch unpacks the actor ranks' records and commits their environments, one adds the new leaves."""
        rp, bus = self.replay, self.bus
        rp.commit_packed(bus.slot_scal[slot], self.cfg.n_envs, 0, bus.slot_obs[slot])
        rp.add_masked()
        rp.note_commit()

    def _step_replay(self):
        rp, c, U, s_now = self.replay, self.codec, self.updates, self.step_count
        if self._in_flight:  # the slab that travelled during the previous lock-step (straight into staging slot (s_now - 1) % 2)
            self.bus.recv_end()
            self._commit_slab((s_now - 1) % 2)
        self.bus.recv_begin(s_now % 2)
        self._in_flight = True
        # the group of the previous lock-step: its batches are out, the write-backs the learner produced during that lock-step are in -- apply them
        self._complete(self._works)
        if s_now >= 1 and self._warm_at(self._trained_from(s_now - 1)):
            B = c.B
            for u in range(U):
                buf = self.wb_rx[((s_now - 1) % 2) * U + u]
                wb = self.wb_dev
                wb.copy_(buf, non_blocking=True)
                if self.check_headers:
                    assert int(wb[:8].view(torch.int64).item()) == 1, "a write-back the schedule calls valid says it is not"
                rp.update(wb[16 : 16 + 8 * B].view(torch.int64), wb[16 + 8 * B :].view(torch.float32))
                self.step_dev.add_(1)
        # serve `updates` batches (built into this lock-step's half of the ring; the other half may still be read by the previous group's sends -- completed above)
        warm = self._warm_at(s_now)
        assert warm == (not rp.is_warmup_needed())
        ops = []
        for u in range(U):
            k = (s_now % 2) * U + u
            m = self.msg[k]
            c.view(m, "header", torch.int64)[0] = 1 if warm else 0
            if warm:
                b = rp.sample_items(self.step_dev, all_states=True)
                for name, src in (("indices", b.indices), ("weights", b.weights), ("actions", b.actions), ("rewards", b.rewards), ("terminated", b.terminated)):
                    c.view(m, name, src.dtype).copy_(src.reshape(-1))
                rel = c.view(m, "rel_all", torch.int64)
                N.check(rp.lib.srlx_pack_frames(N.c_p(rp.obs_base), N.tptr(rp.frame_off_all), c.rows, c.F, N.c_p(m.data_ptr() + c.off["frames"][0]), N.tptr(rel),
                                                N.torch_stream_ptr()))
                c.view(m, "rel_next", torch.int64).copy_(rel.view(c.B, c.n + 1, c.W)[:, 1:].reshape(-1))  # s_1..s_n are rows 1.. of every item
            if self.staged:
                self.msg_tx[k].copy_(m)
            ops.append(dist.P2POp(dist.irecv, self.wb_rx[k], LEARNER))
        for u in range(U):
            ops.append(dist.P2POp(dist.isend, self.msg_tx[(s_now % 2) * U + u], LEARNER))
        if not self.staged and self.host_sync:
            torch.cuda.current_stream(self.dev).synchronize()  # (RCCL orders a group behind everything on the CURRENT stream: the wait is optional, `host_sync`)
        self._works = dist.batch_isend_irecv(ops)
        self.served += U

    def _step_learner(self):
        eng, c, sb, U, s_now = self.local, self.codec, self.served_batch, self.updates, self.step_count
        # the group of the previous lock-step: that lock-step's batches have arrived, its write-backs are out
        self._complete(self._works)
        self._rx.extend(self._posted)
        ops, self._posted = [], []
        for u in range(U):  # this lock-step's receives
            buf = self._rx_bufs[self._rx_n % len(self._rx_bufs)]
            self._rx_n += 1
            self._posted.append((s_now, buf))
            ops.append(dist.P2POp(dist.irecv, buf, REPLAY))
        t_serve = self._trained_from(s_now)
        for u in range(U):
            out = self.wb_tx[(s_now % 2) * U + u]
            valid = False
            if t_serve >= 0:  # train on the batch that arrived `prefetch` lock-steps ago
                tag, buf = self._rx.pop(0)
                assert tag == t_serve
                valid = self._warm_at(t_serve)
                if valid:
                    sb.stage.copy_(buf, non_blocking=True)
                    if self.check_headers:
                        assert int(c.view(sb.stage, "header", torch.int64)[0].item()) == 1, "a batch the schedule calls warm says it is not"
                    eng.run_updates(1)  # (target sync and the update count inside: RainbowEngine.learner_step)
                    if not self._graphs and eng.fast:
                        eng.enable_lazy_capture()  # from the second update on: one graph (the staging buffer's address is fixed)
                        self._graphs = True
                    self.trained += 1
            sb.out[:8].view(torch.int64)[0] = 1 if valid else 0
            out.copy_(sb.out, non_blocking=not self.staged)
            ops.append(dist.P2POp(dist.isend, out, REPLAY))
        if not self.staged and self.host_sync:
            torch.cuda.current_stream(self.dev).synchronize()
        self._works = dist.batch_isend_irecv(ops)

    def finish(self):
        """Complete what is in flight: every message posted is matched (each lock-step moved the same number in both directions)."""
        if self.role == "actor" and self._in_flight:
            self.bus.send_end()
        if self.role == "replay":
            if self._in_flight:
                self.bus.recv_end()
                self._commit_slab((self.step_count - 1) % 2)
            self._complete(self._works)
        if self.role == "learner":
            self._complete(self._works)  # (the batches received last are never trained on)
        self._works = []
        self._in_flight = False
        torch.cuda.synchronize(self.dev)

    def info(self) -> dict:
        d = dict(role=self.role, steps=self.step_count, env_steps_local=self.env_steps_local)
        if self.role == "replay":
            d.update(memory=self.replay.length(), served=self.served, write_backs=int(self.step_dev.item()))
        if self.role == "learner":
            d.update(train_count=self.local.train_count, loss=float(self.local.loss.item()) if self.trained else None)
        return d
