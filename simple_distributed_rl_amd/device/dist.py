"""Multi-GPU actor/learner topology over RCCL (torch.distributed backend "nccl" on ROCm).

One process per GPU.  Actor ranks run E lock-stepped actors each against their own copy of the online
network; rank 0 owns the replay (frame ring + sum-tree for the environments of ALL actor ranks) and the
learner.  With `learner_acts=True` (default for 2 ranks) rank 0 is an actor rank as well; with
`learner_acts=False` (default from 4 ranks: BASELINE.json config 4, "7 actor GPUs + 1 learner GPU") it only
learns: its update runs beside the gather instead of beside its own actors, so the step time of the job is
the actor ranks' step, not rank 0's actor + commit + learner.  Per step:
    actors -> learner : one GROUP of point-to-point transfers (grouped ncclSend / ncclRecv) of fixed-size transition
                        slabs (next frame uint8 [E,F] + one packed record buffer: action / reward / terminated /
                        done) straight into the learner's HBM staging buffers, from where one commit kernel writes
                        them into the ring and the PER tree; a rank that only learns sends nothing
    learner -> actors : every `sync_interval` steps one broadcast of the flat float32 parameter buffer
                        (32 MB for the Atari network) that the actor networks alias (no unpack copy)

This replaces the reference's multiprocessing transport -- pickled+zlib'd items on a Manager queue with
back-pressure (srl/base/run/play_mp.py:76-118,248-286) and a pickled state_dict polled from a
Manager.Value board on a timer (play_mp.py:121-165,289-318) -- for the intra-node case.  xGMI is a
point-to-point mesh: the gather is 7 independent link transfers into rank 0 (7 MB per actor rank per
step at E=1024, far below the ~153 GB/s per link), the broadcast is a 32 MB fan-out.
`TransitionBus` only needs torch.distributed and tensors, so its protocol is tested on CPU with gloo
(tests/test_dist_cpu.py); the kernels around it are tested on the GPU.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from simple_distributed_rl_amd import _native as N
from simple_distributed_rl_amd.device.qnet import DeviceAdam


def rccl_options():
    """`pg_options` for `init_process_group("nccl", ...)`: the communicator's streams at HIGH priority.  HIP keeps one pool of hardware queues per priority level; at
    ProcessGroupNCCL's default (normal) the receives of a learner rank share the normal pool with the branches of its captured update, and the period under the
    transfers' stream semantics is 1.35 x the bare one; at high priority 1.12 x (bench.py --roles-only: `learner_rank.fabric_ms_per_period`, same box)."""
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    return opts


class TransitionBus:
    """Fixed-size per-step transition exchange and parameter fan-out between ranks."""

    def __init__(self, n_envs_local: int, obs_elems: int, obs_dtype: torch.dtype, device: torch.device, group=None, learner_rank: int = 0,
                 always_collective: bool = False, extra_floats: int = 0, actor_ranks=None, p2p: Optional[bool] = None):
        """extra_floats: further float32 fields per environment that ride in the packed record buffer (Agent57_light's intrinsic reward,
        arm, previous action / rewards).  actor_ranks: the ranks that have transitions to ship (default: all).  p2p (default: whenever there is
        more than one rank): the exchange is ONE group of point-to-point transfers -- every actor rank sends its two buffers to the learner rank,
        which posts the matching receives straight into its staging buffers (`batch_isend_irecv` = grouped ncclSend / ncclRecv on RCCL); a rank
        that only learns sends nothing, and the learner rank's own transitions (when it also acts) are a copy inside its HBM.  p2p=False keeps the
        gather collective (every rank contributes an equal part, a learner-only rank a placeholder)."""
        self.E, self.F, self.K = n_envs_local, obs_elems, int(extra_floats)
        self.always_collective = always_collective  # run the collectives even at world size 1 (transport tests on a 1-GPU box)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.learner_rank = learner_rank
        self.device = device
        self.is_learner = self.rank == learner_rank
        self.actor_ranks = list(range(self.world)) if actor_ranks is None else [int(r) for r in actor_ranks]
        self.contributes = self.rank in self.actor_ranks
        self.p2p = (self.world > 1) if p2p is None else bool(p2p)  # (p2p=False: the gather collective)
        self.sent_bytes = self.recv_bytes = 0  # what this rank put on / took off the wire (tests: a learner-only rank sends nothing)
        self._pending, self._direct, self._keep, self._staged_in = [], None, None, []
        if self.is_learner:
            T = self.world * self.E
            self.g_actions = torch.zeros(T, dtype=torch.int32, device=device)
            self.g_rewards = torch.zeros(T, dtype=torch.float32, device=device)
            self.g_terminated = torch.zeros(T, dtype=torch.uint8, device=device)
            self.g_done = torch.zeros(T, dtype=torch.uint8, device=device)
            self.g_next_obs = torch.zeros((T, obs_elems), dtype=obs_dtype, device=device)
            self.g_scal = torch.zeros((self.world, (10 + 4 * self.K) * self.E), dtype=torch.uint8, device=device)  # packed scalar records
            self.g_extra = torch.zeros((T, self.K), dtype=torch.float32, device=device) if self.K else None

    def _views(self, buf) -> Optional[List[torch.Tensor]]:
        if not self.is_learner:
            return None
        return [buf[r * self.E : (r + 1) * self.E] for r in range(self.world)]

    def push_begin(self, actions, rewards, terminated, done, next_obs, extra=None):
        """Start the per-step exchange: every rank contributes its E transitions, the learner rank receives them in rank order
        (env index = rank * E + local index).  On RCCL the two gathers are issued `async_op=True`: they run on the communicator's
        own stream once everything enqueued on the current stream so far has finished, and nothing waits for them until
        `push_end` -- an actor rank runs its NEXT network pass meanwhile (the reference's actors likewise keep playing while
        their items sit in the queue, srl/base/run/play_mp.py:76-118).  The caller must not overwrite the five tensors
        before `push_end`."""
        self._pending = []
        self._direct = None
        assert (extra is not None) == (self.K > 0)
        if self.world == 1 and not self.always_collective:
            self._direct = (actions, rewards, terminated, done, next_obs) + ((extra.clone(),) if self.K else ())  # (`extra` is assembled per call by the caller: keep a copy)
            return
        # two collectives per step: the frames, and ONE packed record buffer for the four scalar fields
        # ([actions 4E | rewards 4E | terminated E | done E] bytes per rank) that the learner unpacks with strided copies
        fields = [actions.contiguous().view(torch.uint8), rewards.contiguous().view(torch.uint8), terminated.contiguous().view(torch.uint8),
                  done.contiguous().view(torch.uint8)]
        if self.K:
            fields.append(extra.to(torch.float32).contiguous().view(-1).view(torch.uint8))  # [E][K] row-major
        scal = torch.cat(fields)
        self._keep = (scal, next_obs)  # inputs stay alive until the collectives are done
        staged = dist.get_backend(self.group) == "gloo" and actions.is_cuda  # test rigs: ranks sharing one GPU
        if self.p2p:
            ops, self._staged_in = [], []
            for t, name in ((scal, "g_scal"), (next_obs.contiguous(), "g_next_obs")):
                if self.is_learner:
                    buf = getattr(self, name)
                    views = [buf[r] for r in range(self.world)] if name == "g_scal" else self._views(buf)
                    for r in self.actor_ranks:
                        if r == self.rank:
                            views[r].view(-1).copy_(t.view(-1).view(buf.dtype))  # this rank's own transitions never leave its HBM
                            continue
                        dst = torch.empty(views[r].shape, dtype=views[r].dtype, device="cpu") if staged else views[r]
                        if staged:
                            self._staged_in.append((dst, views[r]))
                        ops.append(dist.P2POp(dist.irecv, dst, r, self.group))
                        self.recv_bytes += dst.numel() * dst.element_size()
                elif self.contributes:
                    src = t.cpu() if staged else t
                    self._keep = self._keep + (src,)
                    ops.append(dist.P2POp(dist.isend, src, self.learner_rank, self.group))
                    self.sent_bytes += src.numel() * src.element_size()
            self._pending = list(dist.batch_isend_irecv(ops)) if ops else []
            return
        for t, name in ((scal, "g_scal"), (next_obs.contiguous(), "g_next_obs")):
            if staged:
                parts = [torch.empty_like(t, device="cpu") for _ in range(self.world)] if self.is_learner else None
                dist.gather(t.cpu(), parts, dst=self.learner_rank, group=self.group)
                if self.is_learner:
                    getattr(self, name).view(-1).copy_(torch.cat([p.view(-1) for p in parts]).to(self.device).view(getattr(self, name).dtype))
            else:
                views = None
                if self.is_learner:
                    buf = getattr(self, name)
                    views = [buf[r] for r in range(self.world)] if name == "g_scal" else self._views(buf)
                self._pending.append(dist.gather(t, views, dst=self.learner_rank, group=self.group, async_op=True))

    def push_end(self):
        """The current stream waits for the exchange started by `push_begin` (a stream-level wait on RCCL, the host does not
        block).  Returns the gathered tensors on the learner rank, None elsewhere."""
        if self._direct is not None:
            out, self._direct = self._direct, None
            return out
        for work in self._pending:
            work.wait()
        self._pending = []
        for host, view in self._staged_in:
            view.copy_(host.to(self.device))
        self._staged_in = []
        self._keep = None
        if self.is_learner:
            E, g = self.E, self.g_scal
            self.g_actions.view(torch.uint8).view(self.world, 4 * E).copy_(g[:, : 4 * E])
            self.g_rewards.view(torch.uint8).view(self.world, 4 * E).copy_(g[:, 4 * E : 8 * E])
            self.g_terminated.view(self.world, E).copy_(g[:, 8 * E : 9 * E])
            self.g_done.view(self.world, E).copy_(g[:, 9 * E : 10 * E])
            if self.K:
                self.g_extra.view(torch.uint8).view(self.world, 4 * self.K * E).copy_(g[:, 10 * E :])
                return self.g_actions, self.g_rewards, self.g_terminated, self.g_done, self.g_next_obs, self.g_extra
            return self.g_actions, self.g_rewards, self.g_terminated, self.g_done, self.g_next_obs
        return None

    def push(self, actions, rewards, terminated, done, next_obs, extra=None):
        """`push_begin` + `push_end` back to back."""
        self.push_begin(actions, rewards, terminated, done, next_obs, extra)
        return self.push_end()

    # ---- slot API (DistributedRainbow): packed records + frames, the learner's staging buffers rotate so that a slab can be committed a lock-step or two after it
    #      arrived while the next one is being received ----------------------------------------------------------------------------------------------------------
    def enable_slots(self, slots: int):
        """Learner rank: `slots` staging buffers of (packed records uint8 [actor ranks][record bytes], frames [actor ranks x E][F]); only actor ranks have rows."""
        self.rec_bytes = (10 + 4 * self.K) * self.E
        self.row_of = {r: i for i, r in enumerate(self.actor_ranks)}
        if self.is_learner:
            R = len(self.actor_ranks)
            dt = self.g_next_obs.dtype if hasattr(self, "g_next_obs") else torch.uint8
            self.slot_scal = [torch.zeros((R, self.rec_bytes), dtype=torch.uint8, device=self.device) for _ in range(slots)]
            self.slot_obs = [torch.zeros((R * self.E, self.F), dtype=dt, device=self.device) for _ in range(slots)]
        self._staged = dist.is_initialized() and self.world > 1 and dist.get_backend(self.group) == "gloo" and self.device.type == "cuda"  # test rigs: ranks sharing one GPU

    def pack(self, actions, rewards, terminated, done, extra=None) -> torch.Tensor:
        """One rank's record: [action int32 x E | reward float32 x E | terminated u8 x E | done u8 x E | extra float32 x E x K] as uint8 (a fresh tensor: ONE launch)."""
        assert (extra is not None) == (self.K > 0)
        fields = [actions.view(torch.uint8), rewards.view(torch.uint8), terminated.view(torch.uint8), done.view(torch.uint8)]
        if self.K:
            fields.append(extra.to(torch.float32).contiguous().view(-1).view(torch.uint8))  # [E][K] row-major
        return torch.cat(fields)

    def send_begin(self, scal: torch.Tensor, next_obs: torch.Tensor):
        """Actor rank that is not the learner: ONE group of two sends (record, frames) to the learner rank; nothing may overwrite `next_obs` before `send_end`."""
        if self.world == 1 or self.is_learner or not self.contributes:
            return
        src = [scal, next_obs.contiguous()]
        if self._staged:
            src = [t.cpu() for t in src]
        self._keep = src
        ops = [dist.P2POp(dist.isend, t, self.learner_rank, self.group) for t in src]
        self.sent_bytes += sum(t.numel() * t.element_size() for t in src)
        self._pending = list(dist.batch_isend_irecv(ops))

    def send_end(self):
        """The current stream waits for the sends `send_begin` started (a stream-level wait on RCCL)."""
        for work in self._pending:
            work.wait()
        self._pending, self._keep = [], None

    def recv_begin(self, slot: int):
        """Learner rank: ONE group of receives, two per other actor rank, straight into staging slot `slot`."""
        self._staged_in = []
        if self.world == 1 or not self.is_learner:
            return
        ops = []
        for r in self.actor_ranks:
            if r == self.rank:
                continue
            i = self.row_of[r]
            for view in (self.slot_scal[slot][i], self.slot_obs[slot][i * self.E : (i + 1) * self.E]):
                dst = torch.empty(view.shape, dtype=view.dtype, device="cpu") if self._staged else view
                if self._staged:
                    self._staged_in.append((dst, view))
                ops.append(dist.P2POp(dist.irecv, dst, r, self.group))
                self.recv_bytes += dst.numel() * dst.element_size()
        self._pending = list(dist.batch_isend_irecv(ops)) if ops else []

    def recv_end(self):
        for work in self._pending:
            work.wait()
        self._pending = []
        for host, view in self._staged_in:
            view.copy_(host.to(self.device))
        self._staged_in = []

    def put_own(self, slot: int, scal: torch.Tensor, next_obs: torch.Tensor):
        """Learner rank that also acts: its own transitions never leave its HBM (two device copies into its rows of the slot)."""
        i = self.row_of[self.rank]
        self.slot_scal[slot][i].copy_(scal)
        self.slot_obs[slot][i * self.E : (i + 1) * self.E].copy_(next_obs.view(self.E, self.F))

    def broadcast_params(self, flat: torch.Tensor):
        if self.world <= 1 and not self.always_collective:
            return
        if dist.get_backend(self.group) == "gloo" and flat.is_cuda:
            host = flat.cpu()
            dist.broadcast(host, src=self.learner_rank, group=self.group)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=self.learner_rank, group=self.group)


def flatten_parameters(module: torch.nn.Module) -> torch.Tensor:
    """Re-homes every parameter of `module` into ONE contiguous float32 buffer and returns it; the
    parameters become views of the buffer, so a broadcast into it updates the network in place."""
    params = list(module.parameters())
    align = 64  # floats: every parameter starts on a 256-byte boundary (vector loads, srlx_adam_step's float4 path)
    total = sum(-(-p.numel() // align) * align for p in params)
    flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        # keep each parameter's own dense memory format (channels_last conv weights stay channels_last)
        view = flat[off : off + n].as_strided(p.size(), p.stride())
        view.copy_(p.data)
        p.data = view
        off += -(-n // align) * align
    return flat


class DistributedRainbow:
    """world ranks x E actors, learner + replay on rank 0 (BASELINE.json config 4 topology applied to Rainbow; the reference: srl/base/run/play_mp.py:121-165 actor
    loop, :248-318 trainer + drain thread, :540-571 process layout).

    Roles.  An ACTOR rank runs `RainbowEngine(role="actor")`: the round-4 policy pass (fused policy head, parameter sets published out of band after every
    weight broadcast, CU-filling first dense layer), one-launch environments, one-launch commit into a short local ring that only stacks frames; per lock-step it
    ships ONE packed record + its frames to the learner rank (a group of two sends) while its next pass runs.  The LEARNER rank owns ring + tree for the actor
    ranks' environments.  With `learner_acts` it runs the single-GPU lock-step (update graph beside its own actors' pass) on top; without, only updates.

    The exchange is one lock-step late by construction (slab t arrives while the learner works on t - 1), and the learner commits a slab INSIDE its update: the
    update's draw samples the tree first (one add older than a commit-first order would show it), the ring commit + tree add of the arrived slab run on a side
    stream beside the update's network passes, and the priority write-back waits for them -- the tree sees draw, add, write-back in that order
    (tests/test_dist_gpu.py replays it against the oracle), and the learner rank's period is max(update, receive) instead of update + commit + add.
    With actor-side initial priorities (cfg.actor_initial_priority, rainbow.py:389-398) the estimates of the items a slab completes travel in the NEXT slab (their last
    state is evaluated by the next pass); the learner then holds a slab back one more lock-step and commits ring and tree together, so no leaf ever carries the
    priority of the item it replaced.
    """

    def __init__(self, cfg, device: int, episode_len: int = 200, sync_interval: int = 16, overlap: bool = True, always_collective: bool = False,
                 learner_acts: Optional[bool] = None, env=None, actor_stream: Optional[str] = None):
        import dataclasses

        from simple_distributed_rl_amd.device.rainbow import RainbowEngine
        from simple_distributed_rl_amd.device.replay import DeviceReplay

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.cfg = cfg
        self.dev = torch.device(f"cuda:{device}")
        self.sync_interval = int(sync_interval)
        self.is_learner = self.rank == 0
        # does the learner rank run actors too?  (a 1-rank group has nobody else to act)
        self.learner_acts = (self.world < 4) if learner_acts is None else bool(learner_acts)
        if self.world == 1:
            self.learner_acts = True
        self.acts = self.learner_acts or not self.is_learner  # this rank runs actors
        self.first_actor_rank = 0 if self.learner_acts else 1
        self.n_actor_ranks = self.world - self.first_actor_rank
        E = self.E = cfg.n_envs
        H, W_ = cfg.obs_hw
        pad = cfg.multisteps + cfg.window_length
        self.actor_priority = bool(cfg.actor_initial_priority)
        self.K = 1 if self.actor_priority else 0
        self.slots = 3 if self.actor_priority else 2
        # every rank: a short local ring, only for frame stacking of its own envs (no PER use); every rank draws its environments, its exploration and its padding
        # actions from its OWN stream: ranks acting on the same broadcast weights must not produce byte-identical transitions (the learner's replay seed stays cfg.seed)
        local_cfg = dataclasses.replace(cfg, memory_capacity=E * 4, memory_warmup_size=1 << 62, seed=cfg.seed + 1_000_003 * self.rank)
        self.bus = TransitionBus(E, H * W_, torch.uint8, self.dev, always_collective=always_collective, actor_ranks=range(self.first_actor_rank, self.world),
                                 extra_floats=self.K)
        self.bus.enable_slots(self.slots)
        if self.is_learner:
            total = self.n_actor_ranks * E
            ring_len = -(-cfg.memory_capacity // total) + pad
            self.replay = DeviceReplay(
                total, ring_len, H * W_, cfg.window_length, cfg.multisteps, cfg.n_actions, cfg.batch_size, True, cfg.enable_reward_clip,
                cfg.memory_alpha, cfg.memory_beta_initial, cfg.memory_beta_steps, cfg.memory_epsilon, cfg.memory_warmup_size, cfg.seed, device,
                has_duplicate=cfg.memory_has_duplicate,
            )
            self.replay.enable_deferred_advance()  # every ring commit of the global replay is followed by its tree add (`_ingest_fn`): the add moves the position
            self.est_buf = torch.full((total,), -1.0, dtype=torch.float32, device=self.dev)
            role = "both" if self.acts else "learner"
            self.local = RainbowEngine(local_cfg, device, episode_len, ring_len=pad + 4, env=env, overlap=overlap and self.acts, role=role, learner_replay=self.replay,
                                       actor_stream=actor_stream if self.acts else None)
        else:
            self.local = RainbowEngine(local_cfg, device, episode_len, ring_len=pad + 4, env=env, role="actor")
            self.replay = self.local.replay
            self.local.before_env = self.bus.send_end  # the previous slab's frames must have left before the environments overwrite them
        self.overlap = self.local.overlap
        self.flat = flatten_parameters(self.local.q_online)
        # the parameters moved: point the kernels (and the fused Adam) at their new home
        for inf in (self.local.inf_actor, self.local.inf_online, self.local.inf_target):
            if inf is not None and (inf.net is self.local.q_online):
                inf.bind()
        if isinstance(self.local.optimizer, DeviceAdam):
            self.local.optimizer.bind()
        self._minus_one = torch.full((E, 1), -1.0, dtype=torch.float32, device=self.dev)
        self.step_count = 0
        self._next_ingest = 0  # the next slab (= lock-step index) the learner rank has not committed yet
        self.env_steps_local = 0  # environment steps taken by THIS rank's actors
        self.bus.broadcast_params(self.flat)
        if self.local.fast:
            self.local._publish_out_of_band()
        elif self.acts:
            self.local.inf_actor.weights_changed()
        # first observations of every env -> global ring position 0 (a one-off synchronous exchange through slot 0; the records are not used)
        eng = self.local
        scal = self.bus.pack(eng.actions, eng.env.rewards, eng.env.terminated, eng.env.done, self._minus_one if self.K else None)
        if self.is_learner:
            self.bus.recv_begin(0)
            if self.acts:
                self.bus.put_own(0, scal, eng.first_obs)
            self.bus.recv_end()
            self.replay.reset_all(self.bus.slot_obs[0])
        else:
            self.bus.send_begin(scal, eng.first_obs)
            self.bus.send_end()
        torch.cuda.synchronize(self.dev)

    # ---- learner rank: committing arrived slabs -----------------------------------------------------------------------------------------------------------------
    def _ingest_fn(self, j: int, with_est: bool):
        """The launches that commit slab j (ring + tree): staging slot j % slots, estimates (if any) out of the slab behind it."""
        rp, bus, E, K = self.replay, self.bus, self.E, self.K
        a = j % self.slots
        if not self.actor_priority:
            def fn():
                rp.commit_packed(bus.slot_scal[a], E, K, bus.slot_obs[a])
                rp.add_masked()
        else:
            est_src = bus.slot_scal[(j + 1) % self.slots] if with_est else None

            def fn():
                rp.commit_packed(bus.slot_scal[a], E, K, bus.slot_obs[a], est_records=est_src, est_out=self.est_buf)
                rp.add_estimates(self.est_buf)
        return (a, with_est), fn

    def _ingest_ready(self, k: int):
        """Slab to commit during lock-step k, or None: slab j has arrived when lock-step j is over; with estimates it also needs slab j + 1."""
        j = self._next_ingest
        return j if j <= k - (2 if self.actor_priority else 1) else None

    def _extra(self, random_policy: bool):
        if not self.actor_priority:
            return None
        est = None if random_policy else self.local.actor_td_estimates()
        return self._minus_one if est is None else est.view(-1, 1)

    def _act(self, events, random_policy: bool):
        eng = self.local
        if random_policy:
            eng.random_front()
            extra = self._extra(True)
        else:
            eng.actor_front(events)
            extra = self._extra(False)  # estimates for the items this rank committed one lock-step ago (their last state has just been evaluated)
        eng.actor_commit()  # the local ring (frame stacking of this rank's environments)
        self.env_steps_local += self.E
        env = eng.env
        return self.bus.pack(eng.actions, env.rewards, env.terminated, env.done, extra), env.next_obs

    def step(self, learner_updates: int = 1, events=None, random_policy: bool = False):
        """One lock-step of the whole job.  Every rank issues exactly one group of point-to-point transfers per lock-step (and every `sync_interval` lock-steps the
        parameter broadcast behind it), in the same order everywhere."""
        eng, bus, k = self.local, self.bus, self.step_count
        U = 0 if random_policy else learner_updates
        if self.is_learner:
            bus.recv_begin(k % self.slots)  # slab k lands while this lock-step runs
            j = self._ingest_ready(k)
            if j is not None:
                eng.ingest = self._ingest_fn(j, with_est=True)
            if self.acts:
                if eng.overlap:
                    eng.fork_learner(U)  # the update (and the slab's commit inside it) beside this rank's own actors
                scal, obs = self._act(events, random_policy)
                bus.put_own(k % self.slots, scal, obs)
                if eng.overlap:
                    eng.join_learner()
                else:
                    self._learn_inline(U)
                bus.recv_end()
                if eng.overlap:
                    eng.refresh_actor_copy()
            else:
                if events is not None:
                    events[0].record()
                    events[1].record()
                self._learn_inline(U)
                bus.recv_end()
            if j is not None:  # (after the updates: their warm-up gate saw the replay as the draw did)
                self.replay.note_commit()
                self._next_ingest = j + 1
            if getattr(self, "_capture_pending", False) and eng.train_count > 0:
                self._capture_pending = False
                eng.join_learner()
                eng.enable_lazy_capture()
        else:
            scal, obs = self._act(events, random_policy)  # (bus.send_end() of the previous slab sits between the policy pass and the environments)
            bus.send_begin(scal, obs)
        self.step_count += 1
        if self.step_count % self.sync_interval == 0:
            if self.is_learner:
                eng.join_learner()  # broadcast consistent weights: not while Adam is writing them
            bus.broadcast_params(self.flat)
            if not self.is_learner:
                eng.on_weights_broadcast()

    def _learn_inline(self, updates: int):
        """Updates that do not run beside this rank's own actors (a learner-only rank; a learner rank without overlap)."""
        self.local.run_updates(updates)

    def flush(self):
        """Commit the slabs that have arrived and are still staged (end of a run / of the filling phase); the last one of a run with actor-side priorities has no
        estimates behind it and enters at max_priority."""
        if not self.is_learner:
            self.bus.send_end()
            torch.cuda.synchronize(self.dev)
            return
        self.local.join_learner()
        while self._next_ingest < self.step_count:
            j = self._next_ingest
            self._ingest_fn(j, with_est=j + 1 < self.step_count)[1]()
            self.replay.note_commit()
            self._next_ingest = j + 1
        torch.cuda.synchronize(self.dev)

    def prefill(self):
        steps = 0
        if self.is_learner:
            steps = self.replay.item_len + self.cfg.multisteps - 1
        t = torch.tensor([steps], dtype=torch.int64)
        if dist.get_backend() != "gloo":
            t = t.to(self.dev)
        dist.broadcast(t, src=0)
        for _ in range(int(t.item())):
            self.step(0, random_policy=True)
        self.flush()
        if self.is_learner:
            g = torch.Generator(device=self.dev)
            g.manual_seed(self.cfg.seed + 1)
            pri = torch.rand(self.replay.capacity, dtype=torch.float32, device=self.dev, generator=g)
            N.check(self.replay.lib.srlx_per_set_range(self.replay.h_per, 0, self.replay.capacity, N.tptr(pri), N.PRIO_F32, 1, N.torch_stream_ptr()))
        torch.cuda.synchronize(self.dev)

    def capture_graphs(self):
        """The actors' launches stay eager; the learner rank's update is captured per variant (set it publishes into x staging slot it commits) the first time
        each runs."""
        if self.is_learner:
            # a variant's first run allocates inside the library (packed filters, scratch, a side stream, event records): that must not happen inside a capture --
            # before the learner's first (eager) update the switch waits for it (`step`)
            if self.local.train_count > 0:
                self.local.enable_lazy_capture()
            else:
                self._capture_pending = True

    def actor_forward_flops(self):
        return self.local.actor_forward_flops()

    def conv_gemm_flops(self, with_conv1: bool = False):
        return self.local.conv_gemm_flops(with_conv1)

    @property
    def fused_convs(self):
        return self.local.fused_convs

    @property
    def mfma(self):
        return self.local.mfma

    @property
    def global_envs(self) -> int:
        """Environments stepped per lock-step over the whole job."""
        return self.n_actor_ranks * self.cfg.n_envs

    def info(self):
        if self.is_learner:
            return self.local.info()
        return dict(loss=float("nan"), train_count=0, sync=0, memory=0)


class DistributedAgent57LightGeneral:
    """The round-3 form of the configs[3] job, kept for the geometries the all-libsrlx engine does not cover (`DistributedAgent57Light` below the class is what
    84 x 84 x 4 configs get: mp_runner picks): image trunks in libsrlx, dense tails and optimisers in torch (device/agent57_light.py), live tensors exchanged with
    `TransitionBus.push_begin / push_end`.
    Agent57_light on `world` ranks -- actor ranks x E environments, learner + global replay on rank 0 (7 actor
    GPUs + 1 learner GPU from 4 ranks up; with fewer ranks rank 0 also acts).  Every rank runs an `Agent57LightEngine` on a short local ring
    (frame stacking, its environments' episodic memories and UCB controllers: the intrinsic reward is an ACTOR-side quantity in the
    reference too, agent57_light.py:383-391).  Per lock-step an actor rank ships next frame, action, reward, flags and the five UVFA /
    intrinsic fields of its E environments (`TransitionBus` with `extra_floats=5`); rank 0 commits them to the global ring + tree + field
    arrays and trains; every `sync_interval` lock-steps the five online networks travel back as ONE flat broadcast.  The reference moves
    the same information as pickled 11-field items on a queue and a pickled list of five state_dicts on a timer
    (srl/base/run/play_mp.py:76-118,289-318; model_torch.py:148-156)."""

    FIELDS = 5  # r_int, arm, prev_action, prev_r_ext, prev_r_int

    def __init__(self, rl_config, n_envs: int, device: int, episode_len: int = 200, sync_interval: int = 16, learner_acts: Optional[bool] = None, seed: int = 0,
                 env=None, parameter=None, always_collective: bool = False):
        import copy

        from simple_distributed_rl_amd.device.agent57_light import Agent57LightEngine
        from simple_distributed_rl_amd.device.replay import DeviceReplay

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dev = torch.device(f"cuda:{device}")
        self.sync_interval = int(sync_interval)
        self.is_learner = self.rank == 0
        self.learner_acts = (self.world < 4) if learner_acts is None else bool(learner_acts)
        if self.world == 1:
            self.learner_acts = True
        self.acts = self.learner_acts or not self.is_learner
        self.first_actor_rank = 0 if self.learner_acts else 1
        self.n_actor_ranks = self.world - self.first_actor_rank
        E = self.E = int(n_envs)
        self.cfg = rl_config
        local_cfg = copy.deepcopy(rl_config)
        local_cfg.memory.capacity, local_cfg.memory.warmup_size = E * 4, 1 << 60  # the local ring only stacks frames; nobody samples it
        self.local = Agent57LightEngine(local_cfg, E, device, episode_len, seed=seed + 1_000_003 * self.rank, env=env, parameter=parameter,
                                        ring_len=1 + rl_config.window_length + 4)
        p = self.local.parameter
        self.nets = torch.nn.ModuleList([p.q_ext_online, p.q_int_online, p.emb_network, p.lifelong_target, p.lifelong_train])  # model_torch.py:148-156
        self.flat = flatten_parameters(self.nets)
        H, W_ = self.local.hw
        self.bus = TransitionBus(E, H * W_, torch.uint8, self.dev, always_collective=always_collective, extra_floats=self.FIELDS,
                                 actor_ranks=range(self.first_actor_rank, self.world))
        self.step_count, self._in_flight, self.env_steps_local = 0, False, 0
        if self.is_learner:
            total = self.n_actor_ranks * E
            mem = rl_config.memory
            kw = mem.kwargs if mem.name != "ReplayBuffer" else {}
            ring_len = -(-mem.capacity // total) + 1 + rl_config.window_length
            self.replay = DeviceReplay(total, ring_len, H * W_, rl_config.window_length, 1, self.local.A, rl_config.batch_size, True, False,
                                       float(kw.get("alpha", 0.0)), float(kw.get("beta_initial", 0.4)), int(kw.get("beta_steps", 1_000_000)),
                                       float(kw.get("epsilon", 1e-4)), mem.warmup_size, seed, device)
            L = self.replay.L
            self.x = torch.zeros((L, total, self.FIELDS), dtype=torch.float32, device=self.dev)
            B = rl_config.batch_size
            self.loc_env = torch.zeros(B, dtype=torch.int64, device=self.dev)
            self.loc_slot = torch.zeros(B, dtype=torch.int64, device=self.dev)
            self.train_count_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
        else:
            self.replay = self.local.replay
        self.bus.broadcast_params(self.flat)
        gathered = self.bus.push(self.local.actions, self.local.env.rewards, self.local.env.terminated, self.local.env.done, self.local.first_obs,
                                 torch.zeros((E, self.FIELDS), dtype=torch.float32, device=self.dev))
        if self.is_learner:
            self.replay.reset_all(self._actor_rows(gathered)[4])

    def _actor_rows(self, gathered):
        if self.first_actor_rank == 0:
            return gathered
        k = self.first_actor_rank * self.E
        return tuple(t[k:] for t in gathered)

    @property
    def global_envs(self) -> int:
        return self.n_actor_ranks * self.E

    @property
    def train_count(self) -> int:
        return self.local.learner.train_count

    def _slab(self):
        eng = self.local
        slot = (eng.replay._steps_committed - 1) % eng.L  # the slot the lock-step just taken was written to
        extra = torch.stack([eng.x_r_int[slot], eng.x_actor[slot].float(), eng.x_prev_action[slot].float(), eng.x_prev_r_ext[slot], eng.x_prev_r_int[slot]], dim=1)
        env = eng.env
        return eng.actions, env.rewards, env.terminated, env.done, env.next_obs, extra

    def _commit(self, gathered):
        a, r, t, d, obs, extra = self._actor_rows(gathered)
        slot = self.replay._steps_committed % self.replay.L
        self.x[slot] = extra
        self.replay.commit(a, r, t, d, obs)

    def learner_step(self) -> bool:
        rp, eng = self.replay, self.local
        if rp.is_warmup_needed():
            return False
        hand = eng._ltrunks is not None  # the learner's image blocks through the hand-written trunks, straight from the global ring
        b = rp.sample_items(self.train_count_dev, all_states=True) if hand else rp.sample(self.train_count_dev)
        N.check(rp.lib.srlx_store_locate(rp.h_store, rp.B, N.tptr(b.indices), N.tptr(self.loc_env), N.tptr(self.loc_slot), None, N.torch_stream_ptr()))
        x = self.x[self.loc_slot, self.loc_env]  # [B][5]
        obs = None if hand else b.obs.view(rp.B, 2, eng.Wn, *eng.hw)
        pri = eng.learner.update_networks(None if hand else obs[:, 0], None if hand else obs[:, 1], b.actions.view(-1), b.rewards.view(-1), x[:, 0].contiguous(),
                                          1.0 - b.terminated.view(-1), x[:, 2].long(), x[:, 3].contiguous(), x[:, 4].contiguous(), x[:, 1].long(), b.weights,
                                          features=eng.learner_features(rp) if hand else None)
        eng.learner.after_update()
        rp.update(b.indices, pri)
        self.train_count_dev.add_(1)
        return True

    def step(self, learner_updates: int = 1, events=None):
        """One lock-step of the job, pipelined over the exchange like DistributedRainbow.step: the slab of lock-step t travels while the
        actor ranks play lock-step t+1; rank 0 commits it at the start of its next call and trains while the next exchange is in flight."""
        # Order (as DistributedRainbow.step): network pass -> push_end(t-1) -> commit(t-1) -> selection + environments -> push_begin(t).  The exchange
        # in flight holds the LIVE tensors of lock-step t-1 (actions, rewards, flags, next_obs -- uncopied on the world-1 direct path and on RCCL's
        # asynchronous gather), so nothing may overwrite them before push_end; the network pass only reads the ring and the per-lane UVFA state.
        net = None
        if self.acts:
            if events is not None:
                events[0].record()
            net = self.local.actor_net()
        gathered = self.bus.push_end() if self._in_flight else None
        self._in_flight = False
        if self.is_learner and gathered is not None:
            self._commit(gathered)
        if self.acts:
            self.local.actor_rest(*net)
            if events is not None:
                events[1].record()
            self.env_steps_local += self.E
        slab = self._slab() if self.acts else (self.local.actions, self.local.env.rewards, self.local.env.terminated, self.local.env.done, self.local.first_obs,
                                               torch.zeros((self.E, self.FIELDS), dtype=torch.float32, device=self.dev))
        self.bus.push_begin(*slab)
        self._in_flight = True
        if self.is_learner:
            for _ in range(learner_updates):
                self.learner_step()
        self.step_count += 1
        if self.step_count % self.sync_interval == 0:
            self.bus.broadcast_params(self.flat)

    def flush(self):
        if self._in_flight:
            gathered = self.bus.push_end()
            self._in_flight = False
            if self.is_learner and gathered is not None:
                self._commit(gathered)
        torch.cuda.synchronize(self.dev)

    def info(self):
        d = dict(train_count=self.train_count, memory=self.replay.length())
        if self.is_learner and self.train_count > 0:
            d.update(self.local.learner.losses())
        return d


class DistributedAgent57Light:
    """BASELINE.json configs[3]: Agent57_light on `world` ranks -- actor ranks x E environments, learner + global replay on rank 0 (7 actor GPUs + 1 learner GPU from
    4 ranks up; with fewer ranks rank 0 also acts) -- on the all-libsrlx engine (device/agent57_fast.py) and the slot exchange of `DistributedRainbow` (round 6; the
    reference: srl/base/run/play_mp.py:121-165 actor loop, :248-318 trainer + drain thread, :540-571 process layout; the actor-side intrinsic reward of
    agent57_light.py:383-391).

    An ACTOR rank runs `Agent57LightFastEngine(role="actor")`: its five networks' passes read parameter sets published out of band after every weight broadcast, its
    environments' episodic memories and UCB controllers live on it; per lock-step it ships ONE packed record (srlx_agent57_pack_record: action, reward, flags and the
    five item fields -- intrinsic reward, arm, previous action, previous rewards -- of its E lanes) + its frames to the learner rank as a group of two sends, waited
    for between its NEXT policy passes and its environments.  The LEARNER rank owns ring + tree + field arrays for all actor ranks' environments: it posts one group
    of receives per lock-step into staging slot t mod 2 and commits the slab that arrived during the PREVIOUS lock-step INSIDE its update -- ring commit
    (srlx_store_commit_step_packed), item fields (srlx_agent57_unpack_fields), tree add on a side stream behind the update's draw, the priority write-back behind
    them: the tree sees draw, add, write-back in that order and the learner rank's period is max(update, receive).  The reference moves the same information as
    pickled 11-field items on a queue and a pickled list of five state_dicts on a timer (play_mp.py:76-118,289-318; model_torch.py:148-156)."""

    FIELDS = 5  # r_int, arm, prev_action, prev_r_ext, prev_r_int (csrc/srlx_agent57.hip: kA57Fields)
    SLOTS = 2

    def __init__(self, rl_config, n_envs: int, device: int, episode_len: int = 200, sync_interval: int = 16, learner_acts: Optional[bool] = None, seed: int = 0,
                 env=None, parameter=None, always_collective: bool = False, overlap: Optional[bool] = None):
        import copy

        from simple_distributed_rl_amd.device.agent57_fast import Agent57LightFastEngine
        from simple_distributed_rl_amd.device.replay import DeviceReplay

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dev = torch.device(f"cuda:{device}")
        self.sync_interval = int(sync_interval)
        self.is_learner = self.rank == 0
        self.learner_acts = (self.world < 4) if learner_acts is None else bool(learner_acts)
        if self.world == 1:
            self.learner_acts = True
        self.acts = self.learner_acts or not self.is_learner
        self.first_actor_rank = 0 if self.learner_acts else 1
        self.n_actor_ranks = self.world - self.first_actor_rank
        E = self.E = int(n_envs)
        self.cfg = rl_config
        W = rl_config.window_length
        H, W_ = int(rl_config.observation_space.shape[0]), int(rl_config.observation_space.shape[1])
        local_cfg = copy.deepcopy(rl_config)
        local_cfg.memory.capacity, local_cfg.memory.warmup_size = E * 4, 1 << 60  # the local ring only stacks frames; nobody samples it
        self.bus = TransitionBus(E, H * W_, torch.uint8, self.dev, always_collective=always_collective, extra_floats=self.FIELDS,
                                 actor_ranks=range(self.first_actor_rank, self.world))
        self.bus.enable_slots(self.SLOTS)
        # every rank draws its environments, its exploration and its arms from its OWN stream (ranks acting on the same broadcast weights must not produce
        # byte-identical transitions); the learner's replay seed stays `seed`
        kw = dict(episode_len=episode_len, seed=seed + 1_000_003 * self.rank, env=env, parameter=parameter, ring_len=1 + W + 4)
        if self.is_learner:
            total = self.n_actor_ranks * E
            mem = rl_config.memory
            mk = mem.kwargs if mem.name != "ReplayBuffer" else {}
            ring_len = -(-mem.capacity // total) + 1 + W
            self.replay = DeviceReplay(total, ring_len, H * W_, W, 1, int(rl_config.action_space.n), rl_config.batch_size, True, False, float(mk.get("alpha", 0.0)),
                                       float(mk.get("beta_initial", 0.4)), int(mk.get("beta_steps", 1_000_000)), float(mk.get("epsilon", 1e-4)), mem.warmup_size, seed,
                                       device)
            self.replay.enable_deferred_advance()  # every ring commit of the global replay is followed by its tree add (`_ingest_fn`): the add moves the position
            self.local = Agent57LightFastEngine(local_cfg, E, device, role="both" if self.acts else "learner", learner_replay=self.replay, overlap=overlap, **kw)
        else:
            self.local = Agent57LightFastEngine(local_cfg, E, device, role="actor", **kw)
            self.replay = self.local.replay
            self.local.before_env = self.bus.send_end  # the previous slab's frames must have left before the environments overwrite them
        eng = self.local
        self.flat = flatten_parameters(torch.nn.ModuleList(eng.modules()))  # model_torch.py:148-156: the five networks as ONE buffer
        eng.rebind()  # the parameters moved: handles, optimisers and pointer tables read their new addresses
        self.step_count, self._next_ingest, self.env_steps_local = 0, 0, 0
        self.bus.broadcast_params(self.flat)
        eng.on_weights_broadcast()
        # first observations of every environment -> global ring position 0 (a one-off synchronous exchange through slot 0; the records are not used)
        if self.is_learner:
            self.bus.recv_begin(0)
            if self.acts:
                self.bus.put_own(0, eng.record, eng.first_obs)
            self.bus.recv_end()
            self.replay.reset_all(self.bus.slot_obs[0])
        else:
            self.bus.send_begin(eng.record, eng.first_obs)
            self.bus.send_end()
        torch.cuda.synchronize(self.dev)

    @property
    def global_envs(self) -> int:
        return self.n_actor_ranks * self.E

    @property
    def train_count(self) -> int:
        return self.local.train_count

    @property
    def overlap(self) -> bool:
        return self.local.overlap

    def _ingest_fn(self, j: int):
        """The launches that commit slab j: ring (frames, scalars, item masks), item fields, tree add -- out of staging slot j % SLOTS."""
        rp, bus, eng, E = self.replay, self.bus, self.local, self.E
        a = j % self.SLOTS

        def fn():
            rp.commit_packed(bus.slot_scal[a], E, self.FIELDS, bus.slot_obs[a])
            eng.ingest_fields(bus.slot_scal[a], E)  # (reads the ring position the add below moves)
            rp.add_masked()

        return (a,), fn

    def _ingest_ready(self, k: int):
        """Slab to commit during lock-step k, or None: slab j has arrived when lock-step j is over."""
        j = self._next_ingest
        return j if j <= k - 1 else None

    def _act(self, events):
        eng = self.local
        if events is not None:
            events[0].record()
        eng.actor_step()  # (an actor rank: bus.send_end() of the previous slab sits between the policy passes and the environments)
        if events is not None:
            events[1].record()
        self.env_steps_local += self.E
        return eng.pack_record(), eng.env.next_obs

    def step(self, learner_updates: int = 1, events=None):
        """One lock-step of the whole job.  Every rank issues exactly one group of point-to-point transfers per lock-step (and every `sync_interval` lock-steps the
        parameter broadcast behind it), in the same order everywhere."""
        eng, bus, k = self.local, self.bus, self.step_count
        if self.is_learner:
            bus.recv_begin(k % self.SLOTS)  # slab k lands while this lock-step runs
            j = self._ingest_ready(k)
            if j is not None:
                eng.ingest = self._ingest_fn(j)
            if self.acts:
                if eng.overlap:
                    eng.fork_learner(learner_updates)  # the update (and the slab's commit inside it) beside this rank's own actors
                rec, obs = self._act(events)
                bus.put_own(k % self.SLOTS, rec, obs)
                if eng.overlap:
                    eng.join_learner()
                else:
                    eng.run_updates(learner_updates)
                bus.recv_end()
                eng._flip()
            else:
                if events is not None:
                    events[0].record()
                    events[1].record()
                eng.run_updates(learner_updates)
                bus.recv_end()
            if j is not None:  # (after the updates: their warm-up gate saw the replay as the draw did)
                self.replay.note_commit()
                self._next_ingest = j + 1
        else:
            rec, obs = self._act(events)
            bus.send_begin(rec, obs)
        self.step_count += 1
        if self.step_count % self.sync_interval == 0:
            if self.is_learner:
                eng.join_learner()  # broadcast consistent weights: not while Adam is writing them
            bus.broadcast_params(self.flat)
            if not self.is_learner:
                eng.on_weights_broadcast()

    def flush(self):
        """Commit the slabs that have arrived and are still staged (end of a run)."""
        if not self.is_learner:
            self.bus.send_end()
            torch.cuda.synchronize(self.dev)
            return
        self.local.join_learner()
        while self._next_ingest < self.step_count:
            j = self._next_ingest
            self._ingest_fn(j)[1]()
            self.replay.note_commit()
            self._next_ingest = j + 1
        torch.cuda.synchronize(self.dev)

    def capture_graphs(self):
        """The actors' launches stay eager; the learner rank's update is captured per variant (set it publishes into x staging slot it commits) the first time each
        runs -- after at least one eager update (library scratch, event creation)."""
        if self.is_learner and not self.replay.is_warmup_needed():
            self.local.capture_graphs(warm_updates=1 if self.local.train_count == 0 else 0)

    def info(self):
        d = dict(train_count=self.train_count, memory=self.replay.length() if self.is_learner else 0)
        if self.is_learner and self.train_count > 0:
            d.update(self.local.losses())
            d["loss"] = d["ext_loss"]
        return d
