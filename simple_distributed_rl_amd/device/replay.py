"""DeviceReplay -- the HBM-resident replay of the vectorised engine: a lock-step frame/transition ring
(libsrlx `srlx_store_*`) plus the proportional sum-tree (`srlx_per_*`) whose leaf j is the item of
environment j % E at ring time j // E.

Plays the role of srl/rl/memories/priority_replay_buffer.py:170-258 (PriorityReplayBuffer: add /
sample / update with a warm-up gate) for E environments at once, without pickle/zlib and without the
host: every method only enqueues kernels on torch's current stream and works on device tensors, so a
whole actor or learner step can be captured in a HIP graph.
"""
import ctypes
from typing import Optional
from dataclasses import dataclass

import torch

from simple_distributed_rl_amd import _native as N


@dataclass
class ReplayBatch:
    indices: torch.Tensor  # int64 [B] tree indices (update_args)
    weights: torch.Tensor  # float32 [B] IS weights / max
    obs: torch.Tensor  # float32 [B, n+1, W, F]
    actions: torch.Tensor  # int32 [B, n]
    rewards: torch.Tensor  # float32 [B, n]
    terminated: torch.Tensor  # float32 [B, n]


class DeviceReplay:
    def __init__(
        self,
        n_envs: int,
        ring_len: int,
        obs_elems: int,
        window: int,
        n_step: int,
        n_actions: int,
        batch_size: int,
        obs_uint8: bool = True,
        reward_clip: bool = False,
        alpha: float = 0.6,
        beta_initial: float = 0.4,
        beta_steps: int = 1_000_000,
        epsilon: float = 0.0001,
        warmup_size: int = 1000,
        seed: int = 0,
        device: int = 0,
        sample_slack: int = 8,
        has_duplicate: bool = True,
        lagged_add: bool = False,
        fused_draw: bool = True,
    ):
        """lagged_add (the single-GPU round-5 lock-step): the tree add of a lock-step is not launched behind its ring commit but handed to the NEXT update
        (`take_pending_add`), which runs it on a side branch between its draw and its priority write-back -- off the lock-step's serial tail.  Commits then carry
        their ring position as a launch argument (the host's count), the device-resident position is the learner's view and advances with the add; `ring_len` must
        include ONE spare slot (srlx_store_set_item_slack), and the item masks alternate between two buffers (the add of lock-step t reads its mask while the
        commit of t + 1 writes the other)."""
        self.lib = N.lib()
        self.E, self.L, self.F, self.W, self.n, self.A, self.B = n_envs, ring_len, obs_elems, window, n_step, n_actions, batch_size
        self.obs_uint8 = obs_uint8
        self.device_index = int(device)
        self.dev = torch.device(f"cuda:{device}")
        self.seed = int(seed)
        self.warmup_size = int(warmup_size)
        self.slack = int(sample_slack)
        hs = N.c_p()
        N.check(
            self.lib.srlx_store_create(
                ctypes.byref(hs), n_envs, ring_len, obs_elems, N.OBS_U8 if obs_uint8 else N.OBS_F32, window, n_step, n_actions,
                int(bool(reward_clip)), self.seed, self.device_index,
            )
        )
        self.h_store = hs
        self.lagged = bool(lagged_add)
        if self.lagged:
            N.check(self.lib.srlx_store_set_item_slack(hs, 1))
        self.item_len = int(self.lib.srlx_store_item_len(hs))
        self.capacity = int(self.lib.srlx_store_per_capacity(hs))
        hp = N.c_p()
        N.check(
            self.lib.srlx_per_create(ctypes.byref(hp), self.capacity, float(alpha), float(beta_initial), float(beta_steps), int(bool(has_duplicate)), float(epsilon), self.device_index)
        )
        self.h_per = hp
        d = self.dev
        B, n = batch_size, n_step
        # preallocated device buffers (stable addresses: required for HIP-graph capture)
        self.item_masks = [torch.zeros(n_envs, dtype=torch.uint8, device=d) for _ in range(2 if self.lagged else 1)]
        self.item_mask = self.item_masks[0]  # (the mask the LAST commit wrote)
        self._pending_add = None  # lagged: parity of the commit whose tree add has not been launched yet
        self.u = torch.zeros(B + self.slack, dtype=torch.float64, device=d)
        self.rng_counter = torch.zeros(1, dtype=torch.int64, device=d)
        obs = torch.zeros((B, n + 1, window, obs_elems), dtype=torch.float32, device=d)  # (the float32 windows of the autograd yardstick: shared by both sets)
        # TWO sets of everything a draw writes (batch, frame-offset tables, consumed-uniform count): an engine draws the batch of update u + 1 while update u's
        # backward pass still reads its own (RainbowEngine: pre-draw); `use_set(k)` points the plain attribute names at one of them
        self.two_sets = True
        self._sets = []
        for _ in range(2):
            self._sets.append(dict(
                batch=ReplayBatch(indices=torch.zeros(B, dtype=torch.int64, device=d), weights=torch.zeros(B, dtype=torch.float32, device=d), obs=obs,
                                  actions=torch.zeros((B, n), dtype=torch.int32, device=d), rewards=torch.zeros((B, n), dtype=torch.float32, device=d),
                                  terminated=torch.zeros((B, n), dtype=torch.float32, device=d)),
                used=torch.zeros(1, dtype=torch.int64, device=d),
                frame_off_next=torch.zeros((B, n, window), dtype=torch.int64, device=d),
                frame_off_all=torch.zeros((B, n + 1, window), dtype=torch.int64, device=d)))
        self.use_set(0)
        self.stacked = torch.zeros((n_envs, window, obs_elems), dtype=torch.float32, device=d)
        # frame-offset tables for the matrix-core network (conv1 reads the uint8 ring directly)
        self.frame_off_actor = torch.zeros((n_envs, window), dtype=torch.int64, device=d)
        self.obs0 = torch.zeros((B, 1, window, obs_elems), dtype=torch.float32, device=d)
        base, fb = N.c_p(), N.c_i64()
        N.check(self.lib.srlx_store_obs_base(hs, ctypes.byref(base), ctypes.byref(fb)))
        self.obs_base = base.value
        self._steps_committed = 0
        pos, nr, sie = N.c_p(), N.c_p(), N.c_p()
        N.check(self.lib.srlx_store_views(hs, ctypes.byref(pos), ctypes.byref(nr), ctypes.byref(sie)))
        self._views = (pos, nr, sie)
        self.needs_reset_ptr = nr  # uint8 [E]: lanes whose next lock-step only delivers the first frame of a new episode
        self.deferred_advance = False
        self._fused_draw = bool(fused_draw)  # the draw and the item gather as ONE launch (srlx_per_sample_gather_train)
        self.table_fresh = False  # `frame_off_actor` holds the table of the CURRENT ring position (written by the last commit)
        self.has_duplicate = bool(has_duplicate)
        self._drew = False

    def use_set(self, k: int):
        """`batch`, `used`, `frame_off_all`, `frame_off_next` := buffer set k (0 / 1): what the next draw writes and what readers of these names see."""
        st = self._sets[k]
        self.set_index = k
        self.batch, self.used, self.frame_off_next, self.frame_off_all = st["batch"], st["used"], st["frame_off_next"], st["frame_off_all"]

    def enable_deferred_advance(self):
        """The round-4 lock-step: a commit leaves the ring position where it is and the PER add that closes the lock-step advances it inside its own launch
        (srlx_per_set_add_counters), so the ring commit may run while a learner still reads the replay (ring slot p + 1 and the scalars of p belong to no stored
        item; p itself feeds the learner's item lookup) and the add is the only launch behind the join.  Every commit must then be followed by exactly one add."""
        N.check(self.lib.srlx_per_set_add_counters(self.h_per, self._views[0], None))
        self.deferred_advance = True

    def close(self):
        if getattr(self, "h_store", None):
            torch.cuda.synchronize(self.dev)
            self.lib.srlx_store_destroy(self.h_store)
            self.lib.srlx_per_destroy(self.h_per)
            self.h_store = self.h_per = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- actor side ---------------------------------------------------------------------------
    def reset_all(self, first_obs: torch.Tensor):
        N.check(self.lib.srlx_store_reset_all(self.h_store, N.tptr(first_obs), N.torch_stream_ptr()))
        self.table_fresh = False

    def stack_current(self) -> torch.Tensor:
        """float32 [E, W, F] policy input at the current ring position (oldest frame first)."""
        N.check(self.lib.srlx_store_stack_current(self.h_store, N.tptr(self.stacked), N.torch_stream_ptr()))
        return self.stacked

    def commit(self, actions, rewards, terminated, done, next_obs, defer_add: bool = False, next_table: bool = False, bump: Optional[torch.Tensor] = None):
        """One lock-step transition of all E envs (ONE launch: frames, scalars, item mask, ring position) + the PER add of the items that became complete
        (priority=None -> max_priority, proportional_memory.py:121-122; reset positions get 0).
        defer_add: the ring commit only; the caller adds the E leaves later with priorities of its own (`add_raw` / `add_masked`; `item_mask` says which lanes
        completed an item).  next_table: the commit also writes `frame_off_actor` for the NEXT policy pass.  bump: an int64 device counter advanced by the launch."""
        st = N.torch_stream_ptr()
        if self.lagged:
            self.flush_pending_add()  # (a commit whose add nobody took: the adds stay in commit order)
            par = self._steps_committed & 1
            self.item_mask = self.item_masks[par]
            N.check(self.lib.srlx_store_commit_step_at(self.h_store, self._steps_committed, N.tptr(actions), N.tptr(rewards), N.tptr(terminated), N.tptr(done),
                                                       N.tptr(next_obs), N.tptr(self.item_mask), N.tptr(self.frame_off_actor) if next_table else None, N.tptr(bump), st))
            self.table_fresh = bool(next_table)
            self._steps_committed += 1
            self._pending_add = par
            if not defer_add:
                self.flush_pending_add()
            return
        N.check(
            self.lib.srlx_store_commit_step_ex(
                self.h_store, N.tptr(actions), N.tptr(rewards), N.tptr(terminated), N.tptr(done), N.tptr(next_obs), N.tptr(self.item_mask),
                N.tptr(self.frame_off_actor) if next_table else None, 0 if self.deferred_advance else 1, N.tptr(bump), st
            )
        )
        self.table_fresh = bool(next_table)
        if not defer_add:
            self.add_masked()
        self._steps_committed += 1

    def commit_packed(self, records: torch.Tensor, envs_per_record: int, extra_floats: int, next_obs: torch.Tensor, est_records: Optional[torch.Tensor] = None,
                      est_out: Optional[torch.Tensor] = None):
        """The ring commit of a lock-step that arrived as packed records (uint8 [ranks][record bytes], device/dist.py) -- LAUNCH ONLY: no host bookkeeping, so the call
        may sit inside a captured graph; the caller reports every execution with `note_commit()`.  The ring position advances inside the launch -- or, after
        `enable_deferred_advance()`, with the tree add that must follow (the launch then spreads over eight times as many workgroups: no arrival tickets).
        est_records / est_out: srlx_store_commit_step_packed."""
        assert records.dtype == torch.uint8 and records.dim() == 2 and records.is_contiguous()
        N.check(self.lib.srlx_store_commit_step_packed(self.h_store, N.tptr(records), records.shape[1], int(envs_per_record), int(extra_floats), N.tptr(next_obs),
                                                       N.tptr(self.item_mask), N.tptr(est_records), N.tptr(est_out), 0 if self.deferred_advance else 1, N.torch_stream_ptr()))

    def note_commit(self):
        """One `commit_packed` launch has been enqueued for execution (eagerly or by a graph replay)."""
        self._steps_committed += 1
        self.table_fresh = False

    def add_estimates(self, est: torch.Tensor):
        """The PER add of the last committed lock-step with actor-side estimates: float32 [E], >= 0 an |td| estimate, -1 max_priority, -2 no item (SRLX_PRIO_EST_F32)."""
        assert est.dtype == torch.float32 and est.numel() == self.E
        N.check(self.lib.srlx_per_add(self.h_per, self.E, N.tptr(est), N.PRIO_EST_F32, 1, N.torch_stream_ptr()))

    def check_draws(self):
        """Host-side check of the last draw (synchronises): a draw without duplicates that ran out of uniforms left the batch untouched (`used` = -1) -- silently
        training on the previous batch would be the alternative.  Called where the host synchronises anyway (engine.info())."""
        if not self._drew:
            return
        used = int(self.used.item())
        if used < 0:
            raise RuntimeError(f"DeviceReplay: the last draw of {self.B} items (has_duplicate={self.has_duplicate}) ran out of uniforms ({self.u.numel()} supplied): "
                               "fewer distinct non-zero leaves than the batch needs, or raise sample_slack")

    def add_masked(self, mask: Optional[torch.Tensor] = None):
        """The PER add of the last committed lock-step at max_priority (0 where `item_mask` says the lock-step completed no item for the lane)."""
        if self.lagged and mask is None:
            return self.flush_pending_add()
        N.check(self.lib.srlx_per_add(self.h_per, self.E, N.tptr(self.item_mask if mask is None else mask), N.PRIO_NONE_MASKED, 1, N.torch_stream_ptr()))

    def take_pending_add(self):
        """lagged: (key, callable) for the add of the last commit -- the caller runs the callable exactly once (inside an update: RainbowEngine.ingest) --, or None."""
        if self._pending_add is None:
            return None
        par, self._pending_add = self._pending_add, None
        mask = self.item_masks[par]
        return ("add", par), (lambda: self.add_masked(mask))

    def flush_pending_add(self):
        """lagged: launch the pending add now, on the current stream (nothing of the learner in flight)."""
        pend = self.take_pending_add() if self.lagged else None
        if pend is not None:
            pend[1]()

    def add_raw(self, priorities_f64: torch.Tensor):
        """The deferred PER add of the last committed lock-step: E final leaf values (already transformed; 0 = no item), float64 on the device."""
        assert priorities_f64.dtype == torch.float64 and priorities_f64.numel() == self.E
        N.check(self.lib.srlx_per_add(self.h_per, self.E, N.tptr(priorities_f64), N.PRIO_RAW, 1, N.torch_stream_ptr()))

    def max_priority_into(self, out: torch.Tensor):
        N.check(self.lib.srlx_per_max_priority(self.h_per, N.tptr(out), N.torch_stream_ptr()))

    def length(self) -> int:
        """Items in the tree.  Counted from committed lock-steps (E adds each) rather than the library's
        host mirror, so it stays exact when commits are replayed from a HIP graph."""
        # (lagged: the tree -- what a draw sees -- trails the commits by the one pending add; the warm-up gate must not open on leaves that are not there yet)
        pending = 1 if (self.lagged and self._pending_add is not None) else 0
        return min(self.capacity, (self._steps_committed - pending) * self.E)

    def is_warmup_needed(self) -> bool:
        return self.length() < self.warmup_size

    # ---- learner side -------------------------------------------------------------------------
    def _draw(self, d_step: torch.Tensor, uniforms: Optional[torch.Tensor], st):
        """B indices + IS weights into `self.batch`.  Without explicit uniforms the sampler generates them itself from the keyed counter
        generator (srlx_per_sample_keyed: the values srlx_rng_uniform would have written for this seed and counter, one launch less)."""
        b = self.batch
        self._drew = True
        if uniforms is None:
            N.check(self.lib.srlx_per_sample_keyed(self.h_per, self.B, N.tptr(d_step), self.seed ^ 0x5EED, N.tptr(self.rng_counter), self.u.numel(), N.tptr(b.indices),
                                                   None, N.tptr(b.weights), N.tptr(self.used), st))
        else:
            N.check(self.lib.srlx_per_sample(self.h_per, self.B, 0, N.tptr(d_step), N.tptr(uniforms), uniforms.numel(), N.tptr(b.indices), None, N.tptr(b.weights),
                                             N.tptr(self.used), 1, st))

    def sample(self, d_step: torch.Tensor, uniforms: torch.Tensor = None) -> ReplayBatch:
        """PER sample of B items + gather of their n-step windows.  `d_step` is the device int64 train
        count feeding the beta schedule.  Uniforms come from the counter RNG unless given."""
        st = N.torch_stream_ptr()
        b = self.batch
        self._draw(d_step, uniforms, st)
        N.check(
            self.lib.srlx_store_gather_nstep(
                self.h_store, self.B, N.tptr(b.indices), N.tptr(b.obs), N.tptr(b.actions), N.tptr(b.rewards), N.tptr(b.terminated), st
            )
        )
        return b

    def draw_like_random_sample(self, population: int = None) -> torch.Tensor:
        """The uniform ReplayBuffer's draw (srl/rl/memories/priority_memories/replay_buffer.py:33-34: `random.sample(self.memory, batch_size)`)
        with the reference's own generator: `random.sample(range(len), B)` consumes Python's `random` stream exactly like sampling the list does (the
        draws depend on the population SIZE only), so under `random.seed(s)` the device store hands out the items the reference's list would -- item j
        of the list is leaf j of the store (insertion order, then ring order; one environment, or lane-major insertion with E lanes).  Without
        replacement, every weight 1.0.  The indices land in `batch.indices` (tree indices, like a PER draw): follow with `gather_drawn()`.  Host-side
        by nature (the generator is Python's); the engines' own draws stay on the device (`has_duplicate=False`: without replacement there too)."""
        import random

        n = self.length() if population is None else int(population)
        slots = random.sample(range(n), self.B)
        idx = torch.tensor(slots, dtype=torch.int64) + (self.capacity - 1)
        self.batch.indices.copy_(idx.to(self.dev, non_blocking=False))
        self.batch.weights.fill_(1.0)
        return self.batch.indices

    def gather_drawn(self, all_states: bool = True) -> ReplayBatch:
        """n-step scalars and frame-offset tables (or float32 windows) of the items whose tree indices sit in `batch.indices`."""
        st = N.torch_stream_ptr()
        b = self.batch
        if all_states:
            N.check(self.lib.srlx_store_gather_train(self.h_store, self.B, N.tptr(b.indices), N.tptr(self.frame_off_all), N.tptr(self.frame_off_next), N.tptr(b.actions),
                                                     N.tptr(b.rewards), N.tptr(b.terminated), st))
        else:
            N.check(self.lib.srlx_store_gather_nstep(self.h_store, self.B, N.tptr(b.indices), N.tptr(b.obs), N.tptr(b.actions), N.tptr(b.rewards), N.tptr(b.terminated), st))
        return b

    def frame_table_current(self) -> torch.Tensor:
        """int64 [E, W] byte offsets of the frames that form every env's current stacked observation (no launch when the last commit wrote it)."""
        if not self.table_fresh:
            self.flush_pending_add()  # (the kernel reads the device position: it must have caught up with the commits)
            N.check(self.lib.srlx_store_frame_table_current(self.h_store, N.tptr(self.frame_off_actor), N.torch_stream_ptr()))
        return self.frame_off_actor

    def sample_items(self, d_step: torch.Tensor, uniforms: torch.Tensor = None, all_states: bool = False) -> ReplayBatch:
        """PER sample, then: frame-offset table of s_1..s_n (`frame_off_next`, for srlx_qnet_forward_u8),
        float32 pixels of s_0 only (`obs0`, the one state autograd needs), and the n-step scalars.
        all_states=True (hand-written training pass): the table of s_0..s_n (`frame_off_all`) instead of the pixels."""
        st = N.torch_stream_ptr()
        b = self.batch
        self._drew = True
        if all_states and uniforms is None and self.B <= 64 and self._fused_draw:  # the draw and the gather as ONE launch (srlx_per_sample_gather_train)
            N.check(self.lib.srlx_per_sample_gather_train(self.h_per, self.h_store, self.B, N.tptr(d_step), self.seed ^ 0x5EED, N.tptr(self.rng_counter), self.u.numel(),
                                                          N.tptr(b.indices), N.tptr(b.weights), N.tptr(self.used), N.tptr(self.frame_off_all), N.tptr(self.frame_off_next),
                                                          N.tptr(b.actions), N.tptr(b.rewards), N.tptr(b.terminated), st))
            return b
        self._draw(d_step, uniforms, st)
        if all_states:  # item location, n-step scalars and both offset tables (s_0..s_n, s_1..s_n) in one launch
            N.check(
                self.lib.srlx_store_gather_train(
                    self.h_store, self.B, N.tptr(b.indices), N.tptr(self.frame_off_all), N.tptr(self.frame_off_next), N.tptr(b.actions), N.tptr(b.rewards),
                    N.tptr(b.terminated), st
                )
            )
        else:
            N.check(
                self.lib.srlx_store_gather_items(
                    self.h_store, self.B, N.tptr(b.indices), 1, self.n, N.tptr(self.frame_off_next), N.tptr(b.actions), N.tptr(b.rewards), N.tptr(b.terminated), st
                )
            )
            N.check(self.lib.srlx_store_gather_obs(self.h_store, self.B, 0, 1, N.tptr(self.obs0), st))
        return b

    def count_updates_in(self, counter: torch.Tensor):
        """Every later `update()` also adds 1 to `counter` (int64 device scalar): srlx_per_set_update_counter."""
        self._update_counter = counter  # keeps it alive
        N.check(self.lib.srlx_per_set_update_counter(self.h_per, N.tptr(counter)))

    def update(self, indices: torch.Tensor, priorities: torch.Tensor):
        """float32 |td| priorities, transformed on the device like numpy would (proportional_memory.py:172)."""
        N.check(self.lib.srlx_per_update(self.h_per, indices.numel(), N.tptr(indices), N.tptr(priorities), N.PRIO_F32, 1, N.torch_stream_ptr()))

    # ---- introspection ------------------------------------------------------------------------
    def per_state(self):
        mp, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
        N.check(self.lib.srlx_per_backup(self.h_per, ctypes.byref(mp), ctypes.byref(size), ctypes.byref(write), None))
        return dict(max_priority=mp.value, size=size.value, write=write.value)

    def hbm_bytes(self) -> int:
        frame = self.F * (1 if self.obs_uint8 else 4)
        return self.E * self.L * (frame + 13) + 16 * self.capacity
