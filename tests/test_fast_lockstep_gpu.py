"""GPU tests of the round-4 lock-step (RainbowEngine(fast=True)): every fused launch against the launches it replaces, bit for bit, and the whole
engine against the fifteen-launch engine (same seed: same actions, same ring, same tree, same weights).

Reference behaviour under test (file:line under the reference root): the batched Worker.policy step, srl/algorithms/rainbow/rainbow.py:301-329 behind
srl/base/rl/worker_run.py:316-322; Worker.on_step / add_tracking, rainbow.py:331-352; the parameter hand-over from trainer to actors,
srl/base/run/play_mp.py:289-303,151-165."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _store(E=37, L=12, F=7056, W=4, n=3, A=6, seed=5):
    from simple_distributed_rl_amd.device.replay import DeviceReplay

    return DeviceReplay(E, L, F, W, n, A, 8, True, True, 0.5, 0.4, 1e6, 1e-4, 4, seed, 0)


def _drive(r, steps, fused, episode_len=5, table=False):
    """`steps` synthetic lock-steps committed with the one-launch commit (fused) or through srlx_store_commit_step; returns what the ring then serves."""
    from simple_distributed_rl_amd import _native as N

    E, F = r.E, r.F
    d = r.dev
    g = torch.Generator(device=d).manual_seed(11)
    first = torch.randint(0, 256, (E, F), dtype=torch.uint8, device=d, generator=g)
    r.reset_all(first)
    next_obs = torch.zeros((E, F), dtype=torch.uint8, device=d)
    rew = torch.zeros(E, dtype=torch.float32, device=d)
    term = torch.zeros(E, dtype=torch.uint8, device=d)
    done = torch.zeros(E, dtype=torch.uint8, device=d)
    tables, masks = [], []
    for k in range(steps):
        N.check(r.lib.srlx_synth_env_step(r.h_store, episode_len, N.tptr(next_obs), N.tptr(rew), N.tptr(term), N.tptr(done), N.torch_stream_ptr()))
        act = torch.randint(0, r.A, (E,), dtype=torch.int32, device=d, generator=g)
        if fused:
            r.commit(act, rew, term, done, next_obs, next_table=table)
            if table:
                assert r.table_fresh
                tables.append(r.frame_off_actor.clone())
        else:
            N.check(r.lib.srlx_store_commit_step(r.h_store, N.tptr(act), N.tptr(rew), N.tptr(term), N.tptr(done), N.tptr(next_obs), N.tptr(r.item_mask), N.torch_stream_ptr()))
            N.check(r.lib.srlx_per_add(r.h_per, E, N.tptr(r.item_mask), N.PRIO_NONE_MASKED, 1, N.torch_stream_ptr()))
            r._steps_committed += 1
            if table:
                r.table_fresh = False
                tables.append(r.frame_table_current().clone())
        masks.append(r.item_mask.clone())
    stacked = r.stack_current().clone()
    idx = torch.arange(r.capacity - 1, r.capacity - 1 + min(r.capacity, 8), dtype=torch.int64, device=d)
    r.batch.indices.copy_(idx[: r.B] if idx.numel() >= r.B else idx.repeat(r.B)[: r.B])
    b = r.gather_drawn(all_states=False)
    torch.cuda.synchronize()
    return dict(stacked=stacked, obs=b.obs.clone(), actions=b.actions.clone(), rewards=b.rewards.clone(), terminated=b.terminated.clone(), tables=tables, masks=masks,
                per=r.per_state())


def test_one_launch_commit_with_deferred_advance_equals_commit_then_add():
    """commit_ex(advance = 0) + the add that moves the ring position (srlx_per_set_add_counters) against commit + add: the same ring, masks, next-pass frame
    tables (vs srlx_store_frame_table_current after the advance) and tree bookkeeping over episode boundaries and ring wrap-around."""
    a, b = _store(), _store()
    b.enable_deferred_advance()
    ra = _drive(a, 29, fused=False, table=True)
    rb = _drive(b, 29, fused=True, table=True)
    for k in ("stacked", "obs", "actions", "rewards", "terminated"):
        assert torch.equal(ra[k], rb[k]), k
    for i, (x, y) in enumerate(zip(ra["masks"], rb["masks"])):
        assert torch.equal(x, y), i
    for i, (x, y) in enumerate(zip(ra["tables"], rb["tables"])):
        assert torch.equal(x, y), i
    assert ra["per"] == rb["per"]
    # (srlx_store_commit_step / srlx_synth_env_step themselves -- now one launch each -- are held to the model of the reference's tracking ring and to the items
    # the reference's worker emitted by tests/test_hot_path_gpu.py)


def _qnet_pair(E=512, seed=7):
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference

    torch.manual_seed(seed)
    net = EngineQNet(6, (84, 84), 4, 512, 32, "average").cuda()
    with torch.no_grad():  # Q rows with clear and with tied maxima
        net.a2.bias.add_(torch.tensor([0.0, 0.3, 0.0, 0.3, -0.1, 0.2], device="cuda"))
    return net, QNetInference(net, E, 0)


def _frames(E, seed=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    ring = torch.randint(0, 256, (E * 4, 7056), dtype=torch.uint8, device="cuda", generator=g)
    off = (torch.arange(E * 4, dtype=torch.int64, device="cuda") * 7056).view(E, 4).clone()
    off[1, :3] = -1  # an episode start: zero history
    return ring, off


def test_policy_in_the_head_kernel_equals_rng_plus_epsilon_greedy():
    """srlx_qnet_forward_u8_policy against srlx_qnet_forward_u8 + srlx_rng_uniform + srlx_policy_epsilon_greedy: the same Q rows and the same actions for several
    counter values, with per-row epsilons (0, 0.1, 1), with and without an invalid-action mask; the counter is left alone."""
    from simple_distributed_rl_amd import _native as N

    E, A = 512, 6
    net, inf = _qnet_pair(E)
    ring, off = _frames(E)
    lib = N.lib()
    eps = torch.full((E,), 0.1, device="cuda")
    eps[::3] = 1.0
    eps[1::7] = 0.0
    g = torch.Generator(device="cuda").manual_seed(1)
    invalid = (torch.rand((E, A), device="cuda", generator=g) < 0.3).to(torch.uint8)
    invalid[:, 2] = 0  # every row keeps a valid action
    seed = 0xAC7 ^ 5
    for inv in (None, invalid):
        for c0 in (0, 1, 12345):
            counter = torch.tensor([c0], dtype=torch.int64, device="cuda")
            act_f = torch.full((E,), -1, dtype=torch.int32, device="cuda")
            qc = torch.zeros((E, A), device="cuda")
            q_f = inf.forward_u8_policy(ring.data_ptr(), off, eps, seed, counter, act_f, invalid=inv, q_copy=qc).clone()
            assert int(counter.item()) == c0
            q = inf.forward_u8(ring.data_ptr(), off).clone()
            u = torch.zeros(2 * E, dtype=torch.float64, device="cuda")
            N.check(lib.srlx_rng_uniform(seed, N.tptr(counter), 2 * E, N.tptr(u), N.torch_stream_ptr()))
            act = torch.full((E,), -1, dtype=torch.int32, device="cuda")
            N.check(lib.srlx_policy_epsilon_greedy(E, A, N.tptr(q), N.tptr(eps), N.tptr(u), N.tptr(inv), N.tptr(act), N.torch_stream_ptr()))
            torch.cuda.synchronize()
            assert int(counter.item()) == c0 + 1
            assert torch.equal(q_f, q) and torch.equal(qc, q)
            assert torch.equal(act_f, act), (inv is not None, c0)
            assert 0 < int((act != q.argmax(1).to(torch.int32)).sum())  # exploration happened somewhere


def test_published_set_serves_the_same_q_values_and_follows_updates():
    """An actor handle reading a published set (packed filters + operand planes + small vectors) returns the Q rows of the network it was published from, bit for
    bit, also through the other set after the weights moved; the packed filters a publish leaves behind serve the source handle's next forward."""
    from simple_distributed_rl_amd.device.qnet import QNetInference

    E = 512
    net, src = _qnet_pair(E)
    ring, off = _frames(E)
    actor = QNetInference(net, E, 0)
    actor.enable_fc1_planes(private_weights=True)
    actor.enable_actor_sets()
    want0 = src.forward_u8(ring.data_ptr(), off).clone()
    src.publish_to(actor, 0, with_fc1=True)
    actor.select_set(0)
    got0 = actor.forward_u8(ring.data_ptr(), off).clone()
    assert torch.equal(got0, want0)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))
    got_stale = actor.forward_u8(ring.data_ptr(), off).clone()
    assert torch.equal(got_stale, want0), "a selected set must not follow the bound tensors"
    src.weights_changed()
    want1 = src.forward_u8(ring.data_ptr(), off).clone()
    assert not torch.equal(want1, want0)
    src.publish_to(actor, 1, with_fc1=True)
    assert torch.equal(src.forward_u8(ring.data_ptr(), off), want1)  # (the source's own next forward: packed filters from the publish)
    actor.select_set(1)
    assert torch.equal(actor.forward_u8(ring.data_ptr(), off), want1)
    actor.select_set(0)
    assert torch.equal(actor.forward_u8(ring.data_ptr(), off), want0)
    actor.select_set(-1)
    actor.weights_changed()
    assert torch.equal(actor.forward_u8(ring.data_ptr(), off), want1)


def test_half_cu_first_dense_layer_equals_the_other_forms():
    """k_fc1_planes_h (256-thread workgroups, half-slab stages) against k_fc1_planes and the staging-split k_gemm_s16: bit-identical Q rows at equal K splits (4);
    at 8 splits the partial sums associate differently: float32 round-off of the first dense layer only."""
    from simple_distributed_rl_amd.device.qnet import QNetInference

    E = 1024
    net, ref = _qnet_pair(E)
    ring, off = _frames(E)
    want = ref.forward_u8(ring.data_ptr(), off).clone()  # k_gemm_s16, 4 splits at 1024 rows
    pl = QNetInference(net, E, 0)
    pl.enable_fc1_planes(private_weights=True)
    pl.set_fc1_neighbour(8)  # (the largest split count first: it sizes the partial-sum buffer, which may not move once a forward has used it)
    got8 = pl.forward_u8(ring.data_ptr(), off).clone()
    torch.testing.assert_close(got8, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    assert float((got8 - want).abs().max()) < 1e-5 * float(want.abs().max())
    pl.set_fc1_neighbour(4)
    assert torch.equal(pl.forward_u8(ring.data_ptr(), off), want)
    pl.set_fc1_neighbour(0)
    assert torch.equal(pl.forward_u8(ring.data_ptr(), off), want)
    with pytest.raises(Exception):
        pl.set_fc1_neighbour(16)  # would have to reallocate a buffer a forward (or a captured graph) already points at


def _engines(E=512, actor_stream=None, **cfgkw):
    import dataclasses

    from simple_distributed_rl_amd.device.rainbow import EngineSchedule, RainbowDeviceConfig, RainbowEngine

    kw = dict(n_envs=E, batch_size=32, memory_capacity=E * 12, memory_warmup_size=E * 4, target_model_update_interval=5, lr=1e-4, seed=3)
    kw.update(cfgkw)
    cfg = RainbowDeviceConfig(**kw)
    # fc1_neighbour = 4, dgrad_split = 1: the K splits of the fifteen-launch engine's first dense layer and of its conv3 data-gradient GEMM (split-K partial sums associate
    # alike, Q-values and gradients bit-equal); lagged_add off: the
    # tree add behind the join, as the fifteen-launch lock-step orders it (round 5's default runs it inside the NEXT update: the update then samples the tree one add
    # older -- pinned against the oracle in test_lagged_add_tree_order_against_the_oracle)
    fast = RainbowEngine(dataclasses.replace(cfg, schedule=EngineSchedule(fc1_neighbour=4, lagged_add=False, dgrad_split=1)), 0, episode_len=7, overlap=True, fast=True, actor_stream=actor_stream)
    slow = RainbowEngine(cfg, 0, episode_len=7, overlap=True, fast=False)
    assert fast.fast and not slow.fast
    slow.q_online.load_state_dict(fast.q_online.state_dict())
    slow.q_target.load_state_dict(fast.q_target.state_dict())
    slow.q_actor.load_state_dict(fast.q_online.state_dict())
    return fast, slow


def _same_state(fast, slow, tag):
    torch.cuda.synchronize()
    assert torch.equal(fast.actions, slow.actions), tag
    assert fast.train_count == slow.train_count and int(fast.train_count_dev.item()) == int(slow.train_count_dev.item()), tag
    assert fast.replay.per_state() == slow.replay.per_state(), tag
    assert torch.equal(fast.replay.batch.indices, slow.replay.batch.indices), tag
    assert torch.equal(fast.priorities, slow.priorities) and torch.equal(fast.loss, slow.loss), tag
    for (name, p), q in zip(fast.q_online.named_parameters(), slow.q_online.parameters()):
        assert torch.equal(p, q), (tag, name)
    # the optimiser state too: the fast engine takes every tensor's Adam step inside the launch that finishes its gradient (srlx_qnet_fuse_adam_fc1 / _rest:
    # no optimiser launch at all), the fifteen-launch engine all but the first dense layer's in srlx_adam_step
    if hasattr(fast.optimizer, "exp_avg") and hasattr(slow.optimizer, "exp_avg"):
        assert fast.optimizer._rest and not slow.optimizer._rest
        for k, (m1, m2, v1, v2) in enumerate(zip(fast.optimizer.exp_avg, slow.optimizer.exp_avg, fast.optimizer.exp_avg_sq, slow.optimizer.exp_avg_sq)):
            assert torch.equal(m1, m2) and torch.equal(v1, v2), (tag, "adam state", k)
    assert torch.equal(fast.replay.stack_current(), slow.replay.stack_current()), tag


def _tree(eng):
    from simple_distributed_rl_amd import _native as N

    r = eng.replay
    tree = np.empty(2 * r.capacity - 1)
    N.check(r.lib.srlx_per_backup(r.h_per, ctypes.byref(N.c_f64(0)), ctypes.byref(N.c_i64(0)), ctypes.byref(N.c_i64(0)), N.np_ptr(tree)))
    return tree


@pytest.fixture
def _restore_stream():
    yield
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())


@pytest.mark.parametrize("actor_stream", [None, "low"])
def test_fast_lockstep_equals_the_fifteen_launch_lockstep(actor_stream, _restore_stream):
    """Two overlapping engines on one seed, one with the round-4 lock-step: identical actions, sampled indices, losses, priorities, weights, ring and tree at
    every lock-step -- through the warm-up gate, target syncs, episode ends, eager steps, graph capture and replays.  actor_stream="low": the actors' side on a
    low-priority stream of its own (it becomes the thread's current stream) and the update three branches wide (the first dense layer's Adam-fused weight gradient
    on a branch of its own): the same bits."""
    fast, slow = _engines(actor_stream=actor_stream)
    assert (fast.actor_stream is not None) == (actor_stream is not None)
    for eng in (fast, slow):
        for _ in range(6):
            eng._random_rest()
    _same_state(fast, slow, "prefill")
    for k in range(10):
        for eng in (fast, slow):
            eng.step(learner_updates=1)
        _same_state(fast, slow, ("eager", k))
    assert fast.train_count >= 5 and fast.sync_count >= 2
    for eng in (fast, slow):
        eng.capture_graphs()
    _same_state(fast, slow, "captured")
    for k in range(12):
        for eng in (fast, slow):
            eng.step(learner_updates=1)
        _same_state(fast, slow, ("graphs", k))
    np.testing.assert_array_equal(_tree(fast), _tree(slow))
    # two updates per lock-step, then none: the publishing update is the last one; without an update the actors keep their set
    for k, u in enumerate((2, 0, 1, 2)):
        for eng in (fast, slow):
            eng.step(learner_updates=u)
        _same_state(fast, slow, ("updates", k, u))
    np.testing.assert_array_equal(_tree(fast), _tree(slow))
    assert fast.total_env_steps == slow.total_env_steps


def test_fast_lockstep_follows_a_loaded_state_dict():
    """Weights loaded behind the engine's back (restore, a checkpoint) reach the actors' published set and the target handle's cached filters."""
    fast, slow = _engines()
    for eng in (fast, slow):
        for _ in range(6):
            eng._random_rest()
        for _ in range(3):
            eng.step(learner_updates=1)
    torch.cuda.synchronize()
    sd = {k: v + 0.01 * torch.randn_like(v) for k, v in fast.q_online.state_dict().items()}
    for eng in (fast, slow):
        eng.join_learner()
        torch.cuda.synchronize()
        eng.q_online.load_state_dict(sd)
        eng.q_target.load_state_dict(sd)
        if eng.q_actor is not eng.q_online:
            eng.q_actor.load_state_dict(sd)
            eng.inf_actor.weights_changed()
    for k in range(4):
        for eng in (fast, slow):
            eng.step(learner_updates=1)
        _same_state(fast, slow, ("after load", k))


def test_draw_and_gather_in_one_launch_equals_the_two_calls():
    """srlx_per_sample_gather_train against srlx_per_sample_keyed + srlx_store_gather_train: the same indices, weights, n-step scalars (terminal padding included)
    and frame-offset tables, and the generator's counter advances alike -- on a ring with episode ends, zero-priority leaves and wrap-around."""
    import copy

    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    cfg = RainbowDeviceConfig(n_envs=8, batch_size=32, memory_capacity=8 * 40, memory_warmup_size=16, seed=9)
    eng = RainbowEngine(cfg, 0, episode_len=6)
    for _ in range(70):
        eng.step(learner_updates=0)
    r = eng.replay
    step = torch.tensor([123], dtype=torch.int64, device="cuda")
    outs = []
    for fused in (True, False):
        r._fused_draw = fused
        r.rng_counter.fill_(5)
        r.batch.indices.fill_(-1)
        r.frame_off_all.fill_(-7)
        r.frame_off_next.fill_(-7)
        b = r.sample_items(step, all_states=True)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (b.indices, b.weights, b.actions, b.rewards, b.terminated, r.frame_off_all, r.frame_off_next, r.used, r.rng_counter)])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert int(outs[0][7].item()) >= 32 and int(outs[0][8].item()) == 6 and float(outs[0][4].sum()) > 0 and int((outs[0][5] < 0).sum()) > 0


@pytest.mark.parametrize("rows", [128, 96])
def test_learner_passes_on_operand_planes_equal_the_staging_split_gemm(rows):
    """A learner's 128- / 96-row pass with srlx_qnet_set_planes_small (the convolution kernel writes float32 act3 AND operand planes, the first dense layer runs on
    the half-CU planes kernel, rows padded to its tile) returns the Q rows of the staging-split GEMM bit for bit -- with borrowed weight planes (an actor set's)
    and with the handle's own -- and leaves the float32 activations the backward pass reads untouched (same gradients)."""
    from simple_distributed_rl_amd.device.qnet import QNetInference

    net, ref = _qnet_pair(rows)
    ring, off = _frames(rows)
    want = ref.forward_u8(ring.data_ptr(), off).clone()
    actor = QNetInference(net, 512, 0)
    actor.enable_fc1_planes(private_weights=True)
    actor.enable_actor_sets()
    ref.publish_to(actor, 1, with_fc1=True)
    pl = QNetInference(net, rows, 0)
    pl.enable_fc1_planes(private_weights=False)
    pl.set_planes_small(True, actor.set_planes_ptr(1))
    assert torch.equal(pl.forward_u8(ring.data_ptr(), off), want)
    own = QNetInference(net, rows, 0)
    own.enable_fc1_planes(private_weights=True)
    own.set_planes_small(True, None)
    own.refresh_own_planes()
    assert torch.equal(own.forward_u8(ring.data_ptr(), off), want)
    pl.set_planes_small(False, None)
    assert torch.equal(pl.forward_u8(ring.data_ptr(), off), want)
    if rows == 128:  # the training pass: gradients from the float32 activations the planes pass also wrote
        B = 32
        g = torch.Generator(device="cuda").manual_seed(4)
        gq = torch.randn((B, 6), device="cuda", generator=g)
        grads = []
        for use_planes in (False, True):
            h = QNetInference(net, rows, 0)
            h.enable_training(B)
            if use_planes:
                h.enable_fc1_planes(private_weights=False)
                h.set_planes_small(True, actor.set_planes_ptr(1))
            h.forward_u8(ring.data_ptr(), off)
            h.backward_u8(ring.data_ptr(), off, gq, sample_stride=4)
            torch.cuda.synchronize()
            grads.append([p.grad.clone() for p in net.kernel_parameters()])
        for a, b in zip(*grads):
            assert torch.equal(a, b)


@pytest.mark.parametrize("actor_stream", [None, "low"])
def test_lagged_add_tree_order_against_the_oracle(actor_stream):
    """The shipped single-GPU lock-step (round 5): the tree add of lock-step t is not launched behind the join but rides on a side branch of update t + 1, between
    that update's draw and its priority write-back (the ring commit carries its position as a launch argument, the device position is the learner's view and moves
    with the add, the ring has one spare slot).  Replayed on the CPU oracle in the order the tree must have seen -- draw (keyed uniforms), E adds at max_priority / 0,
    write-back with the priorities the update produced -- every sampled index and the final tree are bit-equal, eagerly and from the lazily captured graphs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import hot_path_oracle as H
    from oracle_bindings import ADD_RAW, OraclePER
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    cfg = RainbowDeviceConfig(n_envs=512, batch_size=32, memory_capacity=512 * 9, memory_warmup_size=512 * 4, seed=7, target_model_update_interval=5)
    eng = RainbowEngine(cfg, 0, episode_len=7, overlap=True, fast=True, actor_stream=actor_stream)
    try:
        rp = eng.replay
        assert eng.fast and rp.lagged and rp.capacity == 512 * 9
        o = OraclePER(rp.capacity, cfg.memory_alpha, cfg.memory_beta_initial, cfg.memory_beta_steps, True, cfg.memory_epsilon)
        B, checked = cfg.batch_size, 0

        def oracle_add(mask):
            for m in mask.cpu().numpy():
                o.add(None) if m else o.add(0.0, mode=ADD_RAW)

        for T in range(30):
            if T == 16:
                eng.enable_lazy_capture()  # (capture_graphs() would also take a warm-up actor step: one more commit + add than this replay counts)
            counter0, trained0 = int(rp.rng_counter.item()), eng.train_count
            eng.step(learner_updates=1)
            torch.cuda.synchronize()
            trained = eng.train_count > trained0
            if trained:
                used, idx, w, _ = o.sample(B, trained0, H.rng_uniform(cfg.seed ^ 0x5EED, trained0, rp.u.numel()))
                assert used == int(rp.used.item()) and used > 0
                np.testing.assert_array_equal(rp.batch.indices.cpu().numpy(), idx)
                checked += 1
            if T >= 1:  # the add of lock-step T - 1 ran inside this lock-step's update (or alone, below the warm-up); its mask buffer is the one commit T left alone
                oracle_add(rp.item_masks[(T - 1) & 1])
            if trained:
                o.update(idx, eng.priorities.cpu().numpy())
        info = eng.info()  # (launches the last commit's add)
        torch.cuda.synchronize()
        oracle_add(rp.item_masks[(rp._steps_committed - 1) & 1])
        st = rp.per_state()
        import ctypes

        from simple_distributed_rl_amd import _native as N
        tree = np.empty(2 * rp.capacity - 1)
        mp_, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
        N.check(rp.lib.srlx_per_backup(rp.h_per, ctypes.byref(mp_), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
        assert checked >= 20 and info["train_count"] == eng.train_count and len(eng._learner_graphs) >= 2
        assert (tree == o.tree()).all() and mp_.value == o.max_priority and write.value == o.write and st["size"] == min(rp.capacity, rp._steps_committed * 512)
    finally:
        eng.close()


def test_full_size_engine_tree_order_against_the_oracle():
    """BASELINE.json configs[2] at its full size -- E = 1024 environments, 1 000 448 PER leaves (977 ring slots), batch 32, n = 3 -- on the shipped lock-step
    (`bench.py`'s engine: fast, lagged add, actors on a low-priority stream, lazily captured update graphs): after `prefill()` the C oracle takes over the device
    tree (its fill is what tests/test_per_gpu.py checks at this size), then 20 lock-steps are replayed on it in the order the tree must have seen -- draw (keyed
    uniforms), 1024 adds at max_priority / 0, write-back of the priorities the update produced: every drawn index, the uniform consumption and the final tree
    (2 000 895 nodes), max_priority and write position are bit-equal."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes

    import numpy as np

    import hot_path_oracle as H
    from oracle_bindings import ADD_RAW, OraclePER
    from simple_distributed_rl_amd import _native as N
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine

    cfg = RainbowDeviceConfig()
    assert cfg.n_envs == 1024 and cfg.memory_capacity == 1_000_000 and cfg.batch_size == 32 and cfg.multisteps == 3
    eng = RainbowEngine(cfg, 0, episode_len=200, overlap=True, fast=True, actor_stream="low")
    try:
        rp = eng.replay
        assert eng.fast and rp.lagged and rp.capacity == 1024 * 977
        eng.prefill()
        eng.refresh_host_mirrors()
        torch.cuda.synchronize()

        def backup():
            tree = np.empty(2 * rp.capacity - 1)
            mp_, size, write = N.c_f64(0), N.c_i64(0), N.c_i64(0)
            N.check(rp.lib.srlx_per_backup(rp.h_per, ctypes.byref(mp_), ctypes.byref(size), ctypes.byref(write), N.np_ptr(tree)))
            return mp_.value, size.value, write.value, tree

        mp0, size0, write0, tree0 = backup()
        assert size0 == rp.capacity
        o = OraclePER(rp.capacity, cfg.memory_alpha, cfg.memory_beta_initial, cfg.memory_beta_steps, True, cfg.memory_epsilon)
        o.set_state(mp0, size0, write0, tree0)
        B, checked = cfg.batch_size, 0

        def oracle_add(mask):
            for m in mask.cpu().numpy():
                o.add(None) if m else o.add(0.0, mode=ADD_RAW)

        for T in range(20):
            if T == 6:
                eng.enable_lazy_capture()
            trained0 = eng.train_count
            assert int(rp.rng_counter.item()) == trained0
            eng.step(learner_updates=1)
            torch.cuda.synchronize()
            assert eng.train_count == trained0 + 1
            used, idx, w, _ = o.sample(B, trained0, H.rng_uniform(cfg.seed ^ 0x5EED, trained0, rp.u.numel()))
            assert used == int(rp.used.item()) and used > 0
            np.testing.assert_array_equal(rp.batch.indices.cpu().numpy(), idx)
            np.testing.assert_allclose(rp.batch.weights.cpu().numpy(), np.asarray(w, np.float32), rtol=1e-6)
            checked += 1
            if T >= 1:  # the add of lock-step T - 1 ran inside this lock-step's update
                oracle_add(rp.item_masks[(T - 1) & 1])
            o.update(idx, eng.priorities.cpu().numpy())
        eng.info()  # (launches the last commit's add)
        torch.cuda.synchronize()
        oracle_add(rp.item_masks[(rp._steps_committed - 1) & 1])
        mp1, size1, write1, tree1 = backup()
        assert checked == 20 and len(eng._learner_graphs) >= 2
        assert (tree1 == o.tree()).all() and mp1 == o.max_priority and write1 == o.write and size1 == rp.capacity
    finally:
        eng.close()


def test_synthetic_environments_do_not_depend_on_the_lagged_add():
    """The synthetic environments key frames, rewards and episode ends by the ring position.  With the lagged add the device-resident position is the learner's view
    (it moves with the tree add, on the learner's stream), so the environments take the ACTORS' position as a launch argument (srlx_synth_env_step_at): rewards, episode
    ends and frames of 40 lock-steps with updates equal those of the engine whose add follows the join (EngineSchedule(lagged_add=False)), whatever the streams do."""
    from simple_distributed_rl_amd.device.rainbow import EngineSchedule, RainbowDeviceConfig, RainbowEngine

    def run(lagged):
        cfg = RainbowDeviceConfig(n_envs=512, batch_size=32, memory_capacity=512 * 9, memory_warmup_size=512 * 4, seed=7, target_model_update_interval=5,
                                  schedule=EngineSchedule(lagged_add=lagged))
        eng = RainbowEngine(cfg, 0, episode_len=7, overlap=True, fast=True, actor_stream="low")
        try:
            assert eng.replay.lagged == lagged
            out = []
            for T in range(40):
                if T == 12:
                    eng.enable_lazy_capture()
                eng.step(learner_updates=1)
                e = eng.env
                out.append((e.rewards.clone(), e.done.clone(), e.terminated.clone(), e.next_obs[:, ::97].clone()))
            torch.cuda.synchronize()
            return out
        finally:
            eng.close()

    a, b = run(True), run(False)
    for T, (x, y) in enumerate(zip(a, b)):
        for k, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), (T, k)
    assert any(int(x[1].sum()) > 0 for x in a)  # episodes ended
