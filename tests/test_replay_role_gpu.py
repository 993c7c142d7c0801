"""f2: the reference's three-role topology (srl/base/run/play_mp_memory.py:253-351 -- actors -> memory process -> trainer, prefetch queue of 5) on the device
path: a REPLAY GPU between the actor GPUs and the learner GPU (device/replay_role.py).  Three ranks share the test GPU over gloo (the transport is then
host-staged; everything else is the real code path)."""
import os
import socket
import sys

import numpy as np
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


def test_a_served_batch_is_the_ring_batch():
    """srlx_pack_frames: the frames a sampled batch's offset table points at, packed into one message, and the table re-based onto it: the network gives
    the SAME bits on (packed frames, re-based table) as on (ring, table) -- zero-history frames (-1) and terminal padding included."""
    from simple_distributed_rl_amd import _native as N
    from simple_distributed_rl_amd.device.qnet import EngineQNet, QNetInference
    from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig, RainbowEngine
    from simple_distributed_rl_amd.device.replay_role import BatchCodec

    cfg = RainbowDeviceConfig(n_envs=8, batch_size=16, memory_capacity=8 * 64, memory_warmup_size=32)
    eng = RainbowEngine(cfg, 0, episode_len=7)  # short episodes: many items with zero history and terminal padding
    for _ in range(70):
        eng.step(learner_updates=0)
    rp = eng.replay
    b = rp.sample_items(eng.train_count_dev, all_states=True)
    c = BatchCodec(cfg.batch_size, cfg.multisteps, cfg.window_length, 84 * 84)
    msg = torch.zeros(c.nbytes, dtype=torch.uint8, device="cuda")
    rel = c.view(msg, "rel_all", torch.int64)
    N.check(rp.lib.srlx_pack_frames(N.c_p(rp.obs_base), N.tptr(rp.frame_off_all), c.rows, c.F, N.c_p(msg.data_ptr() + c.off["frames"][0]), N.tptr(rel), None))
    torch.cuda.synchronize()
    off = rp.frame_off_all.view(-1)
    assert int((off < 0).sum()) > 0 and bool(((rel < 0) == (off < 0)).all())
    qn = QNetInference(eng.q_online, max_batch=c.rows // cfg.window_length)
    q_ring = qn.forward_u8(rp.obs_base, rp.frame_off_all.view(-1, cfg.window_length)).clone()
    q_msg = qn.forward_u8(msg.data_ptr() + c.off["frames"][0], rel.view(-1, cfg.window_length)).clone()
    assert torch.equal(q_ring, q_msg)


def _worker(rank, world, port, ret, prefetch=2):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), SRLX_CHECK_HEADERS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig
        from simple_distributed_rl_amd.device.replay_role import ReplayRoleRainbow

        torch.cuda.set_device(0)
        cfg = RainbowDeviceConfig(n_envs=16, batch_size=8, memory_capacity=2 * 16 * 40, memory_warmup_size=64, target_model_update_interval=5, seed=0)
        top = ReplayRoleRainbow(cfg, 0, episode_len=9, sync_interval=4, prefetch=prefetch, updates=1)
        w0 = float(top.flat.double().sum()) if top.role != "replay" else None
        T = 40
        for _ in range(T):
            top.step()
        top.finish()
        d = top.info()
        if top.role != "replay":
            d["weights_moved"] = float(top.flat.double().sum()) != w0
            d["weights_sum"] = round(float(top.flat.double().sum()), 6)
        ret[rank] = d
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("prefetch", [pytest.param(2, marks=pytest.mark.slow), 5])  # 5 = the reference's queue depth (play_mp_memory.py:595-621)
def test_three_roles_actors_replay_learner(prefetch):
    """Every transfer between the replay and the learner rank of a lock-step is one dist.batch_isend_irecv group per side; whether a batch / write-back is valid
    is computed on the host from the lock-step it belongs to, and SRLX_CHECK_HEADERS=1 (set in the workers) compares that with the header that travelled."""
    world = 4  # learner, replay, two actor ranks
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, prefetch), nprocs=world, join=True)
    r = dict(ret)
    learner, replay, actors = r[0], r[1], [r[2], r[3]]
    assert learner["role"] == "learner" and replay["role"] == "replay" and all(a["role"] == "actor" for a in actors)
    assert all(a["env_steps_local"] == 40 * 16 for a in actors) and learner["env_steps_local"] == 0 and replay["env_steps_local"] == 0
    assert replay["memory"] > 64 and replay["served"] == 40  # one batch message per lock-step, warm or not
    # the learner trained on every WARM batch that arrived `prefetch` lock-steps before the end, and every update came back as a priority write-back
    assert learner["train_count"] >= 40 - 3 - 1 - prefetch - 2 and np.isfinite(learner["loss"])  # warm from lock-step 2 on (64 / 32 environments), trained from prefetch + 1 lock-steps later
    assert learner["train_count"] - 1 <= replay["write_backs"] <= learner["train_count"]
    # the weights reached the actor ranks: after the broadcast of the last multiple of sync_interval all three hold the learner's parameters of that moment;
    # they moved away from the initialisation on every rank that acts or learns
    assert learner["weights_moved"] and all(a["weights_moved"] for a in actors)
    assert actors[0]["weights_sum"] == actors[1]["weights_sum"]


def test_runner_train_mp_with_a_replay_gpu():
    """`Runner.train_mp(actor_num, actor_devices=[...], memory_device="cuda:k")`: the reference's enable_mp_memory topology on the device path -- this
    process is the learner rank, rank 1 the replay GPU, the actors ranks 2.. (here all four ranks time-share the test GPU over gloo).  Trainer-side
    hooks fire, the stop rule is max_train_count, the trained weights come back into runner.parameter."""
    import simple_distributed_rl_amd as srl
    from simple_distributed_rl_amd.algorithms import rainbow
    from simple_distributed_rl_amd.base.run.callback import RunCallback

    cfg = rainbow.Config()
    cfg.set_atari_config()
    cfg.enable_noisy_dense = False
    cfg.window_length = 4
    cfg.memory.capacity, cfg.memory.warmup_size = 2 * 16 * 40, 64
    cfg.hidden_block.set_dueling_network((32,))
    cfg.batch_size = 8
    runner = srl.Runner(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(hw=(84, 84), n_actions=4, episode_len=7)), cfg)
    runner.set_vector_envs(16)
    before = {k: v.detach().clone().cpu() for k, v in runner.parameter.q_online.state_dict().items()}
    seen = []

    class TCB(RunCallback):
        def on_trainer_start(self, context, state, **kw):
            seen.append("start")

        def on_train_after(self, context, state, **kw):
            seen.append(state.train_count)

        def on_trainer_end(self, context, state, **kw):
            seen.append("end")

    st = runner.train_mp(actor_num=2, actor_devices=["cuda:0", "cuda:0"], memory_device="cuda:0", max_train_count=20, timeout=300, callbacks=[TCB()],
                         sync_interval_steps=4, mem_to_train_queue_capacity=2)
    assert runner.vector_reason == ""
    assert st.end_reason == "max_train_count over." and 20 <= st.train_count < 20 + 16
    assert seen[0] == "start" and seen[-1] == "end" and seen[-2] == st.train_count
    after = runner.parameter.q_online.state_dict()
    assert any(not torch.equal(before[k], after[k].cpu()) for k in before)
