"""a6: the uniform ReplayBuffer (srl/rl/memories/priority_memories/replay_buffer.py:10-55) against a seeded add / sample trace RECORDED FROM THE REFERENCE
(tests/golden/uniform_replay_trace.npz, oracle/gen_golden_uniform.py): which stored items every batch holds, and where Python's generator stands afterwards.
CPU: the plugin-level class.  GPU: the device store -- frame ring + tree in HBM -- drawing through `DeviceReplay.draw_like_random_sample` returns the same
items (their identity travels as the reward of the stored transition), without replacement, all weights 1; and the engines' own on-device draw of a uniform
memory never holds an item twice either."""
import os
import random

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
Z = np.load(os.path.join(ROOT, "tests", "golden", "uniform_replay_trace.npz"))


def test_plugin_replay_buffer_replays_the_reference_trace():
    from simple_distributed_rl_amd.rl.memories.priority_memories.replay_buffer import ReplayBuffer

    mem = ReplayBuffer(int(Z["capacity"]))
    random.seed(int(Z["seed"]))
    added = 0
    for upto, want, probe in zip(Z["added_before_sample"], Z["sampled_items"], Z["probe"]):
        while added < upto:
            mem.add(added, None)
            added += 1
        items, w, upd = mem.sample(int(Z["batch_size"]), step=added)
        assert items == list(want) and w == [1.0] * len(items) and upd == []
        assert random.random() == probe


@pytest.mark.gpu
def test_device_store_draws_the_reference_items():
    import torch

    from simple_distributed_rl_amd.device.replay import DeviceReplay

    cap, B, F = int(Z["capacity"]), int(Z["batch_size"]), 64
    rep = DeviceReplay(1, cap + 2, F, 1, 1, 4, B, True, False, alpha=0.0, warmup_size=1, seed=0, has_duplicate=False)  # one lane: leaf j = the j-th stored item
    assert rep.capacity == cap
    dev = rep.dev
    rep.reset_all(torch.zeros((1, F), dtype=torch.uint8, device=dev))
    random.seed(int(Z["seed"]))
    added = 0
    zero8 = torch.zeros(1, dtype=torch.uint8, device=dev)
    for upto, want, probe in zip(Z["added_before_sample"], Z["sampled_items"], Z["probe"]):
        while added < upto:  # transition number `added`: its reward carries its identity, its next frame its number mod 251
            rep.commit(torch.tensor([added % 4], dtype=torch.int32, device=dev), torch.tensor([float(added)], device=dev), zero8, zero8,
                       torch.full((1, F), added % 251, dtype=torch.uint8, device=dev))
            added += 1
        assert rep.length() == min(added, cap)
        idx = rep.draw_like_random_sample()
        b = rep.gather_drawn(all_states=False)
        torch.cuda.synchronize()
        got = b.rewards[:, 0].cpu().numpy().astype(np.int64)
        np.testing.assert_array_equal(got, want)  # the items the reference's list handed out, in its order
        assert len(set(idx.cpu().tolist())) == B and float(b.weights.min()) == 1.0 == float(b.weights.max())
        np.testing.assert_array_equal(b.actions[:, 0].cpu().numpy(), want % 4)
        np.testing.assert_array_equal(b.obs[:, 1, 0, 0].cpu().numpy(), (want % 251).astype(np.float32) / 255)  # s_1 of item i is the frame that came with it
        assert random.random() == probe  # Python's generator stands where the reference left it
    # the device-side draw of the same memory (what a captured learner graph uses): uniform over the stored items, never an item twice
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    seen = np.zeros(cap, np.int64)
    for _ in range(200):
        b = rep.sample(step)
        torch.cuda.synchronize()
        slots = b.indices.cpu().numpy() - (cap - 1)
        assert len(set(slots.tolist())) == B and float(b.weights.min()) == 1.0 == float(b.weights.max())
        np.add.at(seen, slots, 1)
    assert seen.min() > 0 and seen.max() < 4 * seen.mean()
    rep.close()
