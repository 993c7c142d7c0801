"""oracle/gen_golden_agent57_84.py -- TEST INFRASTRUCTURE ONLY.  One full `Trainer.train()` of the reference's Agent57_light
(srl/algorithms/agent57_light/model_torch.py:263-443 with `calc_target_q`, agent57_light.py:218-268) at the BENCHMARK geometry -- 84 x 84 x 4 frames, 6 actions,
dueling 512, embedding 32 -> 128, RND 128, UVFA inputs = previous extrinsic reward + one-hot actor (the reference's defaults) -- run by the imported reference on CPU
torch, so that the all-libsrlx update of device/agent57_fast.py (round 6) is pinned on the reference directly: losses, signed TD errors, priorities, and for every
parameter tensor of the four trained networks 2048 sampled entries of ITS GRADIENT (`p.grad` at `optimizer.step()`) and of its Adam step.

Run here, where /root/reference is importable:  PYTHONPATH=/root/reference python oracle/gen_golden_agent57_84.py
Only data travels (tests/golden/train_step_agent57_light84.npz):
  frames uint8 [B][5][84][84]   the window + 1 consecutive frames of every item (s_0 = frames[b, 0:4], s_1 = frames[b, 1:5], oldest first)
  actions, rewards_ext, rewards_int, dones, prev_actions, prev_rewards_ext, prev_rewards_int, actor_idx, weights   [B]
  td_ext, td_int, priorities [B]; ext_loss, int_loss, emb_loss, lifelong_loss
  pos.<net>.<key> / grad.<net>.<key> / upd.<net>.<key>: sampled positions, gradient entries, (after - before) entries; gsum / gabs: float64 sums of the gradient
The 33 M weights of the seven networks are NOT stored: `recipe_networks` below regenerates them (gen_golden_qnet84.recipe_state_dict, one seed per network;
LayerNorm weights get + 1).
"""
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
B, A, NA = 8, 6, 8
SEEDS = dict(q_ext=20261001, q_int=20261002, q_ext_target=20261003, q_int_target=20261004, emb=20261005, lifelong_target=20261006, lifelong_train=20261007)
NETS = ("q_ext", "q_int", "q_ext_target", "q_int_target", "emb", "lifelong_target", "lifelong_train")


def recipe_networks(keys_shapes_by_net):
    """{net: [(key, shape)]} -> {net: {key: float32 array}}: pure numpy, identical on every platform."""
    from gen_golden_qnet84 import recipe_state_dict

    out = {}
    for net, ks in keys_shapes_by_net.items():
        sd = recipe_state_dict(ks, "init", SEEDS[net])
        for k in sd:
            if "normalize.weight" in k:
                sd[k] = (sd[k] + 1.0).astype(np.float32)
        out[net] = sd
    return out


def make_items(seed=23):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (B, 5, 84, 84), dtype=np.uint8)
    frames[2, 0:2] = 0  # zero history in the two oldest frames of s_0 (an episode start)
    d = dict(frames=frames, actions=rng.integers(0, A, B).astype(np.int32), rewards_ext=rng.integers(-1, 2, B).astype(np.float32),
             rewards_int=(rng.random(B) * 3).astype(np.float32), dones=(rng.random(B) < 0.8).astype(np.float32), prev_actions=rng.integers(0, A, B).astype(np.int32),
             prev_rewards_ext=rng.integers(-1, 2, B).astype(np.float32), prev_rewards_int=(rng.random(B) * 3).astype(np.float32),
             actor_idx=rng.integers(0, NA, B).astype(np.int32), weights=(0.3 + 0.7 * rng.random(B)).astype(np.float32))
    d["dones"][1] = 0.0  # one terminal transition at least
    return d


def main():
    import torch

    import srl
    from srl.algorithms import agent57_light
    from srl.base.context import RunContext
    from srl.base.env import registration

    import _golden_env  # noqa: F401

    torch.set_num_threads(8)
    registration.register("TinyImageEnvGolden", entry_point="_golden_env:TinyImageEnv", check_duplicate=False)
    env = srl.EnvConfig("TinyImageEnvGolden", kwargs=dict(hw=84, actions=A)).make()
    cfg = agent57_light.Config(batch_size=B, actor_num=NA, target_model_update_interval=5)
    cfg.window_length = 4
    cfg.memory.warmup_size, cfg.memory.capacity, cfg.memory.compress = B, 1000, False
    cfg.hidden_block.set_dueling_network((512,))
    cfg.set_torch()
    cfg.setup(env)
    torch.manual_seed(0)
    parameter = cfg.make_parameter()
    memory = cfg.make_memory()
    trainer = cfg.make_trainer(parameter, memory)
    trainer.setup(RunContext())
    nets = dict(q_ext=parameter.q_ext_online, q_int=parameter.q_int_online, q_ext_target=parameter.q_ext_target, q_int_target=parameter.q_int_target,
                emb=parameter.emb_network, lifelong_target=parameter.lifelong_target, lifelong_train=parameter.lifelong_train)
    ks = {n: [(k, tuple(v.shape)) for k, v in m.state_dict().items()] for n, m in nets.items()}
    before = recipe_networks(ks)
    for n, m in nets.items():
        m.load_state_dict({k: torch.tensor(v) for k, v in before[n].items()})
    it = make_items()
    eye, f = np.identity(A, dtype=np.float32), it["frames"]
    st = lambda b, k: np.stack([f[b, k + c] for c in range(4)], axis=-1).astype(np.float32) / 255  # noqa: E731
    batches = [[st(b, 0), st(b, 1), eye[it["actions"][b]], [], float(it["rewards_ext"][b]), np.float32(it["rewards_int"][b]), int(it["dones"][b]), eye[it["prev_actions"][b]],
                float(it["prev_rewards_ext"][b]), np.float32(it["prev_rewards_int"][b]), int(it["actor_idx"][b])] for b in range(B)]
    rec = {}
    memory.sample = lambda *a, **k: (batches, it["weights"].copy(), list(range(B)))
    memory.update = lambda update_args, priorities, step: rec.__setitem__("priorities", np.asarray(priorities).copy())
    _uq = trainer._update_q
    tds = []

    def uq(*a, **k):
        td, loss = _uq(*a, **k)
        tds.append(np.asarray(td).copy())
        return td, loss

    trainer._update_q = uq
    # p.grad at optimizer.step(): what loss.backward() left (model_torch.py:437-439, 345-347, 359-361)
    owner = {id(p): (n, k) for n, m in nets.items() for k, p in m.named_parameters()}
    grads = {}
    _step = torch.optim.Adam.step

    def step(self, *a, **k):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    grads[owner[id(p)]] = p.grad.detach().clone().numpy()
        return _step(self, *a, **k)

    torch.optim.Adam.step = step
    trainer.train_count = 1  # not a target-sync step
    trainer.train()
    torch.optim.Adam.step = _step
    save = dict(it)
    save.update(td_ext=tds[0].astype(np.float32), td_int=tds[1].astype(np.float32), priorities=rec["priorities"].astype(np.float32),
                ext_loss=np.float32(trainer.info["ext_loss"]), int_loss=np.float32(trainer.info["int_loss"]), emb_loss=np.float32(trainer.info["emb_loss"]),
                lifelong_loss=np.float32(trainer.info["lifelong_loss"]), lr_ext=np.float64(cfg.lr_ext), lr_int=np.float64(cfg.lr_int), episodic_lr=np.float64(cfg.episodic_lr),
                lifelong_lr=np.float64(cfg.lifelong_lr), actor_num=np.int64(NA), n_actions=np.int64(A))
    for n in NETS:
        save["keys." + n] = np.array([k for k, _ in ks[n]])
        save["shapes." + n] = np.array([str(tuple(s)) for _, s in ks[n]])
    prng = np.random.default_rng(101)
    for n in ("q_ext", "q_int", "emb", "lifelong_train"):
        after = {k: v.detach().numpy() for k, v in nets[n].state_dict().items()}
        for k, _ in ks[n]:
            g = grads[(n, k)].astype(np.float64).reshape(-1)
            d = (after[k].astype(np.float64) - before[n][k].astype(np.float64)).reshape(-1)
            pos = np.sort(prng.choice(d.size, size=min(2048, d.size), replace=False))
            save[f"pos.{n}.{k}"] = pos.astype(np.int64)
            save[f"grad.{n}.{k}"] = g[pos].astype(np.float32)
            save[f"upd.{n}.{k}"] = d[pos].astype(np.float32)
            save[f"gsum.{n}.{k}"] = np.float64(g.sum())
            save[f"gabs.{n}.{k}"] = np.float64(np.abs(g).sum())
            save[f"gmax.{n}.{k}"] = np.float64(np.abs(g).max())
    np.savez_compressed(os.path.join(OUT, "train_step_agent57_light84.npz"), **save)
    print("train_step_agent57_light84:", {k: round(float(v), 6) for k, v in trainer.info.items() if "loss" in k}, "priorities", rec["priorities"][:4])


if __name__ == "__main__":
    main()
