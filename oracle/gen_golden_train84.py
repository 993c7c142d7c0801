"""oracle/gen_golden_train84.py -- TEST INFRASTRUCTURE ONLY.  One full `Trainer.train()` of the reference's Rainbow (srl/algorithms/rainbow/model_torch.py:85-122
with `calc_target_q`, srl/algorithms/rainbow/rainbow.py:185-287) at the BENCHMARK geometry -- 84 x 84 x 4 frames, 6 actions, dueling 512, n-step 3, double DQN --
run by the imported reference on CPU torch, so that the SHIPPED learner path of the device engine (the round-4 lock-step's update: fused draw + gather, split-bf16
forward, fused TD / Huber / priority head, hand-written backward, Adam fused into the first dense layer's weight gradient) is pinned on the reference directly and
not only through the 8 x 8 toy of train_step_rainbow.npz.

Run here, where /root/reference is importable:  PYTHONPATH=/root/reference python oracle/gen_golden_train84.py
Only data travels (tests/golden/train_step_rainbow84.npz):
  frames uint8 [B][7][84][84]  the n + window = 7 consecutive frames of every item (state k of item b = frames[b, k : k + 4], oldest first)
  actions / reward / done [B][3], weights [B] (importance weights handed to the trainer)
  outputs of the reference: target_q [B], q0 [B][6] (online Q of s_0), loss, priorities [B]
  the Adam step: for every parameter tensor 2048 sampled entries of (after - before) (`upd.<key>`, positions `pos.<key>`), and float64 sums of after - before
  the gradients (round 6): the same positions of every `p.grad` as loss.backward() left it (`grad.<key>`), its largest magnitude and its float64 sum (`gmax.`, `gsum.`)
The 8.0 M weights of the online and the target network are NOT stored: gen_golden_qnet84.recipe_state_dict regenerates them (kind "init", seeds 20260929 / 20260930).
"""
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
SEED_ONLINE, SEED_TARGET = 20260929, 20260930
B, N, A = 16, 3, 6


def make_items(seed=17):
    """The sampled batch, as data: frames, actions, rewards, terminal flags, importance weights.  Item 3 ends its episode at its second transition, item 5 at its last."""
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (B, N + 4, 84, 84), dtype=np.uint8)
    actions = rng.integers(0, A, (B, N)).astype(np.int32)
    reward = rng.integers(-1, 2, (B, N)).astype(np.float32)
    done = np.zeros((B, N), np.float32)
    done[3, 1] = 1.0
    done[5, 2] = 1.0
    weights = (0.3 + 0.7 * rng.random(B)).astype(np.float32)
    return frames, actions, reward, done, weights


def batches_from(frames, actions, reward, done):
    """The reference's item layout (rainbow.py:377-400): n + 1 rows [state (84, 84, 4) float32, onehot, reward, terminated, invalid actions]; after a terminal
    transition the rows are padding (state repeated, reward 0, terminated 1: rainbow.py:354-372)."""
    out = []
    for b in range(B):
        rows = []
        ended = False
        for k in range(N + 1):
            st = np.stack([frames[b, k + c] for c in range(4)], axis=-1).astype(np.float32) / 255
            if k == 0:
                rows.append([st, None, None, None, None])
                continue
            onehot = [1.0 if a == actions[b, k - 1] else 0.0 for a in range(A)]
            if ended:
                rows.append([rows[-1][0], onehot, 0.0, 1, []])
            else:
                rows.append([st, onehot, float(reward[b, k - 1]), int(done[b, k - 1]), []])
                ended = bool(done[b, k - 1])
        out.append(rows)
    return out


def main():
    import torch

    from gen_golden_qnet84 import _build_reference_net, recipe_state_dict

    torch.set_num_threads(8)
    env, rl_config = _build_reference_net()
    rl_config.batch_size = B
    rl_config.memory.warmup_size = B
    rl_config.enable_double_dqn = True
    torch.manual_seed(0)
    parameter = rl_config.make_parameter()
    memory = rl_config.make_memory()
    trainer = rl_config.make_trainer(parameter, memory)
    from srl.base.context import RunContext

    trainer.setup(RunContext())
    keys_shapes = [(k, tuple(v.shape)) for k, v in parameter.q_online.state_dict().items()]
    sd_on = recipe_state_dict(keys_shapes, "init", SEED_ONLINE)
    sd_tg = recipe_state_dict(keys_shapes, "init", SEED_TARGET)
    parameter.q_online.load_state_dict({k: torch.tensor(v) for k, v in sd_on.items()})
    parameter.q_target.load_state_dict({k: torch.tensor(v) for k, v in sd_tg.items()})
    frames, actions, reward, done, weights = make_items()
    batches = batches_from(frames, actions, reward, done)
    rec = {}
    memory.sample = lambda *a, **k: (batches, weights.copy(), list(range(B)))
    memory.update = lambda update_args, priorities, step: rec.__setitem__("priorities", np.asarray(priorities).copy())
    memory.is_warmup_needed = lambda: False
    _calc = parameter.calc_target_q

    def calc(bs):
        out = _calc(bs)
        rec["target_q"] = np.asarray(out[0]).copy()
        return out

    parameter.calc_target_q = calc
    orig_forward = parameter.q_online.forward
    holder = {}

    def fwd(x):
        y = orig_forward(x)
        if y.requires_grad:
            holder["q"] = y.detach().clone()
        return y

    parameter.q_online.forward = fwd
    # p.grad as `loss.backward()` left it (model_torch.py:107-108), caught at `optimizer.step()` (:109): Adam's FIRST step is lr * g / (|g| + eps) ~ lr * sign(g),
    # so the step alone pins signs -- the gradient entries themselves pin the hand-written backward's magnitudes on the reference
    names = {id(p): k for k, p in parameter.q_online.named_parameters()}
    grads = {}
    _step = torch.optim.Adam.step

    def step(self, *a, **k):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None and id(p) in names:
                    grads[names[id(p)]] = p.grad.detach().clone().numpy()
        return _step(self, *a, **k)

    torch.optim.Adam.step = step
    trainer.train_count = 1  # not a sync step
    trainer.train()
    torch.optim.Adam.step = _step
    parameter.q_online.forward = orig_forward
    after = {k: v.detach().numpy() for k, v in parameter.q_online.state_dict().items()}
    save = dict(frames=frames, actions=actions, reward=reward, done=done, weights=weights, target_q=rec["target_q"].astype(np.float32), q0=holder["q"].numpy(),
                loss=np.float32(trainer.info["loss"]), priorities=rec["priorities"].astype(np.float32), lr=np.float64(rl_config.lr), discount=np.float64(rl_config.discount),
                seed_online=np.int64(SEED_ONLINE), seed_target=np.int64(SEED_TARGET), keys=np.array([k for k, _ in keys_shapes]),
                shapes=np.array([str(tuple(s)) for _, s in keys_shapes]))
    prng = np.random.default_rng(99)
    for k, _ in keys_shapes:
        d = (after[k].astype(np.float64) - sd_on[k].astype(np.float64)).reshape(-1)
        pos = np.sort(prng.choice(d.size, size=min(2048, d.size), replace=False))
        save["pos." + k] = pos.astype(np.int64)
        save["upd." + k] = d[pos].astype(np.float32)
        save["sum." + k] = np.float64(d.sum())
        save["abs." + k] = np.float64(np.abs(d).sum())
        g = grads[k].astype(np.float64).reshape(-1)
        save["grad." + k] = g[pos].astype(np.float32)
        save["gmax." + k] = np.float64(np.abs(g).max())
        save["gsum." + k] = np.float64(g.sum())
    np.savez_compressed(os.path.join(OUT, "train_step_rainbow84.npz"), **save)
    print(f"train_step_rainbow84: loss={float(trainer.info['loss']):.6f} target range [{rec['target_q'].min():.4f}, {rec['target_q'].max():.4f}]")


if __name__ == "__main__":
    main()
