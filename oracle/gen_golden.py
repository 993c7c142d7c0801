#!/usr/bin/env python3
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY.

Generates the committed golden vectors under tests/golden/ by IMPORTING THE
REFERENCE (pocokhc/simple_distributed_rl, mounted read-only at /root/reference)
in this container and recording its inputs/outputs on seeded synthetic data.
The reference itself never travels to the GPU box: only the .npz files do.

Run (CPU only, ~1 min):
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--only per,td,...]

Fixtures written (each is data only: inputs + the reference's outputs):
    per_trace_<name>.npz   scripted add/sample/update traces of
                           srl.rl.memories.priority_memories.proportional_memory.ProportionalMemory
                           with the exact random.random() stream it consumed
    functions.npz          srl.rl.functions rescaling/inverse_rescaling/create_*_list
    target_q_<name>.npz    srl.algorithms.rainbow / dqn calc_target_q I/O
    train_step_<name>.npz  one full Trainer.train() (loss, q, priorities, new weights)
    stack_trace.npz        WorkerRun frame stacking sequence
"""
import argparse
import os
import random
import sys

import numpy as np

REF = os.environ.get("SRL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

sys.dont_write_bytecode = True
if REF not in sys.path:
    sys.path.insert(0, REF)


# ----------------------------------------------------------------------------------------
# PER traces
# ----------------------------------------------------------------------------------------
OP_ADD_NONE, OP_ADD_PY, OP_SAMPLE, OP_UPDATE_F32, OP_UPDATE_F64 = 0, 1, 2, 3, 4


class _Recorder:
    """Wraps random.random() inside the reference module to capture the stream it consumes."""

    def __init__(self):
        self.buf = []

    def random(self):
        v = random.random()
        self.buf.append(v)
        return v

    def __getattr__(self, name):  # everything else -> the real module
        return getattr(random, name)


def _record_per_trace(name, capacity, alpha, beta_initial, beta_steps, has_duplicate, epsilon, script, seed):
    """script: generator function(mem_api) that yields ops; we run them on the reference and log."""
    import srl.rl.memories.priority_memories.proportional_memory as pm

    rec = _Recorder()
    pm.random = rec  # the module does `import random`; swap its binding only
    try:
        mem = pm.ProportionalMemory(
            capacity,
            alpha=alpha,
            beta_initial=beta_initial,
            beta_steps=beta_steps,
            has_duplicate=has_duplicate,
            epsilon=epsilon,
        )
        random.seed(seed)
        rng = np.random.default_rng(seed)

        op_code, op_a, op_b, op_off_u, op_off_i, op_off_p, op_off_w = [], [], [], [], [], [], []
        pool_u, pool_idx, pool_w, pool_pri, pool_tx = [], [], [], [], []
        counter = [0]

        n_used = []

        def log(code, a=0.0, b=0):
            n_used.append(0)
            op_code.append(code)
            op_a.append(float(a))
            op_b.append(int(b))
            op_off_u.append(len(pool_u))
            op_off_i.append(len(pool_idx))
            op_off_p.append(len(pool_pri))
            op_off_w.append(len(pool_w))

        class API:
            def add(self, priority=None):
                item = counter[0]
                counter[0] += 1
                if priority is None:
                    log(OP_ADD_NONE)
                else:
                    log(OP_ADD_PY, float(priority))
                mem.add(item, priority)

            def sample(self, batch_size, step):
                log(OP_SAMPLE, step, batch_size)
                rec.buf = []
                batches, weights, indices = mem.sample(batch_size, step)
                pool_u.extend(rec.buf)
                n_used[-1] = len(rec.buf)  # number of uniforms this call consumed
                pool_idx.extend(int(i) for i in indices)
                pool_w.extend(float(w) for w in weights)
                return batches, weights, indices

            def update(self, indices, priorities):
                priorities = np.asarray(priorities)
                code = OP_UPDATE_F32 if priorities.dtype == np.float32 else OP_UPDATE_F64
                log(code, 0.0, len(indices))
                pool_idx.extend(int(i) for i in indices)
                pool_pri.extend(float(p) for p in priorities)
                tx = (np.abs(priorities) + mem.epsilon) ** mem.alpha  # what :172 computes on this host
                pool_tx.extend(float(p) for p in tx)
                mem.update(list(indices), priorities)

            @property
            def mem(self):
                return mem

        api = API()
        script(api, rng)

        assert len(n_used) == len(op_code)
        np.savez_compressed(
            os.path.join(OUT, f"per_trace_{name}.npz"),
            capacity=np.int64(capacity),
            alpha=np.float64(alpha),
            beta_initial=np.float64(beta_initial),
            beta_steps=np.float64(beta_steps),
            has_duplicate=np.int64(has_duplicate),
            epsilon=np.float64(epsilon),
            seed=np.int64(seed),
            op_code=np.asarray(op_code, np.int64),
            op_a=np.asarray(op_a, np.float64),
            op_b=np.asarray(op_b, np.int64),
            op_n_uniforms=np.asarray(n_used, np.int64),
            op_off_u=np.asarray(op_off_u, np.int64),
            op_off_i=np.asarray(op_off_i, np.int64),
            op_off_p=np.asarray(op_off_p, np.int64),
            op_off_w=np.asarray(op_off_w, np.int64),
            pool_u=np.asarray(pool_u, np.float64),
            pool_idx=np.asarray(pool_idx, np.int64),
            pool_w=np.asarray(pool_w, np.float64),
            pool_pri=np.asarray(pool_pri, np.float64),
            pool_tx=np.asarray(pool_tx, np.float64),
            final_tree=np.asarray(mem.tree.tree, np.float64),
            final_max_priority=np.float64(mem.max_priority),
            final_size=np.int64(mem.size),
            final_write=np.int64(mem.tree.write),
            final_next_random=np.float64(random.random()),
            script_uses_random=np.int64(name.startswith('speedtest')),
        )
        print(f"per_trace_{name}: {len(op_code)} ops, {len(pool_u)} uniforms, root={mem.tree.total():.6f}")
    finally:
        pm.random = random


def gen_per():
    # --- (1) the reference's own statistical test scenario, shortened
    #     (tests/quick/rl/memories/test_priority_memories.py:29-91): capacity 10,
    #     alpha .8, beta_initial 1, no duplicates -> exercises the duplicate retry loop
    def script_small(api, rng):
        for _ in range(100):
            api.add(0)
        for i in range(10):
            api.add(i + 1)
        for it in range(300):
            _, _, idx = api.sample(5, 1)
            api.update(idx, np.array([rng.integers(1, 11) for _ in idx]))  # int64 -> float64 path

    _record_per_trace("small_nodup", 10, 0.8, 1.0, 10, False, 1e-4, script_small, seed=1)

    def script_small_dup(api, rng):
        for _ in range(100):
            api.add(0)
        for i in range(10):
            api.add(i + 1)
        for it in range(300):
            _, _, idx = api.sample(5, it)
            api.update(idx, np.array([float(rng.random()) * 3 for _ in idx]))

    _record_per_trace("small_dup", 10, 0.8, 0.4, 1000, True, 1e-4, script_small_dup, seed=2)

    # --- (2) Rainbow/atari parameters (rainbow.py:139-143): alpha .5, beta0 .4, beta_steps 1e6, eps 1e-4;
    #     float32 priorities from the trainer (model_torch.py:113), priority=None adds from the worker.
    #     Non power-of-two capacity, partially filled ring, wrap-around.
    def script_rainbow(api, rng):
        for _ in range(1500):
            api.add(None)
        step = 0
        for it in range(400):
            for _ in range(4):
                api.add(None)
            _, _, idx = api.sample(32, step)
            pri = np.abs(rng.standard_normal(len(idx))).astype(np.float32)
            api.update(idx, pri)
            step += 1

    _record_per_trace("rainbow_cap3000", 3000, 0.5, 0.4, 1_000_000, True, 1e-4, script_rainbow, seed=3)

    # --- (3) speedtest.py-shaped loop (tests/quick/rl/memories/speedtest.py:28-58), scaled down:
    #     add with a python-float priority, sample 64, update with a python list (float64 path)
    def script_speed(api, rng):
        step = 0
        for _ in range(5000):
            api.add(random.random())
            step += 1
        for _ in range(200):
            api.add(random.random())
            step += 1
            _, _, idx = api.sample(64, step)
            api.update(idx, [random.random() for _ in range(64)])

    _record_per_trace("speedtest_cap4096", 4096, 0.8, 0.4, 1000, True, 1e-4, script_speed, seed=4)

    # --- (4) distributed-actor style adds (priority computed on the actor, rainbow.py:389-398),
    #     NOTE: the reference hands a numpy float32 *scalar* to add() there; under NEP-50 numpy (>=2.0)
    #     that scalar silently turns the tree's ancestor sums into float32 (python-float + np.float32 ->
    #     np.float32), under numpy 1.x it does not.  That is a dtype leak, not an algorithm; the build
    #     (like the reference's C++ twin, proportional_memory.cpp:124 `std::optional<double>`) widens the
    #     scalar with float() first, so the trace does the same.
    #     out-of-order sample/sample/update/sample/update/update (test_memories.py:58-77), power-of-two capacity
    def script_mp(api, rng):
        for _ in range(700):
            api.add(float(np.float32(abs(rng.standard_normal()))))
        for it in range(100):
            api.add(float(np.float32(abs(rng.standard_normal()))))
            _, _, i1 = api.sample(16, it * 5000)
            _, _, i2 = api.sample(16, it * 5000)
            api.update(i1, np.abs(rng.standard_normal(16)).astype(np.float32))
            _, _, i3 = api.sample(16, it * 5000)
            api.update(i2, np.abs(rng.standard_normal(16)).astype(np.float32))
            api.update(i3, np.abs(rng.standard_normal(16)).astype(np.float32))

    _record_per_trace("mp_cap1024", 1024, 0.5, 0.4, 200_000, True, 1e-4, script_mp, seed=5)

    # --- (5) duplicates inside one update batch + no-duplicate sampling on a larger tree
    def script_dups(api, rng):
        for _ in range(257):
            api.add(float(rng.random()))
        for it in range(150):
            _, _, idx = api.sample(24, it)
            idx = list(idx) + list(idx[:8])  # repeated tree indices in one update (:173-175)
            api.update(idx, np.abs(rng.standard_normal(len(idx))).astype(np.float32))

    _record_per_trace("dupupdate_cap257", 257, 0.5, 0.4, 100, False, 1e-4, script_dups, seed=6)


def gen_per_is_kat():
    """IS-weight known-answer vectors (test_priority_memories.py:97-117,150-176)."""
    from srl.rl.memories.priority_memories.proportional_memory import ProportionalMemory

    rows = []
    for alpha in [0, 0.2, 0.5, 0.8, 1.0]:
        eps = 1e-4
        mem = ProportionalMemory(capacity=10, alpha=alpha, beta_initial=1, epsilon=eps, has_duplicate=False)
        pri = [1, 2, 4, 3]
        for i, p in enumerate(pri):
            mem.add((i, i, i, i), priority=p)
        random.seed(7)
        batches, weights, idx = mem.sample(4, step=1)
        true_p = [(t + eps) ** alpha for t in pri]
        s = sum(true_p)
        tw = np.array([(4 * (p / s)) ** -1 for p in true_p])
        tw /= tw.max()
        rows.append((alpha, [b[0] for b in batches], list(weights), list(idx), list(tw)))
    np.savez_compressed(
        os.path.join(OUT, "per_is_kat.npz"),
        alpha=np.array([r[0] for r in rows]),
        item=np.array([r[1] for r in rows], np.int64),
        weights=np.array([r[2] for r in rows]),
        indices=np.array([r[3] for r in rows], np.int64),
        true_weights=np.array([r[4] for r in rows]),
    )
    print("per_is_kat: ok")


# ----------------------------------------------------------------------------------------
GENERATORS = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    GENERATORS.update(
        per=gen_per,
        per_is=gen_per_is_kat,
    )
    try:
        from gen_golden_algo import ALGO_GENERATORS  # noqa: E402  (same directory)

        GENERATORS.update(ALGO_GENERATORS)
    except ImportError:
        pass
    from gen_golden_agent57 import AGENT57_GENERATORS  # noqa: E402

    GENERATORS.update(AGENT57_GENERATORS)
    only = [s for s in args.only.split(",") if s]
    for name, fn in GENERATORS.items():
        if only and name not in only:
            continue
        fn()


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
