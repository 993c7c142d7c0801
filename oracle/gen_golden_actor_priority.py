#!/usr/bin/env python3
"""oracle/gen_golden_actor_priority.py -- TEST INFRASTRUCTURE ONLY.  The initial priorities a DISTRIBUTED Rainbow actor of the reference
computes for its own items (srl/algorithms/rainbow/rainbow.py:389-398 and rainbow_nomultisteps.py:108-119: `abs(calc_target_q([batch]) -
q[action])`, because the learner's max_priority is not visible to an actor process).

Drives the reference's worker by hand with `context.distributed = True` on the tiny image environment, with `memory.add` wrapped to
record (item, priority); stores the network weights, the environment log and the priorities.  Data only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_actor_priority.py   ->  tests/golden/actor_priority_{n3,n1}.npz
"""
import os
import random
import sys

import numpy as np

REF = os.environ.get("SRL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def record(name, multisteps):
    import torch

    import srl
    from srl.base.context import RunContext
    from srl.utils import common

    from gen_golden_algo import _make_rainbow, _register_env

    _register_env()
    env_config, rl_config = _make_rainbow(multisteps=multisteps, double_dqn=True, rescale=False)
    env_config.kwargs = dict(hw=8, actions=4, ep_len=6, truncate=False, seed=9)
    rl_config.enable_reward_clip = True
    rl_config.epsilon = 0.3
    runner = srl.Runner(env_config, rl_config)
    runner.set_device("CPU")
    env = runner.make_env()
    parameter = runner.make_parameter()
    memory = runner.make_memory()
    worker = runner.make_worker(parameter, memory)
    got = []
    real_add = memory.add

    def add(batch, priority=None, **kw):
        got.append(None if priority is None else float(priority))
        return real_add(batch, priority, **kw)

    worker.worker.memory.add = add
    memory.add = add
    ctx = RunContext(env_config, rl_config)
    ctx.distributed, ctx.training, ctx.actor_num, ctx.actor_id = True, True, 1, 0
    common.set_seed(6)
    env.setup(ctx)
    worker.setup(ctx)
    steps = 0
    for ep in range(5):
        env.reset()
        worker.reset(0)
        while not env.done:
            a = worker.policy()
            env.step(a)
            worker.on_step()
            steps += 1
    sd = {k: v.detach().numpy() for k, v in parameter.q_online.state_dict().items()}
    log = env.unwrapped.log
    np.savez_compressed(
        os.path.join(OUT, f"actor_priority_{name}.npz"),
        priorities=np.array(got, np.float64), multisteps=np.int64(multisteps), seed=np.int64(6), epsilon=np.float64(rl_config.epsilon), steps=np.int64(steps),
        actions=np.array([l[1] for l in log], np.int32), **{"w:" + k: v for k, v in sd.items()},
    )
    print(name, "ok:", steps, "steps,", len(got), "items, priorities", np.round(got[:5], 5))


if __name__ == "__main__":
    record("n3", 3)
    record("n1", 1)
