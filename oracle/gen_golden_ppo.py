#!/usr/bin/env python3
"""oracle/gen_golden_ppo.py -- TEST INFRASTRUCTURE ONLY.  Pins the PPO arithmetic of the oracle and of the
srlx_ppo_* kernels on outputs of the imported reference.

The reference's `ppo` (srl/algorithms/ppo/ppo.py) is a TensorFlow model and cannot be imported in this image, but the
same clipped-surrogate and entropy arithmetic (ppo.py:126-137,152,166-167) and the same Normal / Categorical log-probability
(srl/rl/tf/distributions/*  ==  srl/rl/torch_/distributions/*) exist in the reference's TORCH code, which IS importable:
    srl/algorithms/ppo_v/torch_model.py:111-178   Trainer.train  (ratio, clipped surrogate, entropy, v / n_v)
    srl/rl/torch_/distributions/normal_dist_block.py:32-63, srl/rl/functions.py:232-238   Normal log-probability
    srl/rl/torch_/distributions/categorical_dist_block.py                                   Categorical log-probability
This script runs real `Trainer.train()` steps of ppo_v on seeded rollouts (a discrete run on the reference's Grid and a
continuous run on a small environment registered from outside its tree) and records, per step: the network outputs the
step saw (v, n_v, new log-probabilities, distribution parameters), the batch, and the losses the reference reported
(`info["loss_policy"]`, `info["loss_e"]`).  Only data is written.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ppo.py      ->  tests/golden/ppo_v_step_{discrete,continuous}.npz

What is NOT covered by reference outputs (no importable reference code runs it) and is pinned by known-answer tests
written out in tests/test_ppo_pinned.py instead: the GAE recursion of ppo.py:389-404 and the value-clip branch of
ppo.py:155-157.
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


def _continuous_env():
    """A 2-D continuous-action toy registered with the reference from outside its tree."""
    from srl.base.define import SpaceTypes
    from srl.base.env.base import EnvBase
    from srl.base.env import registration
    from srl.base.spaces.box import BoxSpace

    class Toy(EnvBase):
        def __init__(self):
            super().__init__()
            self.rng = np.random.default_rng(5)
            self.t = 0

        @property
        def action_space(self):
            return BoxSpace((2,), -1.0, 1.0, np.float32, SpaceTypes.CONTINUOUS)

        @property
        def observation_space(self):
            return BoxSpace((3,), -2.0, 2.0, np.float32, SpaceTypes.CONTINUOUS)

        @property
        def max_episode_steps(self):
            return 100

        @property
        def player_num(self):
            return 1

        def reset(self, **kw):
            self.t = 0
            self.s = self.rng.uniform(-1, 1, 3).astype(np.float32)
            return self.s

        def step(self, action):
            self.t += 1
            a = np.asarray(action, np.float32).reshape(-1)
            r = float(-np.sum((self.s[:2] - a) ** 2))
            self.s = np.clip(self.s + 0.3 * self.rng.standard_normal(3).astype(np.float32), -2, 2).astype(np.float32)
            return self.s, r, self.t >= 9, False

        def backup(self, **kw):
            return None

        def restore(self, d, **kw):
            pass

    mod = sys.modules[__name__]
    mod.Toy = Toy
    registration.register("ToyContinuousGolden", entry_point=__name__ + ":Toy", check_duplicate=False)
    return "ToyContinuousGolden"


def record(name, env_name, continuous, entropy_weight, clip_range, sgp):
    import torch

    import srl
    from srl.algorithms import ppo_v
    from srl.utils import common

    cfg = ppo_v.Config(batch_size=32, lr=0.001, discount=0.9, clip_range=clip_range, entropy_weight=entropy_weight, squashed_gaussian_policy=sgp)
    cfg.memory.warmup_size = 32
    cfg.memory.capacity = 2000
    runner = srl.Runner(env_name, cfg)
    runner.set_device("CPU")
    runner.set_seed(11)
    runner.rollout(max_steps=300)
    trainer = runner.make_trainer()
    trainer.setup(runner.context)
    common.set_seed(3)
    random.seed(3)
    steps = []
    for k in range(4):
        net = trainer.parameter.net
        state_before = {kk: vv.detach().clone() for kk, vv in net.state_dict().items()}
        rng_state = (random.getstate(), np.random.get_state(), torch.random.get_rng_state())
        trainer.train()  # the reference's own step: fills trainer.*_np with the batch it used and info with its losses
        info = dict(trainer.info.to_dict())
        # what that step's forward saw: the same network (weights before the step) on the same batch, through the reference's own modules
        after = {kk: vv.detach().clone() for kk, vv in net.state_dict().items()}
        net.load_state_dict(state_before)
        with torch.no_grad():
            st = torch.from_numpy(trainer.state_np.copy())
            ns = torch.from_numpy(trainer.n_state_np.copy())
            act = torch.from_numpy(trainer.action_np.copy())
            v, dist = net(st)
            n_v, _ = net(ns)
            if not continuous:
                new_logpi = dist.log_prob(act, keepdims=True)
                extra = dict(logits=dist.logits().numpy())
            else:
                new_logpi = dist.log_prob_sgp(act) if sgp else dist.log_prob(act)
                extra = dict(loc=dist.mean().numpy(), log_scale=torch.log(dist.stddev()).numpy(), plain_logprob=dist.log_prob(act).numpy())
        net.load_state_dict(after)
        steps.append(dict(v=v.numpy(), n_v=n_v.numpy(), new_logpi=new_logpi.numpy(), action=trainer.action_np.copy(), old_logpi=trainer.old_logpi_np.copy(),
                          reward=trainer.reward_np.copy(), not_terminated=trainer.not_terminated_np.copy(), loss_policy=np.float64(info["loss_policy"]),
                          loss_e=np.float64(info.get("loss_e", np.nan)), **extra))
        del rng_state
    out = {}
    for k, s in enumerate(steps):
        for key, val in s.items():
            out[f"s{k}_{key}"] = np.asarray(val)
    out.update(n_steps=np.int64(len(steps)), discount=np.float64(cfg.discount), clip_range=np.float64(clip_range), entropy_weight=np.float64(entropy_weight),
               continuous=np.bool_(continuous), squashed=np.bool_(sgp))
    np.savez_compressed(os.path.join(OUT, f"ppo_v_step_{name}.npz"), **out)
    print(name, "ok:", [float(s["loss_policy"]) for s in steps])


def main():
    record("discrete", "Grid", continuous=False, entropy_weight=0.1, clip_range=0.2, sgp=False)
    env = _continuous_env()
    record("continuous", env, continuous=True, entropy_weight=0.05, clip_range=0.1, sgp=False)


if __name__ == "__main__":
    main()
