""""SyntheticAtari-v0": the synthetic 84x84 workload of BASELINE.json configs[2] as a registered environment.

Frames are i.i.d. uint8 U{0..255} handed out as float32 u8/255 in (H, W, 1) GRAY_HW1 -- the output contract of the
reference's ImageProcessor "0to1" mode (srl/rl/processors/image_processor.py:104-151) --, rewards i.i.d. from
{-1, 0, 1}, A discrete actions, episodes of `episode_len` steps that end `terminated`.

Two faces:
  * the ordinary host environment (EnvBase) below, so every plugin path (`Runner.evaluate`, actor processes of the
    multiprocessing topology, CPU tests) can play it one step at a time;
  * `device_vector(replay, ...)`: the same workload as E device-resident lanes (`SyntheticAtariVecEnv`, libsrlx
    `srlx_synth_env_step`), which is what `Runner.train()` uses on a GPU: frames are produced in HBM and never
    cross PCIe.  The two faces draw from different generators (numpy PCG64 here, the keyed counter RNG on the device);
    they are the same distribution, not the same stream.
"""
from typing import Optional

import numpy as np

from simple_distributed_rl_amd.base.define import SpaceTypes
from simple_distributed_rl_amd.base.env.base import EnvBase
from simple_distributed_rl_amd.base.env.registration import register
from simple_distributed_rl_amd.base.spaces.box import BoxSpace
from simple_distributed_rl_amd.base.spaces.discrete import DiscreteSpace

register("SyntheticAtari-v0", __name__ + ":SyntheticAtari", check_duplicate=False)


class SyntheticAtari(EnvBase):
    def __init__(self, hw=(84, 84), n_actions: int = 6, episode_len: int = 200, seed: int = 0):
        super().__init__()
        self.hw = (int(hw[0]), int(hw[1]))
        self.n_actions, self.episode_len = int(n_actions), int(episode_len)
        self._rng = np.random.Generator(np.random.PCG64(seed))
        self._t = 0

    @property
    def action_space(self):
        return DiscreteSpace(self.n_actions)

    @property
    def observation_space(self):
        return BoxSpace(self.hw + (1,), 0, 1, np.float32, SpaceTypes.GRAY_HW1)

    @property
    def max_episode_steps(self) -> int:
        return self.episode_len + 1

    @property
    def player_num(self) -> int:
        return 1

    def _frame(self) -> np.ndarray:
        return self._rng.integers(0, 256, self.hw + (1,), dtype=np.uint8).astype(np.float32) / np.float32(255)

    def reset(self, *, seed: Optional[int] = None, **kwargs):
        if seed is not None:
            self._rng = np.random.Generator(np.random.PCG64(seed))
        self._t = 0
        return self._frame()

    def step(self, action):
        self._t += 1
        reward = float(self._rng.integers(-1, 2))
        return self._frame(), reward, self._t >= self.episode_len, False

    def backup(self, **kwargs):
        return (self._t, self._rng.bit_generator.state)

    def restore(self, data, **kwargs):
        self._t, self._rng.bit_generator.state = data

    @classmethod
    def device_vector(cls, replay, hw=(84, 84), n_actions: int = 6, episode_len: int = 200, seed: int = 0):
        """E = replay.E device-resident lanes of this workload."""
        from simple_distributed_rl_amd.device.rainbow import SyntheticAtariVecEnv

        assert int(hw[0]) * int(hw[1]) == replay.F and int(n_actions) == replay.A
        return SyntheticAtariVecEnv(replay, int(episode_len))
