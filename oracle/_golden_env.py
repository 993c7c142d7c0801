"""oracle/_golden_env.py -- TEST INFRASTRUCTURE ONLY.  A tiny image environment registered with the
imported reference from OUTSIDE its tree (srl/base/env/registration.py:116-136) to record golden vectors."""
import numpy as np
from srl.base.define import SpaceTypes
from srl.base.env.base import EnvBase
from srl.base.spaces.box import BoxSpace
from srl.base.spaces.discrete import DiscreteSpace


class TinyImageEnv(EnvBase):
    """uint8 frames presented as float32 u8/255 (image_processor.py:140-142), episodes of `ep_len`
    steps ending terminated (or truncated when `truncate`), rewards in {-2..2}."""

    def __init__(self, hw=8, actions=4, ep_len=6, truncate=False, seed=0, invalid=False):
        super().__init__()
        self.hw, self.na, self.ep_len, self.truncate, self.invalid = hw, actions, ep_len, truncate, invalid
        self.rng = np.random.default_rng(seed)
        self.t = 0
        self.log = []  # (frame_u8, action, reward, terminated, truncated); reset frames have action -1

    @property
    def action_space(self):
        return DiscreteSpace(self.na)

    @property
    def observation_space(self):
        return BoxSpace((self.hw, self.hw, 1), 0, 1, np.float32, SpaceTypes.GRAY_HW1)

    @property
    def max_episode_steps(self):
        return 1000

    @property
    def player_num(self):
        return 1

    def _frame(self):
        return self.rng.integers(0, 256, (self.hw, self.hw, 1), dtype=np.uint8)

    def reset(self, **kwargs):
        self.t = 0
        f = self._frame()
        self.log.append((f.copy(), -1, 0.0, False, False))
        return f.astype(np.float32) / 255

    def step(self, action):
        self.t += 1
        f = self._frame()
        r = float(self.rng.integers(-2, 3))
        end = self.t >= self.ep_len
        term = end and not self.truncate
        trunc = end and self.truncate
        self.log.append((f.copy(), int(action), r, term, trunc))
        return f.astype(np.float32) / 255, r, term, trunc

    def get_invalid_actions(self, player_index=-1):
        if self.invalid:
            return [int(self.t % self.na)]
        return []

    def backup(self, **kwargs):
        return None

    def restore(self, data, **kwargs):
        pass
