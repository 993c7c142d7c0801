"""base_ppo (srl/base/rl/algorithms/base_ppo.py:17-23): discrete OR continuous (flat float vector) actions, float ndarray
observations."""
from dataclasses import dataclass

from simple_distributed_rl_amd.base.define import RLBaseTypes
from simple_distributed_rl_amd.base.rl.config import RLConfig as _RLConfig
from simple_distributed_rl_amd.base.rl.worker import RLWorker  # noqa: F401


@dataclass
class RLConfig(_RLConfig):
    def get_base_action_type(self) -> RLBaseTypes:
        return RLBaseTypes.DISCRETE | RLBaseTypes.NP_ARRAY

    def get_base_observation_type(self) -> RLBaseTypes:
        return RLBaseTypes.NP_ARRAY
