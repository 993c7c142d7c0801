"""base_ql (srl/base/rl/algorithms/base_ql.py): discrete action, ARRAY_DISCRETE observation (tabular)."""
from dataclasses import dataclass

from simple_distributed_rl_amd.base.define import RLBaseTypes
from simple_distributed_rl_amd.base.rl.config import RLConfig as _RLConfig
from simple_distributed_rl_amd.base.rl.worker import RLWorker  # noqa: F401


@dataclass
class RLConfig(_RLConfig):
    def get_base_action_type(self) -> RLBaseTypes:
        return RLBaseTypes.DISCRETE

    def get_base_observation_type(self) -> RLBaseTypes:
        return RLBaseTypes.ARRAY_DISCRETE

    def get_framework(self) -> str:
        return ""
