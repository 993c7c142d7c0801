"""Model-block configuration objects with the reference's builder names
(srl/rl/models/config/{input_block,hidden_block,dueling_network,framework_config}.py), creating the torch
blocks of simple_distributed_rl_amd.rl.torch_.networks."""
from dataclasses import dataclass, field
from typing import Tuple

from simple_distributed_rl_amd.base.exception import UndefinedError


@dataclass
class RLConfigComponentFramework:
    """framework_config.py: only torch exists on this target (single-backend rule)."""

    framework: str = "torch"

    def set_torch(self):
        self.framework = "torch"
        return self

    def set_tensorflow(self):
        raise UndefinedError("TensorFlow is not available on the MI355X target; the build is torch(ROCm)-only")

    def get_framework(self) -> str:
        return "torch"


@dataclass
class InputValueBlockConfig:
    name: str = "MLP"
    kwargs: dict = field(default_factory=lambda: dict(layer_sizes=()))

    def set(self, layer_sizes: Tuple[int, ...] = (), activation: str = "relu", **kw):
        self.name = "MLP"
        self.kwargs = dict(layer_sizes=tuple(layer_sizes), activation=activation, **kw)
        return self


@dataclass
class InputImageBlockConfig:
    name: str = "DQN"
    kwargs: dict = field(default_factory=lambda: dict(filters=32, activation="relu"))

    def set_dqn_block(self, filters: int = 32, activation: str = "relu"):
        self.name = "DQN"
        self.kwargs = dict(filters=filters, activation=activation)
        return self


@dataclass
class InputBlockConfig:
    value: InputValueBlockConfig = field(default_factory=InputValueBlockConfig)
    image: InputImageBlockConfig = field(default_factory=InputImageBlockConfig)

    def create_torch_block(self, cfg):
        from simple_distributed_rl_amd.rl.torch_ import networks as nw

        space = cfg.observation_space
        if space.is_image_like():
            if self.image.name != "DQN":
                raise UndefinedError(self.image.name)
            shape = space.shape
            if len(shape) == 2:
                shape = (shape[0], shape[1], 1)
            return nw.InputImageBlock(tuple(shape), **self.image.kwargs)
        return nw.InputValueBlock(space.shape, input_flatten=True, **self.value.kwargs)


@dataclass
class HiddenBlockConfig:
    """hidden_block.py:8-69: MLP hidden layers + a plain Linear head."""

    name: str = "MLP"
    kwargs: dict = field(default_factory=lambda: dict(layer_sizes=(512,)))

    def set(self, layer_sizes: Tuple[int, ...] = (512,), activation: str = "relu", **kw):
        self.name = "MLP"
        self.kwargs = dict(layer_sizes=tuple(layer_sizes), activation=activation, **kw)
        return self

    def create_torch_block(self, in_size: int, out_size: int = None, enable_noisy_dense: bool = False):
        """out_size=None: the bare MLP of hidden_block.py:57-61 (embedding / RND trunks); otherwise MLP + Linear head."""
        from simple_distributed_rl_amd.rl.torch_ import networks as nw

        if out_size is None:
            return nw.MLPBlock(in_size, enable_noisy_dense=enable_noisy_dense, **self.kwargs)
        return nw.create_mlp_hidden_block(in_size, out_size, enable_noisy_dense=enable_noisy_dense, **self.kwargs)


@dataclass
class DuelingNetworkConfig:
    """dueling_network.py:8-161."""

    name: str = "DuelingNetwork"
    kwargs: dict = field(default_factory=lambda: dict(layer_sizes=(512,), mlp_kwargs={}, dueling_kwargs=dict(dueling_type="average", activation="relu")))

    def set(self, layer_sizes: Tuple[int, ...] = (512,), activation: str = "relu", **kw):
        self.name = "MLP"
        self.kwargs = dict(layer_sizes=tuple(layer_sizes), activation=activation, **kw)
        return self

    def set_dueling_network(self, layer_sizes: Tuple[int, ...] = (512,), activation: str = "relu", dueling_type: str = "average", **mlp_kwargs):
        self.name = "DuelingNetwork"
        self.kwargs = dict(layer_sizes=tuple(layer_sizes), mlp_kwargs=dict(activation=activation, **mlp_kwargs),
                           dueling_kwargs=dict(dueling_type=dueling_type, activation=activation))
        return self

    def create_torch_block(self, in_size: int, out_size: int, enable_noisy_dense: bool = False):
        from simple_distributed_rl_amd.rl.torch_ import networks as nw

        if self.name == "MLP":
            return nw.create_mlp_hidden_block(in_size, out_size, enable_noisy_dense=enable_noisy_dense, **self.kwargs)
        if self.name == "DuelingNetwork":
            dk = self.kwargs["dueling_kwargs"]
            return nw.create_dueling_hidden_block(in_size, out_size, self.kwargs["layer_sizes"], dk.get("dueling_type", "average"),
                                                  dk.get("activation", "relu"), enable_noisy_dense, self.kwargs.get("mlp_kwargs"))
        raise UndefinedError(self.name)
