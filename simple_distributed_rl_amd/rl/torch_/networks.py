"""Q-network building blocks with the reference's module tree (so state_dicts are interchangeable).

Mirrors, for the hot path only:
  srl/rl/torch_/blocks/dqn_image_block.py:10-67      DQNImageBlock
  srl/rl/torch_/blocks/input_image_block.py:43-82    InputImageBlock (reshape + image block + flatten)
  srl/rl/torch_/blocks/input_value_block.py:11-63    InputValueBlock
  srl/rl/torch_/blocks/mlp_block.py:9-49             MLPBlock
  srl/rl/torch_/blocks/dueling_network.py:8-59       DuelingNetworkBlock
  srl/rl/torch_/modules/noisy_linear.py:8-52         NoisyLinear
  srl/algorithms/rainbow/model_torch.py:15-29        QNetwork

The dense layers are the only MFMA-shaped work on the path; they run through PyTorch-ROCm
(MIOpen / hipBLASLt), everything around them is libsrlx.  Inputs may be given channels-first
(N, C, H, W) -- what the device frame ring produces -- or in the reference's (N, H, W, C).
"""
import math
from typing import Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

_ACT = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "elu": nn.ELU, "leakyrelu": nn.LeakyReLU, "gelu": nn.GELU, "silu": nn.SiLU}


def convert_activation(name):
    if not isinstance(name, str):
        return name
    key = name.lower().replace("_", "")
    if key not in _ACT:
        raise ValueError(f"unknown activation: {name}")
    return _ACT[key]


def apply_initializer(x: torch.Tensor, initializer: str):
    """srl/rl/torch_/converter.py:21-48"""
    import torch.nn.init as init

    with torch.no_grad():
        ini = initializer.lower()
        if ini == "he_normal":
            return init.kaiming_normal_(x, mode="fan_in", nonlinearity="relu")
        if ini == "glorot_uniform":
            return init.xavier_uniform_(x)
        table = {n.lower().replace("_", ""): n for n in dir(init) if not n.startswith("_")}
        if ini in table:
            return getattr(init, table[ini])(x)
    raise ValueError(f"Unknown initializer: {initializer}")


class NoisyLinear(nn.Module):
    """Independent Gaussian noise per weight, resampled every forward (noisy_linear.py:26-52)."""

    def __init__(self, in_features: int, out_features: int, sigma: float = 0.5):
        super().__init__()
        self.in_features, self.out_features, self.sigma = in_features, out_features, sigma
        self.w_mu = nn.Parameter(torch.empty(out_features, in_features))
        self.w_sigma = nn.Parameter(torch.empty(out_features, in_features))
        self.b_mu = nn.Parameter(torch.empty(out_features))
        self.b_sigma = nn.Parameter(torch.empty(out_features))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.w_mu.size(1))
        self.w_mu.data.uniform_(-stdv, stdv)
        self.b_mu.data.uniform_(-stdv, stdv)
        self.w_sigma.data.fill_(self.sigma * stdv)
        self.b_sigma.data.fill_(self.sigma * stdv)

    def forward(self, x):
        w_noise = torch.randn(self.w_mu.size(), dtype=self.w_mu.dtype, device=self.b_mu.device)
        b_noise = torch.randn(self.b_mu.size(), dtype=self.b_mu.dtype, device=self.b_mu.device)
        return F.linear(x, self.w_mu + self.w_sigma * w_noise, self.b_mu + self.b_sigma * b_noise)


class DQNImageBlock(nn.Module):
    def __init__(self, in_shape: Tuple[int, ...], filters: int = 32, activation="ReLU"):
        super().__init__()
        act = convert_activation(activation)
        in_ch = in_shape[-3]
        self.image_layers = nn.ModuleList(
            [
                nn.Conv2d(in_ch, filters, kernel_size=8, stride=4, padding=3, padding_mode="replicate"),
                act(),
                nn.Conv2d(filters, filters * 2, kernel_size=4, stride=2, padding=2, padding_mode="replicate"),
                act(),
                nn.Conv2d(filters * 2, filters * 2, kernel_size=3, stride=1, padding=1, padding_mode="replicate"),
                act(),
            ]
        )
        with torch.no_grad():
            y = self.forward(torch.ones((1,) + tuple(in_shape), dtype=torch.float32))
        self.out_shape = tuple(y.shape[-3:])

    def forward(self, x):
        for layer in self.image_layers:
            x = layer(x)
        return x


class InputImageBlock(nn.Module):
    """Image input: (N,H,W,C) -> permute -> image block -> flatten (input_image_block.py:43-82 with the
    IMAGE_MAP branch of input_image_reshape_block.py:55-60).  `channels_first=True` skips the permute:
    the device frame ring already emits (N, C, H, W)."""

    def __init__(self, hwc_shape: Tuple[int, int, int], filters: int = 32, activation="ReLU", out_flatten: bool = True):
        super().__init__()
        h, w, c = hwc_shape
        self.in_shape = tuple(hwc_shape)
        self.out_flatten = out_flatten
        self.image_block = DQNImageBlock((c, h, w), filters, activation)
        if out_flatten:
            self.img_flat = nn.Flatten()
            self.out_size = int(torch.zeros(self.image_block.out_shape).numel())
        else:
            self.out_shape = self.image_block.out_shape

    def forward(self, x: torch.Tensor, channels_first: bool = False):
        if not channels_first:
            x = x.permute((0, 3, 1, 2))
        x = self.image_block(x)
        if self.out_flatten:
            x = self.img_flat(x)
        return x


class InputValueBlock(nn.Module):
    def __init__(self, in_shape: Tuple[int, ...], layer_sizes: Sequence[int] = (), activation="ReLU", use_bias=True,
                 kernel_initializer="he_normal", bias_initializer="zeros", enable_noisy_dense=False, input_flatten=True):
        super().__init__()
        act = convert_activation(activation)
        self.hidden_layers = nn.ModuleList()
        if input_flatten:
            self.hidden_layers.append(nn.Flatten())
            in_size = 1
            for s in in_shape:
                in_size *= int(s)
        else:
            in_size = in_shape[-1]
        for size in layer_sizes:
            self.hidden_layers.append(_dense(in_size, size, use_bias, kernel_initializer, bias_initializer, enable_noisy_dense))
            self.hidden_layers.append(act())
            in_size = size
        self.out_size = in_size

    def forward(self, x, channels_first: bool = False):
        for layer in self.hidden_layers:
            x = layer(x)
        return x


def _dense(in_size, size, use_bias, kernel_initializer, bias_initializer, noisy):
    if noisy:
        return NoisyLinear(in_size, size)
    layer = nn.Linear(in_size, size, bias=use_bias)
    if kernel_initializer != "":
        apply_initializer(layer.weight, kernel_initializer)
    if use_bias and bias_initializer != "":
        apply_initializer(layer.bias, bias_initializer)
    return layer


class MLPBlock(nn.Module):
    def __init__(self, in_size: int, layer_sizes: Sequence[int] = (512,), activation="ReLU", use_bias=True,
                 kernel_initializer="he_normal", bias_initializer="zeros", enable_noisy_dense=False):
        super().__init__()
        act = convert_activation(activation)
        self.hidden_layers = nn.ModuleList()
        for size in layer_sizes:
            self.hidden_layers.append(_dense(in_size, size, use_bias, kernel_initializer, bias_initializer, enable_noisy_dense))
            self.hidden_layers.append(act())
            in_size = size
        self.out_size = in_size

    def add_layer(self, layer, out_size):
        self.hidden_layers.append(layer)
        self.out_size = out_size

    def forward(self, x):
        for layer in self.hidden_layers:
            x = layer(x)
        return x


class DuelingNetworkBlock(nn.Module):
    def __init__(self, in_size: int, hidden_units: int, out_layer_units: int, dueling_type: str = "average",
                 activation="ReLU", enable_noisy_dense: bool = False):
        super().__init__()
        self.dueling_type = dueling_type
        act = convert_activation(activation)
        lin = NoisyLinear if enable_noisy_dense else nn.Linear
        self.v_layers = nn.ModuleList([lin(in_size, hidden_units), act(), lin(hidden_units, 1)])
        self.adv_layers = nn.ModuleList([lin(in_size, hidden_units), act(), lin(hidden_units, out_layer_units)])

    def forward(self, x):
        v = x
        for layer in self.v_layers:
            v = layer(v)
        adv = x
        for layer in self.adv_layers:
            adv = layer(adv)
        if self.dueling_type == "average":
            return v + adv - torch.mean(adv, dim=-1, keepdim=True)
        if self.dueling_type == "max":
            return v + adv - torch.max(adv, dim=-1, keepdim=True)[0]
        if self.dueling_type == "":
            return v + adv
        raise ValueError("dueling_network_type is undefined")


def create_dueling_hidden_block(in_size: int, out_size: int, layer_sizes: Sequence[int] = (512,), dueling_type="average",
                                activation="ReLU", enable_noisy_dense=False, mlp_kwargs=None):
    """DuelingNetworkConfig.create_torch_block, "DuelingNetwork" branch (dueling_network.py:129-152):
    MLP over layer_sizes[:-1], then a dueling head with layer_sizes[-1] hidden units."""
    layer_sizes = tuple(layer_sizes)
    block = MLPBlock(in_size, layer_sizes[:-1], enable_noisy_dense=enable_noisy_dense, **(mlp_kwargs or {}))
    block.add_layer(DuelingNetworkBlock(block.out_size, layer_sizes[-1], out_size, dueling_type, activation, enable_noisy_dense), out_size)
    return block


def create_mlp_hidden_block(in_size: int, out_size: int, layer_sizes: Sequence[int] = (512,), enable_noisy_dense=False, **kw):
    """"MLP" branch (dueling_network.py:120-127): MLP + plain Linear head."""
    block = MLPBlock(in_size, tuple(layer_sizes), enable_noisy_dense=enable_noisy_dense, **kw)
    block.add_layer(nn.Linear(block.out_size, out_size), out_size)
    return block


class QNetwork(nn.Module):
    """in_block -> hidden_block (rainbow/model_torch.py:15-29, dqn/model_torch.py:17-29)."""

    def __init__(self, in_block: nn.Module, hidden_block: nn.Module, out_layer: nn.Module = None):
        super().__init__()
        self.in_block = in_block
        self.hidden_block = hidden_block
        if out_layer is not None:  # DQN: a separate head module named `out_layer` (dqn/model_torch.py:24)
            self.out_layer = out_layer

    def forward(self, x, channels_first: bool = False):
        x = self.hidden_block(self.in_block(x, channels_first=channels_first))
        return self.out_layer(x) if hasattr(self, "out_layer") else x


def atari_qnetwork(n_actions: int, hw=(84, 84), window: int = 4, hidden: int = 512, enable_noisy_dense: bool = False, filters: int = 32,
                   dueling_type: str = "average"):
    """The network of rainbow.Config.set_atari_config (rainbow.py:116-148): DQN image block + dueling (512,) average."""
    in_block = InputImageBlock((hw[0], hw[1], window), filters=filters)
    return QNetwork(in_block, create_dueling_hidden_block(in_block.out_size, n_actions, (hidden,), dueling_type, "ReLU", enable_noisy_dense))
