"""The engine's Q-network: one set of parameters shared by autograd/Adam (torch) and the libsrlx matrix-core
inference kernels (`srlx_qnet_*`).

`EngineQNet` computes exactly what the reference's QNetwork does for the Atari shape (DQN image block +
one dueling head: srl/rl/torch_/blocks/dqn_image_block.py:10-67, dueling_network.py:8-59,
rainbow/model_torch.py:15-29) but stores its parameters in the layout the kernels read in place:
conv2/conv3 weights in channels_last memory, the V and A first layers fused into one
[2*hidden, flat] matrix whose columns follow the NHWC flatten order.  `load_reference_state_dict` /
`reference_state_dict` convert from/to the reference's state_dict keys and layouts, so checkpoints stay
interchangeable.

`QNetInference` binds those parameters (zero copy) and runs every no-grad forward of the vectorised engine:
the actor's policy step and the learner's online/target evaluation of s_1..s_n, reading uint8 frames
straight from the ring.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from simple_distributed_rl_amd import _native as N

_DUELING = {"average": 0, "max": 1, "": 2}


class EngineQNet(nn.Module):
    def __init__(self, n_actions: int, hw=(84, 84), window: int = 4, hidden: int = 512, filters: int = 32, dueling_type: str = "average"):
        super().__init__()
        self.hw, self.window, self.hidden, self.filters, self.n_actions, self.dueling_type = tuple(hw), window, hidden, filters, n_actions, dueling_type
        Fi = filters
        self.conv1 = nn.Conv2d(window, Fi, 8, 4, padding=3, padding_mode="replicate")
        self.conv2 = nn.Conv2d(Fi, 2 * Fi, 4, 2, padding=2, padding_mode="replicate")
        self.conv3 = nn.Conv2d(2 * Fi, 2 * Fi, 3, 1, padding=1, padding_mode="replicate")
        with torch.no_grad():
            y = self.conv3(self.conv2(self.conv1(torch.zeros(1, window, hw[0], hw[1]))))
        self.out_c, self.out_p = y.shape[1], y.shape[2] * y.shape[3]
        self.flat = self.out_c * self.out_p
        self.fc1 = nn.Linear(self.flat, 2 * hidden)
        self.v2 = nn.Linear(hidden, 1)
        self.a2 = nn.Linear(hidden, n_actions)
        # reference-equivalent initialisation: build the mirrored module and convert it
        from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

        self.fix_formats()
        self.load_reference_state_dict(atari_qnetwork(n_actions, hw, window, hidden, False, filters).state_dict())

    def fix_formats(self):
        """conv2/conv3 weights in channels_last memory = [Cout][ky][kx][Cin], the K order of an NHWC implicit GEMM."""
        for conv in (self.conv2, self.conv3):
            conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def _apply(self, fn, *a, **k):  # .to(device) / .cuda(): keep the kernel layout
        out = super()._apply(fn, *a, **k)
        self.fix_formats()
        return out

    def forward(self, x, channels_first: bool = True):
        if not channels_first:
            x = x.permute(0, 3, 1, 2)
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        x = x.permute(0, 2, 3, 1).flatten(1)  # NHWC flatten: pixel-major, channel-minor
        h = F.relu(self.fc1(x))
        v = self.v2(h[:, : self.hidden])
        adv = self.a2(h[:, self.hidden :])
        if self.dueling_type == "average":
            return v + adv - adv.mean(dim=-1, keepdim=True)
        if self.dueling_type == "max":
            return v + adv - adv.max(dim=-1, keepdim=True)[0]
        return v + adv

    # ---- reference <-> engine layouts -----------------------------------------------------------
    _CONV_KEYS = {"conv1": "in_block.image_block.image_layers.0", "conv2": "in_block.image_block.image_layers.2", "conv3": "in_block.image_block.image_layers.4"}
    _HEAD = "hidden_block.hidden_layers.0"

    def load_reference_state_dict(self, sd):
        C, P, H = self.out_c, self.out_p, self.hidden
        with torch.no_grad():
            for mine, ref in self._CONV_KEYS.items():
                getattr(self, mine).weight.copy_(sd[ref + ".weight"])
                getattr(self, mine).bias.copy_(sd[ref + ".bias"])
            v1, a1 = sd[self._HEAD + ".v_layers.0.weight"], sd[self._HEAD + ".adv_layers.0.weight"]
            w = torch.cat([v1, a1], dim=0).to(self.fc1.weight.device)  # [2H, C*P] columns c*P+p
            self.fc1.weight.copy_(w.view(2 * H, C, P).permute(0, 2, 1).reshape(2 * H, P * C))
            self.fc1.bias.copy_(torch.cat([sd[self._HEAD + ".v_layers.0.bias"], sd[self._HEAD + ".adv_layers.0.bias"]]))
            self.v2.weight.copy_(sd[self._HEAD + ".v_layers.2.weight"])
            self.v2.bias.copy_(sd[self._HEAD + ".v_layers.2.bias"])
            self.a2.weight.copy_(sd[self._HEAD + ".adv_layers.2.weight"])
            self.a2.bias.copy_(sd[self._HEAD + ".adv_layers.2.bias"])
        return self

    def reference_state_dict(self):
        C, P, H = self.out_c, self.out_p, self.hidden
        sd = {}
        for mine, ref in self._CONV_KEYS.items():
            sd[ref + ".weight"] = getattr(self, mine).weight.detach().contiguous().clone()
            sd[ref + ".bias"] = getattr(self, mine).bias.detach().clone()
        w = self.fc1.weight.detach().view(2 * H, P, C).permute(0, 2, 1).reshape(2 * H, C * P)
        sd[self._HEAD + ".v_layers.0.weight"], sd[self._HEAD + ".adv_layers.0.weight"] = w[:H].clone(), w[H:].clone()
        sd[self._HEAD + ".v_layers.0.bias"], sd[self._HEAD + ".adv_layers.0.bias"] = self.fc1.bias.detach()[:H].clone(), self.fc1.bias.detach()[H:].clone()
        sd[self._HEAD + ".v_layers.2.weight"], sd[self._HEAD + ".v_layers.2.bias"] = self.v2.weight.detach().clone(), self.v2.bias.detach().clone()
        sd[self._HEAD + ".adv_layers.2.weight"], sd[self._HEAD + ".adv_layers.2.bias"] = self.a2.weight.detach().clone(), self.a2.bias.detach().clone()
        return sd


class QNetInference:
    """Matrix-core forward over the live parameters of an EngineQNet (zero copy)."""

    def __init__(self, net: EngineQNet, max_batch: int, device: int = 0):
        self.lib = N.lib()
        self.net = net
        self.window, self.n_actions = net.window, net.n_actions
        self.max_batch = int(max_batch)
        self.dev = torch.device(f"cuda:{device}")
        hh = N.c_p()
        N.check(
            self.lib.srlx_qnet_create(ctypes.byref(hh), net.hw[0], net.hw[1], net.window, net.filters, net.hidden, net.n_actions,
                                      _DUELING[net.dueling_type], self.max_batch, int(device))
        )
        self.h = hh
        self.q = torch.zeros((self.max_batch, self.n_actions), dtype=torch.float32, device=self.dev)
        self.bind()

    def __del__(self):
        if getattr(self, "h", None):
            try:
                torch.cuda.synchronize(self.dev)
            except Exception:
                pass
            self.lib.srlx_qnet_destroy(self.h)
            self.h = None

    def bind(self):
        n = self.net
        params = [n.conv1.weight, n.conv1.bias, n.conv2.weight, n.conv2.bias, n.conv3.weight, n.conv3.bias, n.fc1.weight, n.fc1.bias,
                  n.v2.weight, n.v2.bias, n.a2.weight, n.a2.bias]
        for conv in (n.conv2, n.conv3):
            assert conv.weight.is_contiguous(memory_format=torch.channels_last), "EngineQNet.fix_formats() was undone"
        for p in params:
            assert p.is_cuda and p.dtype == torch.float32
        arr = (N.c_p * 12)(*[p.data_ptr() for p in params])
        N.check(self.lib.srlx_qnet_bind(self.h, ctypes.cast(arr, N.c_p)))
        self._bound = [p.data_ptr() for p in params]

    def _params(self):
        n = self.net
        return [n.conv1.weight, n.conv1.bias, n.conv2.weight, n.conv2.bias, n.conv3.weight, n.conv3.bias, n.fc1.weight, n.fc1.bias,
                n.v2.weight, n.v2.bias, n.a2.weight, n.a2.bias]

    def enable_training(self, max_train_batch: int):
        """Allocates the backward scratch and static gradient tensors (`p.grad`, in each parameter's own memory format,
        so that the fused Adam reads what `backward_u8` writes and both can live in one HIP graph)."""
        N.check(self.lib.srlx_qnet_enable_training(self.h, int(max_train_batch)))
        for p in self._params():
            p.grad = torch.zeros_like(p)  # preserve_format: conv2/conv3 stay channels_last
        self._grads = [p.grad for p in self._params()]
        self._grad_arr = (N.c_p * 12)(*[g.data_ptr() for g in self._grads])
        return self

    def backward_u8(self, frame_base_ptr: int, frame_off: torch.Tensor, grad_q: torch.Tensor, sample_stride: int = 1):
        """Parameter gradients of sum(q * grad_q) for the samples at rows 0, stride, 2*stride, ... of the last forward_u8."""
        B = grad_q.shape[0]
        assert grad_q.is_contiguous() and grad_q.shape[1] == self.n_actions
        N.check(self.lib.srlx_qnet_backward_u8(self.h, B, int(sample_stride), N.c_p(frame_base_ptr), N.tptr(frame_off), N.tptr(grad_q),
                                               ctypes.cast(self._grad_arr, N.c_p), N.torch_stream_ptr()))

    def set_probe(self, ev_start: torch.cuda.Event, ev_end: torch.cuda.Event):
        """The next forward records the two (timing-enabled, already created) events around its two conv GEMM launches."""
        N.check(self.lib.srlx_qnet_set_probe(self.h, N.c_p(ev_start.cuda_event), N.c_p(ev_end.cuda_event)))

    def forward_f32(self, obs_nchw: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        B = obs_nchw.shape[0]
        q = self.q[:B] if out is None else out
        N.check(self.lib.srlx_qnet_forward_f32(self.h, B, N.tptr(obs_nchw), N.tptr(q), N.torch_stream_ptr()))
        return q

    def forward_u8(self, frame_base_ptr: int, frame_off: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        B = frame_off.numel() // self.window
        q = self.q[:B] if out is None else out
        N.check(self.lib.srlx_qnet_forward_u8(self.h, B, N.c_p(frame_base_ptr), N.tptr(frame_off), N.tptr(q), N.torch_stream_ptr()))
        return q


class DeviceAdam:
    """torch.optim.Adam(params, lr) for a fixed list of float32 device tensors as ONE libsrlx launch
    (`srlx_adam_step`; reference: `optim.Adam(self.q_online.parameters(), lr=...)`, model_torch.py:71, and
    `optimizer.step()`, :109).  Reads `p.grad`, keeps `exp_avg` / `exp_avg_sq` in each parameter's own memory format,
    takes the step count from a device scalar so the call replays inside a HIP graph."""

    def __init__(self, params, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = [p for p in params]
        assert 0 < len(self.params) <= 16 and all(p.is_cuda and p.dtype == torch.float32 for p in self.params)
        assert all(p.grad is not None for p in self.params), "DeviceAdam needs static gradient tensors (QNetInference.enable_training)"
        self.lib = N.lib()
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]  # preserve_format
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        k = len(self.params)
        self.bind()
        self._g = (N.c_p * k)(*[p.grad.data_ptr() for p in self.params])
        self._m = (N.c_p * k)(*[t.data_ptr() for t in self.exp_avg])
        self._v = (N.c_p * k)(*[t.data_ptr() for t in self.exp_avg_sq])
        self._n = (N.c_i64 * k)(*[p.numel() for p in self.params])
        for p, m in zip(self.params, self.exp_avg):
            assert p.stride() == m.stride() == p.grad.stride(), "parameter, gradient and Adam state must share one memory format"

    def bind(self):
        """(Re)reads the parameters' addresses: call again after they have been re-homed (device/dist.py:flatten_parameters)."""
        self._p = (N.c_p * len(self.params))(*[p.data_ptr() for p in self.params])

    def step(self, steps_taken_dev: torch.Tensor):
        """One Adam step; `steps_taken_dev` (int64 device scalar) = steps already taken (the caller increments it)."""
        k = len(self.params)
        N.check(self.lib.srlx_adam_step(k, ctypes.cast(self._p, N.c_p), ctypes.cast(self._g, N.c_p), ctypes.cast(self._m, N.c_p), ctypes.cast(self._v, N.c_p),
                                        ctypes.cast(self._n, N.c_p), self.lr, self.betas[0], self.betas[1], self.eps, N.tptr(steps_taken_dev), N.torch_stream_ptr()))

    def state_dict(self):
        return {"exp_avg": [t.clone() for t in self.exp_avg], "exp_avg_sq": [t.clone() for t in self.exp_avg_sq]}

    def load_state_dict(self, sd):
        for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd["exp_avg"]) + list(sd["exp_avg_sq"])):
            dst.copy_(src)
