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
import threading
import weakref
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from simple_distributed_rl_amd import _native as N

_DUELING = {"average": 0, "max": 1, "": 2}


class EngineQNet(nn.Module):
    def __init__(self, n_actions: int, hw=(84, 84), window: int = 4, hidden: int = 512, filters: int = 32, dueling_type: str = "average", noisy: bool = False,
                 uvfa_cols: int = 0):
        """noisy=True: the dense layers are NoisyLinear (srl/rl/torch_/modules/noisy_linear.py:8-52): `fc1`/`v2`/`a2` hold the mu
        tensors, `fc1_sigma_w` ... `a2_sigma_b` the sigmas, in the same (fused, NHWC-column) layouts.
        uvfa_cols = X > 0 (round 6): Agent57(_light)'s Q-network (agent57_light/model_torch.py:18-64) -- X further input columns behind the image features (previous
        rewards, one-hot previous action, one-hot actor).  They are kept apart as `fcx` [X][2*hidden] (column-major: what srlx_qnet_bind_uvfa reads); such a
        network is initialised by `load_reference_state_dict` (the reference's He-normal scale depends on the fan-in of the 7744 + X wide layer)."""
        super().__init__()
        self.hw, self.window, self.hidden, self.filters, self.n_actions, self.dueling_type = tuple(hw), window, hidden, filters, n_actions, dueling_type
        self.noisy = bool(noisy)
        self.uvfa_cols = int(uvfa_cols)
        assert not (self.noisy and self.uvfa_cols)
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
        if self.uvfa_cols:
            self.fcx = nn.Parameter(torch.zeros(self.uvfa_cols, 2 * hidden))
        if self.noisy:
            for name, lin in (("fc1", self.fc1), ("v2", self.v2), ("a2", self.a2)):
                setattr(self, name + "_sigma_w", nn.Parameter(torch.zeros_like(lin.weight)))
                setattr(self, name + "_sigma_b", nn.Parameter(torch.zeros_like(lin.bias)))
        # reference-equivalent initialisation: build the mirrored module and convert it
        from simple_distributed_rl_amd.rl.torch_.networks import atari_qnetwork

        self.weights_version = 0  # bumped by every state-dict load: caches derived from the weights (QNetInference's operand planes) compare it
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._bump_version())
        self.fix_formats()
        if not self.uvfa_cols:
            self.load_reference_state_dict(atari_qnetwork(n_actions, hw, window, hidden, self.noisy, filters, dueling_type).state_dict())

    def fix_formats(self):
        """conv2/conv3 weights in channels_last memory = [Cout][ky][kx][Cin], the K order of an NHWC implicit GEMM."""
        for conv in (self.conv2, self.conv3):
            conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def _apply(self, fn, *a, **k):  # .to(device) / .cuda(): keep the kernel layout
        out = super()._apply(fn, *a, **k)
        self.fix_formats()
        return out

    def forward(self, x, channels_first: bool = True, extras: torch.Tensor = None):
        """extras [B][uvfa_cols]: the dense UVFA inputs of a uvfa network (test yardstick; the kernels take them as scalars / indices)."""
        if not channels_first:
            x = x.permute(0, 3, 1, 2)
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        x = x.permute(0, 2, 3, 1).flatten(1)  # NHWC flatten: pixel-major, channel-minor
        if self.noisy:  # one draw per call, shared by every row (noisy_linear.py:35-52)
            def lin(name, inp):
                base = getattr(self, name)
                w = base.weight + getattr(self, name + "_sigma_w") * torch.randn_like(base.weight)
                b = base.bias + getattr(self, name + "_sigma_b") * torch.randn_like(base.bias)
                return F.linear(inp, w, b)

            h = F.relu(lin("fc1", x))
            v = lin("v2", h[:, : self.hidden])
            adv = lin("a2", h[:, self.hidden :])
        else:
            pre = self.fc1(x)
            if self.uvfa_cols:
                pre = pre + extras @ self.fcx
            h = F.relu(pre)
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

    def _fuse_fc1(self, v1, a1):
        """[H, C*P (+ X)] x 2 (columns c*P+p, then the UVFA columns) -> [2H, P*C] (NHWC columns); the UVFA columns go to `fcx` (transposed)"""
        C, P, H = self.out_c, self.out_p, self.hidden
        w = torch.cat([v1, a1], dim=0)
        if self.uvfa_cols:
            with torch.no_grad():
                self.fcx.copy_(w[:, C * P :].t())
            w = w[:, : C * P]
        return w.reshape(2 * H, C, P).permute(0, 2, 1).reshape(2 * H, P * C)

    def _split_fc1(self, w):
        C, P, H = self.out_c, self.out_p, self.hidden
        w = w.detach().view(2 * H, P, C).permute(0, 2, 1).reshape(2 * H, C * P)
        if self.uvfa_cols:
            w = torch.cat([w, self.fcx.detach().t()], dim=1)
        return w[:H].clone(), w[H:].clone()

    def _bump_version(self):
        self.weights_version += 1

    def load_reference_state_dict(self, sd):
        """The reference's keys and layouts (plain layers: `.weight` / `.bias`; NoisyLinear: `.w_mu` / `.w_sigma` / `.b_mu` / `.b_sigma`)."""
        self._bump_version()
        H = self.hidden
        wk, bk = ("w_mu", "b_mu") if self.noisy else ("weight", "bias")
        dev = self.fc1.weight.device
        with torch.no_grad():
            for mine, ref in self._CONV_KEYS.items():
                getattr(self, mine).weight.copy_(sd[ref + ".weight"])
                getattr(self, mine).bias.copy_(sd[ref + ".bias"])
            hd = self._HEAD
            self.fc1.weight.copy_(self._fuse_fc1(sd[f"{hd}.v_layers.0.{wk}"].to(dev), sd[f"{hd}.adv_layers.0.{wk}"].to(dev)))
            self.fc1.bias.copy_(torch.cat([sd[f"{hd}.v_layers.0.{bk}"], sd[f"{hd}.adv_layers.0.{bk}"]]))
            self.v2.weight.copy_(sd[f"{hd}.v_layers.2.{wk}"])
            self.v2.bias.copy_(sd[f"{hd}.v_layers.2.{bk}"])
            self.a2.weight.copy_(sd[f"{hd}.adv_layers.2.{wk}"])
            self.a2.bias.copy_(sd[f"{hd}.adv_layers.2.{bk}"])
            if self.noisy:
                self.fc1_sigma_w.copy_(self._fuse_fc1(sd[f"{hd}.v_layers.0.w_sigma"].to(dev), sd[f"{hd}.adv_layers.0.w_sigma"].to(dev)))
                self.fc1_sigma_b.copy_(torch.cat([sd[f"{hd}.v_layers.0.b_sigma"], sd[f"{hd}.adv_layers.0.b_sigma"]]))
                self.v2_sigma_w.copy_(sd[f"{hd}.v_layers.2.w_sigma"])
                self.v2_sigma_b.copy_(sd[f"{hd}.v_layers.2.b_sigma"])
                self.a2_sigma_w.copy_(sd[f"{hd}.adv_layers.2.w_sigma"])
                self.a2_sigma_b.copy_(sd[f"{hd}.adv_layers.2.b_sigma"])
        return self

    def reference_state_dict(self):
        H = self.hidden
        wk, bk = ("w_mu", "b_mu") if self.noisy else ("weight", "bias")
        sd = {}
        for mine, ref in self._CONV_KEYS.items():
            sd[ref + ".weight"] = getattr(self, mine).weight.detach().contiguous().clone()
            sd[ref + ".bias"] = getattr(self, mine).bias.detach().clone()
        hd = self._HEAD

        def put(layer, sub, w, b, sw=None, sb=None):
            sd[f"{hd}.{layer}.{sub}.{wk}"], sd[f"{hd}.{layer}.{sub}.{bk}"] = w, b
            if self.noisy:
                sd[f"{hd}.{layer}.{sub}.w_sigma"], sd[f"{hd}.{layer}.{sub}.b_sigma"] = sw, sb

        wv, wa = self._split_fc1(self.fc1.weight)
        bv, ba = self.fc1.bias.detach()[:H].clone(), self.fc1.bias.detach()[H:].clone()
        if self.noisy:
            sv, sa = self._split_fc1(self.fc1_sigma_w)
            sbv, sba = self.fc1_sigma_b.detach()[:H].clone(), self.fc1_sigma_b.detach()[H:].clone()
            put("v_layers", 0, wv, bv, sv, sbv)
            put("adv_layers", 0, wa, ba, sa, sba)
            put("v_layers", 2, self.v2.weight.detach().clone(), self.v2.bias.detach().clone(), self.v2_sigma_w.detach().clone(), self.v2_sigma_b.detach().clone())
            put("adv_layers", 2, self.a2.weight.detach().clone(), self.a2.bias.detach().clone(), self.a2_sigma_w.detach().clone(), self.a2_sigma_b.detach().clone())
        else:
            put("v_layers", 0, wv, bv)
            put("adv_layers", 0, wa, ba)
            put("v_layers", 2, self.v2.weight.detach().clone(), self.v2.bias.detach().clone())
            put("adv_layers", 2, self.a2.weight.detach().clone(), self.a2.bias.detach().clone())
        return sd

    def kernel_parameters(self):
        """The tensors libsrlx binds, in its order: 12 (conv1..a2, weight then bias; mu for noisy layers) + the 6 sigmas of a noisy net."""
        n = self
        ps = [n.conv1.weight, n.conv1.bias, n.conv2.weight, n.conv2.bias, n.conv3.weight, n.conv3.bias, n.fc1.weight, n.fc1.bias,
              n.v2.weight, n.v2.bias, n.a2.weight, n.a2.bias]
        if self.noisy:
            ps += [n.fc1_sigma_w, n.fc1_sigma_b, n.v2_sigma_w, n.v2_sigma_b, n.a2_sigma_w, n.a2_sigma_b]
        return ps


class EngineHiddenNet(nn.Module):
    """DQN image block + ONE dense layer with ReLU, in the kernels' layouts -- the trunk of Agent57_light's embedding network (`emb_block`,
    agent57_light/model_torch.py:76-78,96-97) and of its lifelong networks (`hidden_block`, :109-111); what follows (the embedding network's classifier, the
    lifelong networks' LayerNorm) are `tail` tensors in the reference's own layouts (csrc/srlx_agent57.hip).  The dense layer is padded with zero rows to the
    first-dense-layer GEMM's 128-unit tile (`units_padded`; the pad units have zero weight and bias, produce 0 and receive no gradient), and the handle
    (QNetInference + srlx_qnet_set_head_mode) needs a dueling head's tensors to exist: four zero dummies."""

    def __init__(self, units: int, hw=(84, 84), window: int = 4, filters: int = 32, tail_shapes=()):
        super().__init__()
        self.units = int(units)
        self.units_padded = -(-self.units // 128) * 128
        self.hw, self.window, self.filters = tuple(hw), window, filters
        self.hidden, self.n_actions, self.dueling_type, self.noisy, self.uvfa_cols = self.units_padded // 2, 1, "average", False, 0
        Fi = filters
        self.conv1 = nn.Conv2d(window, Fi, 8, 4, padding=3, padding_mode="replicate")
        self.conv2 = nn.Conv2d(Fi, 2 * Fi, 4, 2, padding=2, padding_mode="replicate")
        self.conv3 = nn.Conv2d(2 * Fi, 2 * Fi, 3, 1, padding=1, padding_mode="replicate")
        with torch.no_grad():
            y = self.conv3(self.conv2(self.conv1(torch.zeros(1, window, hw[0], hw[1]))))
        self.out_c, self.out_p = y.shape[1], y.shape[2] * y.shape[3]
        self.flat = self.out_c * self.out_p
        self.fc1 = nn.Linear(self.flat, self.units_padded)
        self.v2 = nn.Linear(self.hidden, 1)
        self.a2 = nn.Linear(self.hidden, 1)
        self.tail = nn.ParameterList([nn.Parameter(torch.zeros(sh)) for sh in tail_shapes])
        with torch.no_grad():
            for p in (self.fc1.weight, self.fc1.bias, self.v2.weight, self.v2.bias, self.a2.weight, self.a2.bias):
                p.zero_()
        self.weights_version = 0
        for conv in (self.conv2, self.conv3):
            conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        for conv in (self.conv2, self.conv3):
            conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
        return out

    _CONV_KEYS = EngineQNet._CONV_KEYS

    def load_reference(self, sd, dense_key: str, tail_keys=()):
        """sd: the reference network's state_dict; dense_key e.g. "emb_block.hidden_layers.0"; tail_keys: the tail tensors' keys in `tail` order."""
        self.weights_version += 1
        C, P, U = self.out_c, self.out_p, self.units
        with torch.no_grad():
            for mine, ref in self._CONV_KEYS.items():
                getattr(self, mine).weight.copy_(sd[ref + ".weight"])
                getattr(self, mine).bias.copy_(sd[ref + ".bias"])
            w = sd[dense_key + ".weight"].to(self.fc1.weight.device)
            self.fc1.weight.zero_()
            self.fc1.bias.zero_()
            self.fc1.weight[:U].copy_(w.reshape(U, C, P).permute(0, 2, 1).reshape(U, P * C))  # channel-major columns -> NHWC
            self.fc1.bias[:U].copy_(sd[dense_key + ".bias"])
            for t, k in zip(self.tail, tail_keys):
                t.copy_(sd[k])
        return self

    def reference_tensors(self, dense_key: str, tail_keys=()):
        C, P, U = self.out_c, self.out_p, self.units
        sd = {}
        for mine, ref in self._CONV_KEYS.items():
            sd[ref + ".weight"] = getattr(self, mine).weight.detach().contiguous().clone()
            sd[ref + ".bias"] = getattr(self, mine).bias.detach().clone()
        sd[dense_key + ".weight"] = self.fc1.weight.detach()[:U].reshape(U, P, C).permute(0, 2, 1).reshape(U, C * P).clone()
        sd[dense_key + ".bias"] = self.fc1.bias.detach()[:U].clone()
        for t, k in zip(self.tail, tail_keys):
            sd[k] = t.detach().clone()
        return sd

    def kernel_parameters(self):
        n = self
        return [n.conv1.weight, n.conv1.bias, n.conv2.weight, n.conv2.bias, n.conv3.weight, n.conv3.bias, n.fc1.weight, n.fc1.bias, n.v2.weight, n.v2.bias,
                n.a2.weight, n.a2.bias]

    def forward(self, x):
        """float32 channels-first stack -> the dense layer's post-ReLU units [B][units] (test yardstick)."""
        x = F.relu(self.conv3(F.relu(self.conv2(F.relu(self.conv1(x))))))
        return F.relu(self.fc1(x.permute(0, 2, 3, 1).flatten(1)))[:, : self.units]


_LIVE_HANDLES = weakref.WeakSet()  # every wrapper of a srlx_qnet handle (check_ranges)


def check_ranges():
    """Raises if a fused convolution pass of ANY live handle met an activation outside float16's range (the two-part split of round 6 cannot hold it: that pass's
    values are meaningless).  Reads one device word per handle: called where the host synchronises anyway (the engines' info())."""
    me = threading.get_ident()  # (handles of the calling thread only: a rank-per-thread job's other threads create and destroy theirs concurrently)
    for w in list(_LIVE_HANDLES):
        if getattr(w, "h", None) and getattr(w, "_owner_thread", None) == me:
            bits = ctypes.c_int(0)
            N.check(w.lib.srlx_qnet_range_flags(w.h, ctypes.byref(bits)))
            if bits.value:
                layers = " and ".join(f"conv{l + 1}" for l in range(3) if bits.value >> l & 1)
                raise RuntimeError(f"srlx: an activation of {layers} exceeded 65504, the range of the two-part float16 split the fused convolution kernel and the first dense "
                                   "layer evaluate their float32 products with -- the outputs since then are not the network's.  Run the process with SRLX_CONV_BF16X3=1 "
                                   "(convolutions on the three-part bf16 split, float32's range) and, if conv3 is named, SRLX_FC1_F32=1 SRLX_NO_CONV_PLANES=1 (first dense layer "
                                   "on the float32 matrix pipe).")


class QNetInference:
    """Matrix-core forward over the live parameters of an EngineQNet (zero copy)."""

    def __init__(self, net: EngineQNet, max_batch: int, device: int = 0, noise_seed: int = 0, uvfa_layout=None):
        self.lib = N.lib()
        self.net = net
        self._uvfa_layout = uvfa_layout
        self.noise_seed = int(noise_seed)
        self.window, self.n_actions = net.window, net.n_actions
        self.max_batch = int(max_batch)
        self.dev = torch.device(f"cuda:{device}")
        hh = N.c_p()
        N.check(
            self.lib.srlx_qnet_create(ctypes.byref(hh), net.hw[0], net.hw[1], net.window, net.filters, net.hidden, net.n_actions,
                                      _DUELING[net.dueling_type], self.max_batch, int(device))
        )
        self.h = hh
        self._owner_thread = threading.get_ident()
        _LIVE_HANDLES.add(self)
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
        """(Re)reads the parameters' addresses (again after they have been re-homed: device/dist.py:flatten_parameters)."""
        n = self.net
        params = n.kernel_parameters()
        for conv in (n.conv2, n.conv3):
            assert conv.weight.is_contiguous(memory_format=torch.channels_last), "EngineQNet.fix_formats() was undone"
        for p in params:
            assert p.is_cuda and p.dtype == torch.float32
        arr = (N.c_p * 12)(*[p.data_ptr() for p in params[:12]])
        N.check(self.lib.srlx_qnet_bind(self.h, ctypes.cast(arr, N.c_p)))
        if n.noisy:  # every forward from now on materialises mu + sigma * eps of a fresh draw first
            sig = (N.c_p * 6)(*[p.data_ptr() for p in params[12:]])
            N.check(self.lib.srlx_qnet_bind_noisy(self.h, ctypes.cast(sig, N.c_p), self.noise_seed))
        self._bound = [p.data_ptr() for p in params]
        if getattr(n, "uvfa_cols", 0):
            assert self._uvfa_layout is not None, "a UVFA network: QNetInference(..., uvfa_layout=(col_ext, col_int, col_action, n_action_in, col_actor, n_actor))"
            ce, ci, ca, na, ck, nk = self._uvfa_layout
            N.check(self.lib.srlx_qnet_bind_uvfa(self.h, N.tptr(n.fcx), n.uvfa_cols, ce, ci, ca, na, ck, nk))

    def _params(self):
        ps = list(self.net.kernel_parameters())
        if getattr(self.net, "uvfa_cols", 0):
            ps.append(self.net.fcx)
        return ps

    # ---- round 6: Agent57(_light)'s networks (srlx.h: srlx_qnet_bind_uvfa ..) -------------------------------------------------------------------------------
    def set_uvfa_inputs(self, r_ext, r_int, action, actor):
        """Per-row UVFA inputs of the next passes: float32 rewards, int32 indices (device tensors the handle keeps reading; None where the input is absent)."""
        self._uvfa_in = (r_ext, r_int, action, actor)
        for t, dt in zip(self._uvfa_in, (torch.float32, torch.float32, torch.int32, torch.int32)):
            assert t is None or (t.dtype == dt and t.is_contiguous())
        N.check(self.lib.srlx_qnet_set_uvfa_inputs(self.h, N.tptr(r_ext), N.tptr(r_int), N.tptr(action), N.tptr(actor)))

    def set_td_extras(self, discount_per_sample, td_signed):
        self._td_extras = (discount_per_sample, td_signed)
        N.check(self.lib.srlx_qnet_set_td_extras(self.h, N.tptr(discount_per_sample), N.tptr(td_signed)))

    def set_head_mode(self, mode: int, out_cols: int = 0, ln_w=None, ln_b=None, ln_eps: float = 1e-5):
        self._ln = (ln_w, ln_b)
        N.check(self.lib.srlx_qnet_set_head_mode(self.h, int(mode), int(out_cols), N.tptr(ln_w), N.tptr(ln_b), float(ln_eps)))

    def grad_params(self):
        """The 12 parameters in the order of the gradient list handed to the backward entry points."""
        return list(self._params())[:12]

    def enable_training(self, max_train_batch: int):
        """Allocates the backward scratch and static gradient tensors (`p.grad`, in each parameter's own memory format,
        so that the fused Adam reads what `backward_u8` writes and both can live in one HIP graph)."""
        N.check(self.lib.srlx_qnet_enable_training(self.h, int(max_train_batch)))
        for p in self._params():
            p.grad = torch.zeros_like(p)  # preserve_format: conv2/conv3 stay channels_last
        self._grads = [p.grad for p in self._params()]
        self._grad_arr = (N.c_p * 12)(*[g.data_ptr() for g in self._grads[:12]])
        if getattr(self.net, "uvfa_cols", 0):  # the columns' gradient buffer (their optimiser step: DeviceAdam.fuse_rest)
            N.check(self.lib.srlx_qnet_fuse_adam_uvfa(self.h, N.tptr(self.net.fcx.grad), None, None))
        if self.net.noisy:
            self._sig_grad_arr = (N.c_p * 6)(*[g.data_ptr() for g in self._grads[12:]])
            N.check(self.lib.srlx_qnet_bind_noisy_grads(self.h, ctypes.cast(self._sig_grad_arr, N.c_p)))
        return self

    def redraw_rows(self, rows: int, row_stride: int, out: torch.Tensor = None) -> torch.Tensor:
        """NoisyLinear: the dense layers of rows 0, stride, ... of the last forward again, under a fresh noise draw (the draw the
        backward pass then differentiates); overwrites their rows of `out` / the handle's q buffer."""
        q = self.q if out is None else out
        N.check(self.lib.srlx_qnet_redraw_rows(self.h, int(rows), int(row_stride), N.tptr(q), N.torch_stream_ptr()))
        return q

    def effective(self, which: int):
        """(copy of an effective tensor = mu + sigma * eps of the current draw, draw id): 0 fc1 weight, 1 fc1 bias, 2 v2 w, 3 v2 b, 4 a2 w, 5 a2 b."""
        n, draw = N.c_i64(), N.c_i64()
        N.check(self.lib.srlx_qnet_noisy_effective(self.h, int(which), None, ctypes.byref(n), None, N.torch_stream_ptr()))
        out = torch.empty(n.value, dtype=torch.float32, device=self.dev)
        N.check(self.lib.srlx_qnet_noisy_effective(self.h, int(which), N.tptr(out), None, ctypes.byref(draw), N.torch_stream_ptr()))
        return out, draw.value

    def backward_u8(self, frame_base_ptr: int, frame_off: torch.Tensor, grad_q: torch.Tensor, sample_stride: int = 1):
        """Parameter gradients of sum(q * grad_q) for the samples at rows 0, stride, 2*stride, ... of the last forward_u8
        (for a noisy net: of the draw the dense layers of those rows were last evaluated under; sigma gradients included)."""
        B = grad_q.shape[0]
        assert grad_q.is_contiguous() and (hasattr(self.net, "units") or grad_q.shape[1] == self.n_actions)  # (a hidden-layer handle takes d loss / d its output)
        N.check(self.lib.srlx_qnet_backward_u8(self.h, B, int(sample_stride), N.c_p(frame_base_ptr), N.tptr(frame_off), N.tptr(grad_q),
                                               ctypes.cast(self._grad_arr, N.c_p), N.torch_stream_ptr()))

    def backward_td_u8(self, frame_base_ptr: int, frame_off: torch.Tensor, n_step: int, q_on_all, q_tg_next, actions, rewards, terminated, weights, discount: float,
                       retrace_h: float, double_dqn: bool, rescale: bool, target, loss, grad_q0, priorities):
        """`srlx_nstep_td_huber_priority_packed` + `backward_u8(..., sample_stride=n_step + 1)` as one call: the backward's head kernel computes
        the TD target / Huber loss / gradient seed / priorities of the B items in its prologue (srlx_qnet_backward_td_u8)."""
        B = grad_q0.shape[0]
        N.check(self.lib.srlx_qnet_backward_td_u8(self.h, B, int(n_step), N.c_p(frame_base_ptr), N.tptr(frame_off), N.tptr(q_on_all), N.tptr(q_tg_next), N.tptr(actions),
                                                  N.tptr(rewards), N.tptr(terminated), None, N.tptr(weights), float(discount), float(retrace_h), int(double_dqn),
                                                  int(rescale), N.tptr(target), N.tptr(loss), N.tptr(grad_q0), N.tptr(priorities), ctypes.cast(self._grad_arr, N.c_p),
                                                  N.torch_stream_ptr()))

    def set_probe(self, ev_start: torch.cuda.Event, ev_end: torch.cuda.Event):
        """The next forward records the two (timing-enabled, already created) events right around its convolution kernel launch(es)."""
        N.check(self.lib.srlx_qnet_set_probe(self.h, N.c_p(ev_start.cuda_event), N.c_p(ev_end.cuda_event)))

    def set_fc1_span(self, span: torch.Tensor):
        """int64 [2] device tensor pre-set to (-1, 0): the next operand-planes first-dense-layer launch leaves its own first-in / last-out wall-clock stamps there
        (100 MHz ticks; srlx_qnet_set_fc1_span)."""
        assert span.dtype == torch.int64 and span.numel() == 2 and span.is_contiguous()
        N.check(self.lib.srlx_qnet_set_fc1_span(self.h, N.tptr(span)))

    def set_probe_fc1(self, ev_start: torch.cuda.Event, ev_end: torch.cuda.Event):
        """The same around the first dense layer's GEMM launch."""
        N.check(self.lib.srlx_qnet_set_probe_fc1(self.h, N.c_p(ev_start.cuda_event), N.c_p(ev_end.cuda_event)))

    # ---- first dense layer on pre-split operand planes (srlx_fc1_planes.hip): for the chip-filling launches of an actor handle ----------------
    def enable_fc1_planes(self, private_weights: bool):
        """Keep the first dense layer's weight also as three bf16 part planes and run chip-filling forwards (>= 512 rows, multiples of 128) on them,
        conversion-free and bit-identical.  `private_weights=True`: nobody but `refresh_from` / `weights_changed` changes this network's weights
        (the engine's private actor copy), so the planes stay valid between those calls; False: the weights are shared with a learner, every
        chip-filling forward re-splits them first (one 80 MB pass)."""
        assert not self.net.noisy
        N.check(self.lib.srlx_qnet_enable_fc1_planes(self.h))
        self._planes, self._planes_private, self._planes_stale, self._planes_version = True, bool(private_weights), True, -1

    def weights_changed(self):
        """The bound float32 weights were written by somebody else (load_state_dict, a broadcast): planes and packed filters derived from them are stale."""
        N.check(self.lib.srlx_qnet_weights_changed(self.h))
        if getattr(self, "_planes", False):
            self._planes_stale = True
            N.check(self.lib.srlx_qnet_invalidate_fc1_planes(self.h))

    # ---- round 4: published parameter sets (an actor handle) / packed filters that outlive a forward -------------------------------------------
    def enable_actor_sets(self):
        """Two device-resident parameter sets for the policy pass (packed convolution filters, first dense layer as bf16 operand planes, small vectors): the learner
        writes one while the actors read the other (`publish_to`, the fused Adam's plane epilogue), `select_set` flips."""
        assert not self.net.noisy
        N.check(self.lib.srlx_qnet_actor_sets_enable(self.h))
        self._set = -1

    def set_planes_ptr(self, k: int) -> int:
        p = N.c_p()
        N.check(self.lib.srlx_qnet_actor_set_planes(self.h, int(k), ctypes.byref(p)))
        return p.value

    def select_set(self, k: int):
        N.check(self.lib.srlx_qnet_actor_set_select(self.h, int(k)))
        self._set = int(k)

    def publish_to(self, actor: "QNetInference" = None, k: int = 0, with_fc1: bool = False, bump: torch.Tensor = None):
        """Pack this handle's convolution filters for its own next forwards and (with `actor`) publish the network into that handle's set k in the same launch;
        with_fc1: also split the first dense layer's weight into the set's planes (out-of-band publishes: start-up, restore)."""
        N.check(self.lib.srlx_qnet_publish(self.h, actor.h if actor is not None else None, int(k), int(bool(with_fc1)), N.tptr(bump), N.torch_stream_ptr()))

    def set_priority_sink(self, replay, indices: Optional[torch.Tensor], priorities: Optional[torch.Tensor]):
        """The replay's priority write-back (model_torch.py:113-114) as the first launch of the backward pass's weight-gradient branch (srlx_qnet_set_priority_sink);
        replay = None removes it.  float32 |td| priorities, transformed on the device like DeviceReplay.update does."""
        if replay is None:
            N.check(self.lib.srlx_qnet_set_priority_sink(self.h, None, 0, None, None, 0))
        else:
            N.check(self.lib.srlx_qnet_set_priority_sink(self.h, replay.h_per, indices.numel(), N.tptr(indices), N.tptr(priorities), N.PRIO_F32))

    def set_sink_wait(self, ev: Optional[torch.cuda.Event]):
        """The priority sink's write-back waits for `ev` first (recorded at least once already: torch creates the HIP event lazily); None: no wait."""
        self._sink_wait = ev  # keeps it alive
        N.check(self.lib.srlx_qnet_set_sink_wait(self.h, N.c_p(ev.cuda_event) if ev is not None else None))

    def set_sink_done(self, ev: Optional[torch.cuda.Event]):
        """`ev` (recorded at least once already) is recorded on the sink's branch right behind every write-back from now on; None: off."""
        self._sink_done = ev
        N.check(self.lib.srlx_qnet_set_sink_done(self.h, N.c_p(ev.cuda_event) if ev is not None else None))

    def set_sink_stream(self, stream: Optional[torch.cuda.Stream]):
        """The write-back runs on `stream` (which the caller joins: `set_sink_done`) instead of first on the weight-gradient branch; None: back there."""
        self._sink_stream = stream
        N.check(self.lib.srlx_qnet_set_sink_stream(self.h, N.c_p(stream.cuda_stream) if stream is not None else None))

    def set_td_event(self, ev: torch.cuda.Event):
        """`ev` (already recorded once: torch creates the HIP event lazily) is recorded right behind the head kernel of every backward pass from now on."""
        self._td_event = ev  # keeps it alive
        N.check(self.lib.srlx_qnet_set_td_event(self.h, N.c_p(ev.cuda_event) if ev is not None else None))

    def fuse_adam_planes(self, planes_ptr):
        """The fused first-dense-layer Adam of the next backward passes also writes the updated weight as operand planes at `planes_ptr` (None: off)."""
        N.check(self.lib.srlx_qnet_fuse_adam_fc1_planes(self.h, N.c_p(planes_ptr) if planes_ptr else None))

    def set_fc1_neighbour(self, splits: int):
        """Chip-filling first-dense-layer launches on operand planes as half-CU workgroups with `splits` K splits (0: CU-filling workgroups, the fastest form alone)."""
        N.check(self.lib.srlx_qnet_set_fc1_neighbour(self.h, int(splits)))

    def set_planes_small(self, on: bool, weight_planes_ptr=None):
        """A learner's handle: operand planes also for its 96 / 128-row passes; `weight_planes_ptr`: borrowed planes of the bound weight for the next forwards
        (None: the handle's own, `refresh_own_planes`)."""
        N.check(self.lib.srlx_qnet_set_planes_small(self.h, int(bool(on)), N.c_p(weight_planes_ptr) if weight_planes_ptr else None))

    def refresh_own_planes(self):
        """The bound first-dense-layer weight split into the handle's own operand planes (one pass over the weight, current stream)."""
        N.check(self.lib.srlx_qnet_refresh_fc1_planes(self.h, None, None, N.torch_stream_ptr()))

    def set_pack_sticky(self, on: bool = True):
        N.check(self.lib.srlx_qnet_set_pack_sticky(self.h, int(bool(on))))

    def forward_u8_policy(self, frame_base_ptr: int, frame_off: torch.Tensor, eps: torch.Tensor, seed: int, counter: torch.Tensor, actions: torch.Tensor,
                          invalid: torch.Tensor = None, q_copy: torch.Tensor = None, out: torch.Tensor = None) -> torch.Tensor:
        """`forward_u8` whose head kernel also selects the actions (epsilon-greedy with the keyed uniforms of (seed, counter): what srlx_rng_uniform +
        srlx_policy_epsilon_greedy would pick); the counter is only read."""
        B = frame_off.numel() // self.window
        q = self.q[:B] if out is None else out
        self._planes_ready(B)
        N.check(self.lib.srlx_qnet_forward_u8_policy(self.h, B, N.c_p(frame_base_ptr), N.tptr(frame_off), N.tptr(q), N.tptr(eps), int(seed) & 0xFFFFFFFFFFFFFFFF, N.tptr(counter),
                                                     N.tptr(invalid), N.tptr(actions), N.tptr(q_copy), N.torch_stream_ptr()))
        return q

    def refresh_from(self, online: "EngineQNet"):
        """This network := `online` (all parameter tensors), the first dense layer's weight copied AND split into planes in one pass."""
        mine, theirs = self.net.kernel_parameters(), online.kernel_parameters()
        if getattr(self, "_planes", False):
            k = [i for i, p in enumerate(mine) if p is self.net.fc1.weight][0]
            with torch.no_grad():
                torch._foreach_copy_([p for i, p in enumerate(mine) if i != k], [p for i, p in enumerate(theirs) if i != k])
            N.check(self.lib.srlx_qnet_refresh_fc1_planes(self.h, N.tptr(theirs[k]), N.tptr(mine[k]), N.torch_stream_ptr()))
            self._planes_stale, self._planes_version = False, self.net.weights_version
        else:
            with torch.no_grad():
                torch._foreach_copy_(mine, theirs)

    def _planes_ready(self, B: int):
        if not getattr(self, "_planes", False) or B < 512 or B % 128 or getattr(self, "_set", -1) >= 0:
            return
        if self._planes_stale or not self._planes_private or self._planes_version != self.net.weights_version:
            N.check(self.lib.srlx_qnet_refresh_fc1_planes(self.h, None, None, N.torch_stream_ptr()))
            self._planes_stale, self._planes_version = False, self.net.weights_version

    @staticmethod
    def forward_convs_multi(handles, frame_base_ptr: int, frame_off: torch.Tensor):
        """The image blocks of several inference handles over the SAME frame stacks as one launch (srlx_qnet_forward_convs_multi_u8); `forward_dense` continues each."""
        h0 = handles[0]
        B = frame_off.numel() // h0.window
        for h in handles:
            h._planes_ready(B)
        arr = (N.c_p * len(handles))(*[h.h for h in handles])
        N.check(h0.lib.srlx_qnet_forward_convs_multi_u8(ctypes.cast(arr, N.c_p), len(handles), B, N.c_p(frame_base_ptr), N.tptr(frame_off), N.torch_stream_ptr()))
        return B

    def forward_dense(self, B: int, out: torch.Tensor = None) -> torch.Tensor:
        """The dense layers on the operand planes the last `forward_convs_multi` left for this handle."""
        q = self.q[:B] if out is None else out
        N.check(self.lib.srlx_qnet_forward_dense_planes(self.h, int(B), N.tptr(q), N.torch_stream_ptr()))
        return q

    def forward_f32(self, obs_nchw: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        B = obs_nchw.shape[0]
        q = self.q[:B] if out is None else out
        self._planes_ready(B)
        N.check(self.lib.srlx_qnet_forward_f32(self.h, B, N.tptr(obs_nchw), N.tptr(q), N.torch_stream_ptr()))
        return q

    def forward_u8(self, frame_base_ptr: int, frame_off: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        B = frame_off.numel() // self.window
        q = self.q[:B] if out is None else out
        self._planes_ready(B)
        N.check(self.lib.srlx_qnet_forward_u8(self.h, B, N.c_p(frame_base_ptr), N.tptr(frame_off), N.tptr(q), N.torch_stream_ptr()))
        return q


class ImageTrunk:
    """The image block of a torch network (DQNImageBlock, dqn_image_block.py:29-54: 8x8/4, 4x4/2, 3x3/1 convolutions with replicate padding
    and ReLU) evaluated by libsrlx straight from the uint8 frame ring -- for networks whose dense part is NOT the dueling head of
    `EngineQNet` (Agent57_light's UVFA Q-networks, embedding network and RND networks).  The module's own parameters are bound by address
    (conv2 / conv3 are switched to channels_last memory, which torch's convolution keeps accepting: autograd training of the same module
    goes on unchanged), so the features always reflect the current weights.  `__call__` returns what `image_block(x).flatten(1)` returns
    on the float32 stack of the same frames (pixel values / 255), to float32 round-off (tests/test_agent57_engine_gpu.py)."""

    @staticmethod
    def supported(image_block) -> bool:
        layers = list(getattr(image_block, "image_layers", []))
        if len(layers) != 6 or not all(isinstance(m, nn.Conv2d) for m in layers[0::2]) or not all(isinstance(m, nn.ReLU) for m in layers[1::2]):
            return False
        c1, c2, c3 = layers[0::2]
        geo = [(c.kernel_size, c.stride, c.padding, c.padding_mode) for c in (c1, c2, c3)]
        want = [((8, 8), (4, 4), (3, 3), "replicate"), ((4, 4), (2, 2), (2, 2), "replicate"), ((3, 3), (1, 1), (1, 1), "replicate")]
        f = c1.out_channels
        return geo == want and f in (32, 64, 128) and c2.out_channels == 2 * f and c3.out_channels == 2 * f and c3.in_channels == 2 * f and c1.bias is not None

    def __init__(self, image_block, hw, max_batch: int, device: int = 0):
        assert ImageTrunk.supported(image_block)
        self.lib = N.lib()
        self.convs = list(image_block.image_layers)[0::2]
        c1 = self.convs[0]
        self.max_batch = int(max_batch)
        self.dev = torch.device(f"cuda:{device}")
        for conv in self.convs[1:]:
            conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
        hh = N.c_p()
        N.check(self.lib.srlx_qnet_create(ctypes.byref(hh), int(hw[0]), int(hw[1]), c1.in_channels, c1.out_channels, 32, 1, 0, self.max_batch, int(device)))
        self.h = hh
        self._owner_thread = threading.get_ident()
        _LIVE_HANDLES.add(self)
        self._unused = torch.zeros(64, dtype=torch.float32, device=self.dev)  # the dense-layer entries of srlx_qnet_bind (never read by forward_convs)
        with torch.no_grad():
            y = image_block(torch.zeros((1, c1.in_channels, int(hw[0]), int(hw[1])), device=self.dev))
        self.channels, self.pixels = y.shape[1], y.shape[2] * y.shape[3]
        self.out = torch.empty((self.max_batch, self.pixels, self.channels), dtype=torch.float32, device=self.dev)
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
        ps = [t for c in self.convs for t in (c.weight, c.bias)]
        assert all(p.is_cuda and p.dtype == torch.float32 for p in ps)
        arr = (N.c_p * 12)(*([p.data_ptr() for p in ps] + [self._unused.data_ptr()] * 6))
        N.check(self.lib.srlx_qnet_bind(self.h, ctypes.cast(arr, N.c_p)))
        self._bound = [p.data_ptr() for p in ps]

    def __call__(self, frame_base_ptr: int, frame_off: torch.Tensor) -> torch.Tensor:
        B = frame_off.shape[0]
        ps = [t for c in self.convs for t in (c.weight, c.bias)]
        if [p.data_ptr() for p in ps] != self._bound:  # the parameters were re-homed (load_state_dict keeps them, .to() / flattening does not)
            self.bind()
        for conv in self.convs[1:]:
            assert conv.weight.is_contiguous(memory_format=torch.channels_last), "ImageTrunk: a convolution weight left channels_last memory"
        N.check(self.lib.srlx_qnet_forward_convs_u8(self.h, B, N.c_p(frame_base_ptr), N.tptr(frame_off), N.tptr(self.out), N.torch_stream_ptr()))
        return self.out[:B].transpose(1, 2).reshape(B, self.channels * self.pixels)  # pixel-major -> torch's channel-major flatten


class _TrunkFunction(torch.autograd.Function):
    """features = image_block(frames) with the forward AND the backward in libsrlx.  The six convolution parameters are inputs only so that autograd
    routes their gradients through `backward`; the kernels read them by address."""

    @staticmethod
    def forward(ctx, trunk, frame_base_ptr, frame_off, grad_stride, w1, b1, w2, b2, w3, b3):
        ctx.trunk, ctx.base, ctx.off, ctx.stride = trunk, frame_base_ptr, frame_off, int(grad_stride)
        return ImageTrunk.__call__(trunk, frame_base_ptr, frame_off)

    @staticmethod
    def backward(ctx, g):
        t = ctx.trunk
        R = g.shape[0]
        ss = ctx.stride
        rows = R // ss
        # rows 0, ss, 2 ss, ... carry gradient (the others were used without one); channel-major flatten -> the kernels' pixel-major rows
        gp = g.view(R, t.channels, t.pixels)[0::ss].transpose(1, 2).contiguous()
        arr = (N.c_p * 6)(*[b.data_ptr() for b in t.grad_bufs])
        N.check(t.lib.srlx_qnet_backward_convs_u8(t.h, rows, ss, N.c_p(ctx.base), N.tptr(ctx.off), N.tptr(gp), ctypes.cast(arr, N.c_p), N.torch_stream_ptr()))
        return (None, None, None, None) + tuple(b.clone() for b in t.grad_bufs)


class TrainableImageTrunk(ImageTrunk):
    """An `ImageTrunk` whose features carry gradients: `features(base, off, grad_stride)` is differentiable with respect to the module's convolution
    parameters, forward and backward both hand-written (srlx_qnet_forward_convs_u8 on a training-enabled handle, srlx_qnet_backward_convs_u8) -- the
    image blocks of Agent57_light's learner (model_torch.py:18-117) without MIOpen: the convolutions' weight and data gradients use fixed summation
    orders, so an update is reproducible run to run (MIOpen picks a solver per instance by timing, and its fp32 Winograd kernels round differently).
    `grad_stride` = s: only rows 0, s, 2s, ... of the batch receive gradient (the others -- next states evaluated under no_grad -- share the forward)."""

    def __init__(self, image_block, hw, max_batch: int, device: int = 0, max_grad_rows: int = 64):
        super().__init__(image_block, hw, max_batch, device)
        assert self.convs[0].out_channels == 32, "the backward kernels cover the 32 / 64 / 64-filter block"
        N.check(self.lib.srlx_qnet_enable_training(self.h, int(max_grad_rows)))
        ps = [t for c in self.convs for t in (c.weight, c.bias)]
        self.grad_bufs = [torch.zeros_like(p) for p in ps]  # each in its parameter's own memory format (channels_last conv2 / conv3)

    def features(self, frame_base_ptr: int, frame_off: torch.Tensor, grad_stride: int = 1) -> torch.Tensor:
        ps = [t for c in self.convs for t in (c.weight, c.bias)]
        return _TrunkFunction.apply(self, frame_base_ptr, frame_off, grad_stride, *ps)


class DeviceAdam:
    """torch.optim.Adam(params, lr) for a fixed list of float32 device tensors as ONE libsrlx launch
    (`srlx_adam_step`; reference: `optim.Adam(self.q_online.parameters(), lr=...)`, model_torch.py:71, and
    `optimizer.step()`, :109).  Reads `p.grad`, keeps `exp_avg` / `exp_avg_sq` in each parameter's own memory format,
    takes the step count from a device scalar so the call replays inside a HIP graph."""

    def __init__(self, params, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = [p for p in params]
        assert 0 < len(self.params) <= 24 and all(p.is_cuda and p.dtype == torch.float32 for p in self.params)
        assert all(p.grad is not None for p in self.params), "DeviceAdam needs static gradient tensors (QNetInference.enable_training)"
        self.lib = N.lib()
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]  # preserve_format
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        for p, m in zip(self.params, self.exp_avg):
            assert p.stride() == m.stride() == p.grad.stride(), "parameter, gradient and Adam state must share one memory format"
        self._fused = None  # index of the tensor the network's backward pass updates itself (fuse_first_dense)
        self._rest = False  # fuse_rest: every other tensor too
        self._tables()

    def _tables(self):
        self._idx = [i for i in range(len(self.params)) if i != self._fused]
        k = len(self._idx)
        self._g = (N.c_p * k)(*[self.params[i].grad.data_ptr() for i in self._idx])
        self._m = (N.c_p * k)(*[self.exp_avg[i].data_ptr() for i in self._idx])
        self._v = (N.c_p * k)(*[self.exp_avg_sq[i].data_ptr() for i in self._idx])
        self._n = (N.c_i64 * k)(*[self.params[i].numel() for i in self._idx])
        self.bind()

    def bind(self):
        """(Re)reads the parameters' addresses: call again after they have been re-homed (device/dist.py:flatten_parameters)."""
        self._p = (N.c_p * len(self._idx))(*[self.params[i].data_ptr() for i in self._idx])

    def fuse_first_dense(self, inf: "QNetInference", steps_taken_dev: torch.Tensor):
        """The first dense layer's weight (97 % of the parameters) gets its Adam step inside `inf.backward_u8`'s weight-gradient kernel
        (srlx_qnet_fuse_adam_fc1): its gradient is never written to memory and `step()` covers the remaining tensors only.
        `steps_taken_dev` must be the device scalar later handed to `step()`.  Not available for NoisyLinear networks."""
        assert self._fused is None and not inf.net.noisy
        k = [i for i, p in enumerate(self.params) if p is inf.net.fc1.weight]
        assert len(k) == 1, "the optimiser does not hold this network's first dense layer"
        self._fused = k[0]
        self._fused_steps = steps_taken_dev  # keeps the scalar alive: the library holds its address
        N.check(self.lib.srlx_qnet_fuse_adam_fc1(inf.h, N.tptr(self.exp_avg[self._fused]), N.tptr(self.exp_avg_sq[self._fused]), self.lr, self.betas[0], self.betas[1],
                                                 self.eps, N.tptr(steps_taken_dev)))
        self._tables()

    def fuse_rest(self, inf: "QNetInference"):
        """After `fuse_first_dense`: every other tensor takes its step inside the launch that finishes its gradient (srlx_qnet_fuse_adam_rest: the convolution
        tensors in their gradient reductions, the small vectors in the packing launch of `inf.publish_to`) -- `step()` then launches nothing, and every
        `inf.backward*_u8` must be followed by `inf.publish_to(...)`.  Only for a handle whose parameter order is the gradient list's (QNetInference.grad_list)."""
        assert self._fused is not None and not self._rest
        order = inf.grad_params()  # the 12 parameters in the gradient list's order
        pos = [next(i for i, q in enumerate(self.params) if q is p_) for p_ in order]
        self._rest_tables = ((N.c_p * 12)(*[self.params[i].grad.data_ptr() for i in pos]), (N.c_p * 12)(*[self.exp_avg[i].data_ptr() for i in pos]),
                             (N.c_p * 12)(*[self.exp_avg_sq[i].data_ptr() for i in pos]))
        N.check(self.lib.srlx_qnet_fuse_adam_rest(inf.h, *[ctypes.cast(t, N.c_p) for t in self._rest_tables]))
        if getattr(inf.net, "uvfa_cols", 0):  # the UVFA columns step in the packing launch too
            k = next(i for i, q in enumerate(self.params) if q is inf.net.fcx)
            N.check(self.lib.srlx_qnet_fuse_adam_uvfa(inf.h, N.tptr(inf.net.fcx.grad), N.tptr(self.exp_avg[k]), N.tptr(self.exp_avg_sq[k])))
        self._rest = True

    def step(self, steps_taken_dev: torch.Tensor):
        """One Adam step; `steps_taken_dev` (int64 device scalar) = steps already taken (the caller increments it)."""
        assert self._fused is None or steps_taken_dev.data_ptr() == self._fused_steps.data_ptr()
        if self._rest:  # nothing left for a launch of its own
            return
        k = len(self._idx)
        N.check(self.lib.srlx_adam_step(k, ctypes.cast(self._p, N.c_p), ctypes.cast(self._g, N.c_p), ctypes.cast(self._m, N.c_p), ctypes.cast(self._v, N.c_p),
                                        ctypes.cast(self._n, N.c_p), self.lr, self.betas[0], self.betas[1], self.eps, N.tptr(steps_taken_dev), N.torch_stream_ptr()))

    def state_dict(self):
        return {"exp_avg": [t.clone() for t in self.exp_avg], "exp_avg_sq": [t.clone() for t in self.exp_avg_sq]}

    def load_state_dict(self, sd):
        for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd["exp_avg"]) + list(sd["exp_avg_sq"])):
            dst.copy_(src)
