"""Device-side arithmetic shared by the DQN / Rainbow plugin trainers: thin wrappers that hand torch
tensors to libsrlx.  A trainer on a non-GPU device fails loudly -- there is no CPU fallback."""
import numpy as np
import torch

from simple_distributed_rl_amd import _native as N


def require_gpu(device_str: str) -> torch.device:
    if not str(device_str).startswith("cuda") or not torch.cuda.is_available():
        raise RuntimeError(
            f"simple_distributed_rl_amd: this trainer runs its TD-target / loss / priority arithmetic in libsrlx HIP kernels and "
            f"needs an MI355X (device was '{device_str}'). There is no CPU fallback; use the reference for CPU-only runs."
        )
    N.lib()
    return torch.device(device_str)


def invalid_mask(invalid_lists, shape, device) -> torch.Tensor:
    """nested lists of invalid action ids -> uint8 mask of `shape` (..., A), or None when all empty."""
    m = np.zeros(shape, np.uint8)
    any_ = False
    flat = m.reshape(-1, shape[-1])
    for i, inv in enumerate(invalid_lists):
        for a in inv:
            flat[i, a] = 1
            any_ = True
    return torch.from_numpy(m).to(device) if any_ else None


class TdOps:
    def __init__(self, device: torch.device):
        self.dev = device
        self.lib = N.lib()

    def nstep(self, q_on_next, q_tg_next, q0, actions, rewards, terminated, invalid, weights, discount, retrace_h, double_dqn, rescale):
        B, n, A = q_tg_next.shape
        d = self.dev
        target = torch.empty(B, dtype=torch.float32, device=d)
        loss = torch.empty(1, dtype=torch.float32, device=d)
        grad = torch.empty((B, A), dtype=torch.float32, device=d)
        pri = torch.empty(B, dtype=torch.float32, device=d)
        keep = [t.contiguous() for t in (q_on_next, q_tg_next, q0.detach(), actions, rewards, terminated, weights)]
        inv = invalid.contiguous() if invalid is not None else None
        N.check(
            self.lib.srlx_nstep_td_huber_priority(
                B, n, A, N.tptr(keep[0]), N.tptr(keep[1]), N.tptr(keep[2]), N.tptr(keep[3]), N.tptr(keep[4]), N.tptr(keep[5]), N.tptr(inv),
                N.tptr(keep[6]), float(discount), float(retrace_h), int(double_dqn), int(rescale), N.tptr(target), N.tptr(loss), N.tptr(grad),
                N.tptr(pri), N.torch_stream_ptr(),
            )
        )
        self._keep = keep + [inv]
        return target, loss, grad, pri

    def huber(self, target, q_all, actions, weights):
        """loss / grad / priorities for a given target: the n=1 degenerate case of the fused kernel
        (reward = target, terminated = 1 passes the target through bit-exactly)."""
        B, A = q_all.shape
        z = torch.zeros((B, 1, A), dtype=torch.float32, device=self.dev)
        ones = torch.ones((B, 1), dtype=torch.float32, device=self.dev)
        return self.nstep(z, z, q_all, actions.view(B, 1), target.view(B, 1), ones, None, weights, 0.0, 1.0, True, False)

    def dqn_target(self, q_on_next, q_tg_next, rewards, undone, invalid, discount, double_dqn, rescale, f64_accum, discount_per_sample=None):
        B, A = q_tg_next.shape
        out = torch.empty(B, dtype=torch.float32, device=self.dev)
        keep = [t.contiguous() if t is not None else None for t in (q_on_next, q_tg_next, rewards, undone, invalid, discount_per_sample)]
        N.check(
            self.lib.srlx_dqn_target(B, A, N.tptr(keep[0]), N.tptr(keep[1]), N.tptr(keep[2]), N.tptr(keep[3]), N.tptr(keep[4]), float(discount),
                                     N.tptr(keep[5]), int(double_dqn), int(rescale), int(f64_accum), N.tptr(out), N.torch_stream_ptr())
        )
        self._keep2 = keep
        return out

    def agent57_priority(self, target_ext, q_ext, target_int, q_int, actions, actor_idx, beta_list, n_actions=None):
        """|td_ext + beta[actor] * td_int| with td = target - q[action] (agent57_light/model_torch.py:442,367-373);
        q_ext=None: target_ext / target_int already are TD errors (Agent57's sequence means).
        Returns (td_ext, td_int or None, priorities)."""
        B, A = (q_ext.shape if q_ext is not None else (target_ext.shape[0], n_actions))
        d = self.dev
        pri = torch.empty(B, dtype=torch.float32, device=d)
        td_e = torch.empty(B, dtype=torch.float32, device=d)
        td_i = torch.empty(B, dtype=torch.float32, device=d) if target_int is not None else None
        keep = [t.detach().contiguous() if t is not None else None for t in (target_ext, q_ext, target_int, q_int, actions, actor_idx, beta_list)]
        N.check(
            self.lib.srlx_agent57_priority(B, A, *[N.tptr(t) for t in keep], N.tptr(td_e), N.tptr(td_i), N.tptr(pri), N.torch_stream_ptr())
        )
        self._keep3 = keep
        return td_e, td_i, pri


class NguOps:
    """Device-resident episodic memories + novelty rewards (csrc/srlx_ngu.hip)."""

    def __init__(self, device: torch.device, n_envs: int, emb_dim: int, capacity: int, k: int, epsilon: float, cluster_distance: float, pseudo_counts: float):
        import ctypes

        self.dev, self.lib, self.n_envs, self.emb_dim = device, N.lib(), n_envs, emb_dim
        h = ctypes.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        N.check(self.lib.srlx_ngu_create(ctypes.byref(h), n_envs, emb_dim, capacity, k, float(epsilon), float(cluster_distance), float(pseudo_counts), idx))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.srlx_ngu_destroy(self.h)
            self.h = None

    def reset(self):
        N.check(self.lib.srlx_ngu_reset(self.h, N.torch_stream_ptr()))

    def episodic(self, emb: torch.Tensor, reset: torch.Tensor = None, active: torch.Tensor = None) -> torch.Tensor:
        emb = emb.detach().to(torch.float32).contiguous().view(self.n_envs, self.emb_dim)
        out = torch.empty(self.n_envs, dtype=torch.float32, device=self.dev)
        self._keep = (emb, reset, active)
        N.check(self.lib.srlx_ngu_episodic_reward(self.h, N.tptr(emb), N.tptr(reset), N.tptr(active), N.tptr(out), N.torch_stream_ptr()))
        return out

    def lifelong(self, target: torch.Tensor, train: torch.Tensor, lifelong_max: float) -> torch.Tensor:
        target, train = target.detach().contiguous(), train.detach().contiguous()
        n, dim = target.shape
        out = torch.empty(n, dtype=torch.float32, device=self.dev)
        self._keep_l = (target, train)
        N.check(self.lib.srlx_ngu_lifelong_reward(n, dim, N.tptr(target), N.tptr(train), float(lifelong_max), N.tptr(out), N.torch_stream_ptr()))
        return out
