"""oracle/gen_golden_qnet84.py -- TEST INFRASTRUCTURE ONLY.  Golden vectors of the reference's OWN Rainbow QNetwork at the
benchmark geometry (84 x 84 x 4 frames, 6 actions, dueling 512: srl/algorithms/rainbow/model_torch.py:15-29 built by
`rl_config.make_parameter()`), evaluated by the reference on CPU torch.

Run here, where /root/reference is importable:  PYTHONPATH=/root/reference python oracle/gen_golden_qnet84.py
Only data travels (tests/golden/qnet84_*.npz): uint8 inputs and the reference's Q-values.  The 8.0 M weights are NOT stored:
`recipe_state_dict` below regenerates them bit for bit from a numpy PCG64 stream (key order = the reference's state_dict order,
which the fixture records), the test rebuilds them with the same function and loads them into the device network.
Two weight sets:
  init      -- uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's default initialisation
  wide      -- the same magnitudes times 2^e, e uniform in [-6, 6] per element, signs alternating in runs of three: large dynamic range and
               heavy cancellation inside every dot product (the case the split-bf16 products have to survive)
`q_ref_f64` is the same reference module evaluated in float64 (module.double()): the yardstick for "how far is a float32 implementation
allowed to be" -- the reference's own float32 result sits at |q_ref_f32 - q_ref_f64|.
"""
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def recipe_state_dict(keys_shapes, kind: str, seed: int = 20260929):
    """keys_shapes: [(key, shape)] in the reference's state_dict order.  Pure numpy: identical on every platform."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for key, shape in keys_shapes:
        shape = tuple(int(s) for s in shape)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        bound = 1.0 / np.sqrt(max(fan_in, 1))
        w = rng.uniform(-bound, bound, size=shape)
        if kind == "wide":
            e = rng.integers(-6, 7, size=shape)
            w = np.abs(w) * np.exp2(e)
            flat = w.reshape(-1)
            sign = np.where((np.arange(flat.size) // 3) % 2 == 0, 1.0, -1.0)  # runs of three: neighbouring products cancel
            w = (flat * sign).reshape(shape) * 0.25
        out[key] = w.astype(np.float32)
    return out


def _build_reference_net():
    import srl
    from srl.algorithms import rainbow
    from srl.base.env import registration

    import _golden_env  # noqa: F401

    registration.register("TinyImageEnvGolden", entry_point="_golden_env:TinyImageEnv", check_duplicate=False)
    env_config = srl.EnvConfig("TinyImageEnvGolden", kwargs=dict(hw=84, actions=6))
    rl_config = rainbow.Config(multisteps=3, batch_size=8, lr=0.00025)
    rl_config.window_length = 4
    rl_config.memory.warmup_size = 8
    rl_config.memory.capacity = 1000
    rl_config.memory.compress = False
    rl_config.hidden_block.set_dueling_network((512,))
    rl_config.set_torch()
    env = env_config.make()
    rl_config.setup(env)
    return env, rl_config


def main():
    import torch

    torch.set_num_threads(8)
    env, rl_config = _build_reference_net()
    torch.manual_seed(0)
    parameter = rl_config.make_parameter()
    net = parameter.q_online
    keys_shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, (6, 84, 84, 4), dtype=np.uint8)
    frames[4] = 0          # an all-black stack
    frames[5, :, :, :2] = 0  # zero history in the two oldest channels (episode start)
    x = torch.tensor(frames.astype(np.float32) / 255)  # the reference feeds (B, H, W, C) float32 in [0, 1]
    for kind in ("init", "wide"):
        sd = recipe_state_dict(keys_shapes, kind)
        net.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
        with torch.no_grad():
            q32 = net(x).numpy()
            net64 = net.double()
            q64 = net64(x.double()).numpy()
            net.float()
        np.savez_compressed(os.path.join(OUT, f"qnet84_{kind}.npz"), frames=frames, q_ref_f32=q32, q_ref_f64=q64,
                            keys=np.array([k for k, _ in keys_shapes]), shapes=np.array([str(s) for _, s in keys_shapes]),
                            kind=np.array(kind), seed=np.array(20260929))
        print(kind, "max|q|", np.abs(q32).max(), "f32 vs f64 max rel", np.abs(q32 - q64).max() / np.abs(q64).max())


if __name__ == "__main__":
    main()
