"""ImageProcessor (reference srl/rl/processors/image_processor.py:18-151).

CPU: the oracle (oracle/image_oracle.py, OpenCV's published 8-bit arithmetic) against the known answers the reference's own tests
hold (tests/quick/rl/processors/test_image_processor.py:29-85,117-140) + the space arithmetic of the mirrored class.
GPU: srlx_image_preprocess bit-equal to the oracle on random images (ALE geometry 210 x 160 x 3 -> 84 x 84, the 2 x 2 area special case,
trimming, gray and RGB), the reference tests' patterns through the class, and the batch path into an engine's frame ring."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import image_oracle as O  # noqa: E402

from simple_distributed_rl_amd.base.define import SpaceTypes  # noqa: E402
from simple_distributed_rl_amd.base.spaces.box import BoxSpace  # noqa: E402
from simple_distributed_rl_amd.rl.processors.image_processor import ImageProcessor  # noqa: E402

W_, H_ = 32, 64
PATTERNS = (
    (SpaceTypes.GRAY_HW, (W_, H_), SpaceTypes.GRAY_HW, (84, 84), True),
    (SpaceTypes.GRAY_HW, (W_, H_), SpaceTypes.GRAY_HW1, (84, 84, 1), True),
    (SpaceTypes.GRAY_HW, (W_, H_), SpaceTypes.RGB, (84, 84, 3), False),
    (SpaceTypes.GRAY_HW1, (W_, H_, 1), SpaceTypes.GRAY_HW, (84, 84), True),
    (SpaceTypes.GRAY_HW1, (W_, H_, 1), SpaceTypes.GRAY_HW1, (84, 84, 1), True),
    (SpaceTypes.GRAY_HW1, (W_, H_, 1), SpaceTypes.RGB, (84, 84, 3), False),
    (SpaceTypes.RGB, (W_, H_, 3), SpaceTypes.GRAY_HW, (84, 84), True),
    (SpaceTypes.RGB, (W_, H_, 3), SpaceTypes.GRAY_HW1, (84, 84, 1), True),
    (SpaceTypes.RGB, (W_, H_, 3), SpaceTypes.RGB, (84, 84, 3), True),
)


def test_oracle_known_answers_of_the_reference_tests():
    """test_image_processor.py:66-85: a constant image stays that constant through gray conversion and resize, for every constant."""
    for c in (0, 1, 7, 128, 254, 255):
        img = np.full((64, 32, 3), c, np.uint8)
        assert (O.rgb_to_gray_u8(img) == c).all()
        assert (O.image_process(img, True, None, (84, 84)) == c).all()
        assert (O.resize_linear_u8(img, (84, 84)) == c).all()
        assert (O.resize_linear_u8(img[:, :, 0], (16, 32)) == c).all()  # the 2 x 2 area path
    out = O.image_process(np.ones((210, 160, 3), np.uint8), True, (10, 10, 20, 20), None, "0to1")  # :117-140
    assert out.shape == (10, 10) and out.dtype == np.float32 and (out == np.float32(1) / np.float32(255)).all()
    np.testing.assert_array_equal(O.image_process(np.ones((4, 4), np.uint8), False, None, None, "-1to1"), np.ones((4, 4), np.float32) / 255 * 2 - 1)
    # pure colours: the 14-bit coefficients round 0.299 / 0.587 / 0.114 of 255 to 76 / 150 / 29
    assert [int(O.rgb_to_gray_u8(np.array([[p]], np.uint8))[0, 0]) for p in ([255, 0, 0], [0, 255, 0], [0, 0, 255])] == [76, 150, 29]
    # a horizontal ramp up-scaled x2: interior samples fall on quarter points of neighbouring source pixels
    ramp = np.arange(0, 80, 10, dtype=np.uint8)[None, :].repeat(2, 0)
    up = O.resize_linear_u8(ramp, (16, 2))
    np.testing.assert_array_equal(up[0], [0, 3, 8, 13, 18, 23, 28, 33, 38, 43, 48, 53, 58, 63, 68, 70])  # (3a + b) / 4 and (a + 3b) / 4, halves round up (the final + 2 >> 2)


@pytest.mark.parametrize("env_type,env_shape,img_type,true_shape,check", PATTERNS)
@pytest.mark.parametrize("norm", ["", "0to1", "-1to1"])
def test_space_arithmetic(env_type, env_shape, img_type, true_shape, check, norm):
    """test_image_processor.py:29-64 without the pixels."""
    p = ImageProcessor(image_type=img_type, resize=(84, 84), normalize_type=norm)
    space = BoxSpace(env_shape, 0, 255, np.uint8, env_type)
    new = p.remap_observation_space(space)
    assert new is not None and new.stype == img_type and new.shape == true_shape
    assert np.dtype(new.dtype) == (np.float32 if norm else np.uint8)
    lo, hi = {"": (0, 255), "0to1": (0, 1), "-1to1": (-1, 1)}[norm]
    np.testing.assert_array_equal(new.low, np.full(true_shape, lo))
    np.testing.assert_array_equal(new.high, np.full(true_shape, hi))
    trim = ImageProcessor(image_type=SpaceTypes.GRAY_HW, trimming=(10, 10, 20, 20))
    assert trim.remap_observation_space(BoxSpace((210, 160, 3), 0, 255, np.uint8, SpaceTypes.RGB)).shape == (10, 10)
    assert ImageProcessor().remap_observation_space(BoxSpace((4,), 0, 1, np.float32, SpaceTypes.CONTINUOUS)) is None


@pytest.mark.gpu
@pytest.mark.parametrize("env_type,env_shape,img_type,true_shape,check", PATTERNS)
@pytest.mark.parametrize("norm", ["", "0to1", "-1to1"])
def test_reference_patterns_through_the_class(env_type, env_shape, img_type, true_shape, check, norm):
    """test_image_processor.py:66-85: image of ones -> ones (/255, *2/255-1), every type combination."""
    p = ImageProcessor(image_type=img_type, resize=(84, 84), normalize_type=norm)
    space = BoxSpace(env_shape, 0, 255, np.uint8, env_type)
    new = p.remap_observation_space(space)
    image = np.ones(env_shape).astype(np.uint8)
    want = {"": np.ones(true_shape).astype(np.uint8), "0to1": np.ones(true_shape).astype(np.float32) / 255,
            "-1to1": np.ones(true_shape).astype(np.float32) / 255 * 2 - 1}[norm]
    got = p.remap_observation(image, space, new)
    assert got.shape == want.shape and got.dtype == want.dtype
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,to_gray,trim,resize", [((210, 160, 3), True, None, (84, 84)), ((210, 160, 3), False, None, (84, 84)), ((168, 168, 3), True, None, (84, 84)),
                                                      ((210, 160, 3), True, (34, 0, 194, 160), (84, 84)), ((100, 60), False, None, (33, 77)),
                                                      ((50, 50, 3), True, (5, 7, 45, 40), None), ((84, 84), False, None, (84, 84))])
def test_kernel_bit_equal_to_the_oracle(shape, to_gray, trim, resize):
    import torch

    from simple_distributed_rl_amd import _native as N

    rng = np.random.default_rng(sum(shape))
    n = 5
    imgs = rng.integers(0, 256, (n,) + shape, dtype=np.uint8)
    h, w = shape[0], shape[1]
    ch = shape[2] if len(shape) == 3 else 1
    top, left, bottom, right = trim if trim else (0, 0, h, w)
    ow, oh = resize if resize else (right - left, bottom - top)
    oc = 1 if (to_gray or ch == 1) else 3
    src = torch.as_tensor(imgs).cuda()
    out = torch.empty((n, oh, ow) + ((3,) if oc == 3 else ()), dtype=torch.uint8, device="cuda")
    outf = torch.empty_like(out, dtype=torch.float32)
    N.check(N.lib().srlx_image_preprocess(n, h, w, ch, N.tptr(src), int(to_gray), top, left, bottom, right, oh, ow, N.tptr(out), N.tptr(outf), 1, 255.0, None))
    torch.cuda.synchronize()
    for k in range(n):
        want = O.image_process(imgs[k], to_gray, trim, resize)
        np.testing.assert_array_equal(out[k].cpu().numpy(), want)
        np.testing.assert_array_equal(outf[k].cpu().numpy(), O.image_process(imgs[k], to_gray, trim, resize, "0to1"))


@pytest.mark.gpu
def test_batch_path_feeds_the_frame_ring():
    """ALE-shaped frames of E environments, already on the device -> one launch -> the uint8 frames a DeviceReplay commits."""
    import torch

    from simple_distributed_rl_amd.device.replay import DeviceReplay

    E = 6
    p = ImageProcessor(image_type=SpaceTypes.GRAY_HW, resize=(84, 84), normalize_type="0to1")
    p.remap_observation_space(BoxSpace((210, 160, 3), 0, 255, np.uint8, SpaceTypes.RGB))
    rng = np.random.default_rng(1)
    raw = rng.integers(0, 256, (E, 210, 160, 3), dtype=np.uint8)
    frames = p.preprocess_batch(torch.as_tensor(raw).cuda())
    assert frames.shape == (E, 84, 84) and frames.dtype == torch.uint8
    r = DeviceReplay(E, 16, 84 * 84, 4, 3, 4, batch_size=4, warmup_size=1)
    r.reset_all(frames.view(E, -1))
    stack = r.stack_current().view(E, 4, 84, 84)
    torch.cuda.synchronize()
    for e in range(E):
        np.testing.assert_array_equal(stack[e, 3].cpu().numpy(), O.image_process(raw[e], True, None, (84, 84), "0to1"))  # newest frame; history is zeros
    assert float(stack[:, :3].abs().max()) == 0.0


@pytest.mark.gpu
def test_runner_on_raw_rgb_frames_with_the_device_processor():
    """An ALE-shaped host environment (210 x 160 x 3 uint8) + `processors=[ImageProcessor(GRAY_HW, resize=(84, 84), "0to1")]`, the reference's
    Atari preprocessing, through `Runner.train()` on the engine: raw frames go up, the kernel fills the ring, Rainbow trains."""
    import simple_distributed_rl_amd as srl
    from simple_distributed_rl_amd.algorithms import rainbow
    from simple_distributed_rl_amd.base.env import registration

    registration.register("RawRgbFrames", "test_image_processor:RawRgb", check_duplicate=False)
    cfg = rainbow.Config()
    cfg.set_atari_config()
    cfg.enable_noisy_dense = False
    cfg.window_length = 4
    cfg.memory.capacity, cfg.memory.warmup_size = 4 * 64, 32
    cfg.batch_size = 8
    cfg.hidden_block.set_dueling_network((64,))
    cfg.processors = [ImageProcessor(SpaceTypes.GRAY_HW, (84, 84), "0to1")]
    runner = srl.Runner("RawRgbFrames", cfg)
    runner.set_vector_envs(4)
    st = runner.train(max_steps=4 * 30, train_interval=4)
    assert runner.vector_reason == "" and st.total_step == 120 and st.train_count > 0
    eng = runner._vector_actor.engine
    env = eng.env
    assert env.processor is not None and env.F == 84 * 84
    # the newest frame of lane 0 in the ring is the processed last raw frame of host environment 0
    stack = eng.replay.stack_current().view(4, 4, 84, 84)[0, 3].cpu().numpy()
    np.testing.assert_array_equal(stack, O.image_process(np.asarray(env.envs[0].state, np.uint8), True, None, (84, 84), "0to1"))


class RawRgb:
    pass


def _define_raw():
    from simple_distributed_rl_amd.base.env.base import EnvBase
    from simple_distributed_rl_amd.base.spaces.discrete import DiscreteSpace

    class _RawRgb(EnvBase):
        def __init__(self):
            super().__init__()
            self.rng = np.random.default_rng(0)

        action_space = property(lambda self: DiscreteSpace(4))
        observation_space = property(lambda self: BoxSpace((210, 160, 3), 0, 255, np.uint8, SpaceTypes.RGB))
        max_episode_steps = property(lambda self: 100)
        player_num = property(lambda self: 1)

        def _frame(self):
            return self.rng.integers(0, 256, (210, 160, 3), dtype=np.uint8)

        def reset(self, **kw):
            self.t = 0
            return self._frame()

        def step(self, action):
            self.t += 1
            return self._frame(), 1.0, self.t >= 11, False

        def backup(self, **kw):
            return None

        def restore(self, d, **kw):
            pass

    return _RawRgb


RawRgb = _define_raw()
