"""ImageProcessor (srl/rl/processors/image_processor.py:18-151): gray / trim / resize / normalise of image observations, with the
reference's fields and space arithmetic.  The pixel work is `srlx_image_preprocess` (csrc/srlx_image.hip): `remap_observation` is the
single-frame drop-in (frame up, result down), `preprocess_batch` the form the engines use -- raw uint8 frames of E environments already
on the device in, gray uint8 frames for the ring out, nothing touches the host.  OpenCV is not needed (and not installed here); there is
no CPU path: without a GPU `remap_observation` raises."""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from simple_distributed_rl_amd.base.define import SpaceTypes
from simple_distributed_rl_amd.base.spaces.box import BoxSpace

_IMAGE_TYPES = (SpaceTypes.GRAY_HW, SpaceTypes.GRAY_HW1, SpaceTypes.RGB)
_NORM = {"": 0, "0to1": 1, "-1to1": 2}


@dataclass
class ImageProcessor:
    image_type: SpaceTypes = SpaceTypes.GRAY_HW
    resize: Optional[Tuple[int, int]] = None  # (w, h)
    normalize_type: str = ""  # "" | "0to1" | "-1to1"
    trimming: Optional[Tuple[int, int, int, int]] = None  # (top, left, bottom, right)

    # ---- space (image_processor.py:28-101) ------------------------------------------------------------
    def remap_observation_space(self, prev_space, **kwargs):
        if not isinstance(prev_space, BoxSpace) or prev_space.stype not in _IMAGE_TYPES:
            return None
        assert self.image_type in _IMAGE_TYPES
        h, w = prev_space.shape[0], prev_space.shape[1]
        self.top, self.left, self.bottom, self.right = 0, 0, h, w
        new_hw = (h, w)
        if self.trimming is not None:
            top, left, bottom, right = self.trimming
            assert top < bottom and left < right
            self.top, self.left, self.bottom, self.right = max(top, 0), max(left, 0), min(bottom, h), min(right, w)
            new_hw = (self.bottom - self.top, self.right - self.left)
        if self.resize is not None:
            new_hw = (self.resize[1], self.resize[0])
        if "float" in str(np.dtype(prev_space.dtype)):
            self.normalize_type = ""  # :75-77: an already normalised image is passed through
        self.max_val, self.min_val = float(np.max(prev_space.high)), float(np.min(prev_space.low))
        if self.normalize_type == "0to1":
            low, high, dtype = 0, 1, np.float32
        elif self.normalize_type == "-1to1":
            low, high, dtype = -1, 1, np.float32
        else:
            low, high, dtype = self.min_val, self.max_val, prev_space.dtype
        shape = new_hw + ((1,) if self.image_type == SpaceTypes.GRAY_HW1 else ((3,) if self.image_type == SpaceTypes.RGB else ()))
        self._src = (h, w, 3 if prev_space.stype == SpaceTypes.RGB else 1)
        self._out_hw = new_hw
        return BoxSpace(shape, low, high, dtype, self.image_type)

    # ---- pixels ---------------------------------------------------------------------------------------
    def _launch(self, src, out_u8, out_f32):
        from simple_distributed_rl_amd import _native as N

        h, w, ch = self._src
        to_gray = int(ch == 3 and self.image_type != SpaceTypes.RGB)
        N.check(N.lib().srlx_image_preprocess(src.shape[0], h, w, ch, N.tptr(src), to_gray, self.top, self.left, self.bottom, self.right, self._out_hw[0],
                                              self._out_hw[1], N.tptr(out_u8), N.tptr(out_f32), _NORM[self.normalize_type], self.max_val, N.torch_stream_ptr()))

    def preprocess_batch(self, frames_u8, out_u8=None):
        """uint8 device tensor [n, H, W(, 3)] -> uint8 device tensor [n, h, w(, 3)] (gray unless image_type is RGB); no host hop."""
        import torch

        n = frames_u8.shape[0]
        ch = 3 if (self._src[2] == 3 and self.image_type == SpaceTypes.RGB) else 1
        if out_u8 is None:
            out_u8 = torch.empty((n,) + self._out_hw + ((ch,) if ch == 3 else ()), dtype=torch.uint8, device=frames_u8.device)
        self._launch(frames_u8.contiguous(), out_u8, None)
        return out_u8

    def remap_observation(self, state, prev_space, new_space, **kwargs):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("ImageProcessor: the pixel work runs in libsrlx on the GPU; there is no CPU path in this build (the reference uses OpenCV)")
        state = np.asarray(state)
        if "float" in str(state.dtype):  # :127-128,135-136: normalised inputs are neither converted nor resized by the reference
            return state[..., np.newaxis] if (state.ndim == 2 and self.image_type == SpaceTypes.GRAY_HW1) else state
        h, w, ch = self._src
        src = torch.as_tensor(np.ascontiguousarray(state, np.uint8).reshape(1, h, w, ch) if ch == 3 else np.ascontiguousarray(state, np.uint8).reshape(1, h, w)).cuda()
        if ch == 1 and self.image_type == SpaceTypes.RGB:  # gray -> colour: tile (:113-116), then resize per channel
            src = src.reshape(1, h, w, 1).repeat(1, 1, 1, 3).contiguous()
            saved, self._src = self._src, (h, w, 3)
        else:
            saved = None
        out_ch = 3 if self.image_type == SpaceTypes.RGB else 1
        shape = (1,) + self._out_hw + ((3,) if out_ch == 3 else ())
        try:
            if self.normalize_type:
                out = torch.empty(shape, dtype=torch.float32, device="cuda")
                self._launch(src, None, out)
            else:
                out = torch.empty(shape, dtype=torch.uint8, device="cuda")
                self._launch(src, out, None)
        finally:
            if saved is not None:
                self._src = saved
        res = out[0].cpu().numpy()
        if not self.normalize_type:
            res = res.astype(np.dtype(new_space.dtype)) if new_space is not None else res
        return res[..., np.newaxis] if (res.ndim == 2 and self.image_type == SpaceTypes.GRAY_HW1) else res
