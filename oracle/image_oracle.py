"""oracle/image_oracle.py -- TEST INFRASTRUCTURE ONLY.  CPU restatement (numpy) of what the reference's ImageProcessor computes
(srl/rl/processors/image_processor.py:104-151): colour -> gray, trimming, resize, normalisation.

The arithmetic lives in a third-party dependency that is absent from /root/reference and from this image: OpenCV (`cv2.cvtColor(...,
COLOR_RGB2GRAY)` at :129, `cv2.resize(state, self.resize)` at :138 with the default INTER_LINEAR; the reference pins no version:
`opencv-python` in its optional requirements).  Restated from OpenCV 4.x's published 8-bit algorithms (modules/imgproc/src/color_rgb.cpp
RGB2Gray<uchar>; resize.cpp resizeGeneric_ / HResizeLinear / VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>):
    gray   = (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14                       (coefficients in 14-bit fixed point: 0.299 / 0.587 / 0.114)
    resize : fx = (dx + 0.5) * (src_w / dst_w) - 0.5 in float32, sx = floor(fx), fx -= sx, clamped at both borders (weight 0 there);
             weights as int16 = round-half-even(w * 2048); rows are interpolated horizontally into int32 (scale 2^11), then
             dst = ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
             an exact 2 x 2 down-scale is OpenCV's INTER_AREA fast path instead: (s00 + s01 + s10 + s11 + 2) >> 2.
Pinned by: the known answers the reference's own tests hold (tests/quick/rl/processors/test_image_processor.py:29-85,117-140: a constant
image stays that constant through gray / resize / normalise, the output shapes, the trimming window) -- see tests/test_image_processor.py.
Parity with cv2 on arbitrary images is UNPINNED here (cv2 is not installed, so no golden vector could be recorded).
"""
import numpy as np


def rgb_to_gray_u8(img: np.ndarray) -> np.ndarray:
    img = np.asarray(img, np.uint8).astype(np.int32)
    return ((img[..., 0] * 4899 + img[..., 1] * 9617 + img[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def _axis_table(src: int, dst: int):
    """(first source index, int16 weights [dst][2]) of one axis."""
    scale = np.float64(src) / np.float64(dst)
    idx = np.zeros(dst, np.int64)
    w = np.zeros((dst, 2), np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        idx[d] = s
        w[d, 0] = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048)))  # cvRound: half to even
        w[d, 1] = int(np.rint(f * np.float32(2048)))
    return idx, w


def resize_linear_u8(img: np.ndarray, size_wh) -> np.ndarray:
    """cv2.resize(img, (w, h)) for uint8 images [H][W] or [H][W][C], INTER_LINEAR."""
    img = np.asarray(img, np.uint8)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    H, W, C = img.shape
    dw, dh = int(size_wh[0]), int(size_wh[1])
    if (dw, dh) == (W, H):
        return img[..., 0].copy() if squeeze else img.copy()
    src = img.astype(np.int64)
    if W == 2 * dw and H == 2 * dh:  # INTER_AREA fast path
        out = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
    else:
        xi, xw = _axis_table(W, dw)
        yi, yw = _axis_table(H, dh)
        x1 = np.minimum(xi + 1, W - 1)
        rows = src[:, xi, :] * xw[:, 0][None, :, None] + src[:, x1, :] * xw[:, 1][None, :, None]  # [H][dw][C], scale 2^11
        y1 = np.minimum(yi + 1, H - 1)
        s0, s1 = rows[yi], rows[y1]
        out = (((yw[:, 0][:, None, None] * (s0 >> 4)) >> 16) + ((yw[:, 1][:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out


def image_process(img: np.ndarray, to_gray: bool, trimming=None, resize=None, normalize_type: str = "", max_val: float = 255.0):
    """remap_observation (:104-151) for a uint8 image: optional RGB -> gray, trimming (top, left, bottom, right), resize (w, h),
    normalisation ("" | "0to1" | "-1to1")."""
    state = np.asarray(img, np.uint8)
    if to_gray and state.ndim == 3 and state.shape[2] == 3:
        state = rgb_to_gray_u8(state)
    if trimming is not None:
        top, left, bottom, right = trimming
        state = state[top:bottom, left:right]
    if resize is not None:
        state = resize_linear_u8(state, resize)
    if normalize_type == "0to1":
        state = state.astype(np.float32)
        state /= np.float32(max_val)
    elif normalize_type == "-1to1":
        state = state.astype(np.float32)
        state = (state * np.float32(2.0) / np.float32(max_val)) - np.float32(1.0)
    return state
