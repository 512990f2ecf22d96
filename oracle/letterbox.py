"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's letterbox
pre-process, utils/utils.py:7-40 (`resize_image`) followed by inference.py:242 (`/255.`).

The reference calls cv2.resize(..., interpolation=cv2.INTER_CUBIC) and cv2.copyMakeBorder; OpenCV
(opencv-python 3.4.4.19, requirements.txt:14) is not installed here and cannot be, so the bicubic below
restates its published algorithm in float64 -- Keys' cubic with a = -0.75, source coordinate
(x + 0.5) * scale - 0.5, replicated border, separable (columns first), result rounded half-up and
saturated to uint8.  cv2 itself evaluates the same kernel in 11-bit fixed point, so single pixels can
differ from it by 1 LSB: PARITY UNPINNED against the real library; pinned only by hand-checked
properties (identity at scale 1, constant images, exact 2x2 -> 4x4 known answer) in tests/.
"""
import numpy as np


def cubic_weights(t, a=-0.75):
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    return [w0, w1, w2, 1.0 - w0 - w1 - w2]


def _axis(new_len, old_len):
    scale = old_len / float(new_len)
    src = (np.arange(new_len) + 0.5) * scale - 0.5
    base = np.floor(src)
    w = cubic_weights(src - base)
    idx = [np.clip(base.astype(np.int64) + k - 1, 0, old_len - 1) for k in range(4)]
    return w, idx


def resize_image(image_data, size):
    """uint8 [h,w,3] -> uint8 [size_h,size_w,3], utils/utils.py:7-40 (new size by int() truncation, :22-23;
    pad top = dy//2, left = dx//2 with 128, :34-38)."""
    image_data = np.asarray(image_data)
    h, w = image_data.shape[:2]
    in_h, in_w = size
    scale = min(in_w / w, in_h / h)
    new_w, new_h = int(w * scale), int(h * scale)
    img = image_data.astype(np.float64)
    if (new_h, new_w) != (h, w):
        wx, ix = _axis(new_w, w)
        wy, iy = _axis(new_h, h)
        acc = np.zeros((new_h, new_w, image_data.shape[2]), dtype=np.float64)
        for ky in range(4):
            row = np.zeros((new_h, new_w, image_data.shape[2]), dtype=np.float64)
            for kx in range(4):
                row = row + wx[kx][None, :, None] * img[iy[ky]][:, ix[kx]]
            acc = acc + wy[ky][:, None, None] * row
        img = acc
    img = np.clip(np.floor(img + 0.5), 0, 255).astype(np.uint8)
    out = np.full((in_h, in_w, image_data.shape[2]), 128, dtype=np.uint8)
    top, left = (in_h - new_h) // 2, (in_w - new_w) // 2
    out[top:top + new_h, left:left + new_w] = img
    return out


def network_input(image_data, size):
    """float32 [1,S,S,3] as the reference feeds it (inference.py:242-243: float64 /255., then the f32 placeholder)."""
    return np.expand_dims(resize_image(image_data, size) / 255.0, 0).astype(np.float32)
