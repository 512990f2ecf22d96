"""Counterparts of the reference's utils/utils.py helpers, same names and argument order:

    resize_image(image_data, size)                                          utils/utils.py:7
    get_head(output, anchors, num_classes, input_shape, calc_loss=False)    utils/utils.py:43
    correct_boxes(box_xy, box_wh, input_shape, image_shape)                 utils/utils.py:100
    get_boxes_and_scores(output, anchors, num_classes, input_shape, image_shape)   utils/utils.py:132

The three decode helpers run HIP kernels on torch cuda tensors (fp32) and return dense
tensors shaped like the reference's; the fused production path is yolo_amd.postprocess.
resize_image is host-side numpy (the reference uses cv2 on the host as well).
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..postprocess import _head_view, letterbox_params


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _cubic_weights(t, a=-0.75):
    """OpenCV INTER_CUBIC kernel (a = -0.75) for fractional offset t in [0,1)."""
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    return np.stack([w0, w1, w2, 1.0 - w0 - w1 - w2], axis=-1)


def _resize_axis_cubic(img, new_len, axis):
    old_len = img.shape[axis]
    scale = old_len / float(new_len)
    src = (np.arange(new_len) + 0.5) * scale - 0.5
    base = np.floor(src).astype(np.int64)
    w = _cubic_weights(src - base)
    idx = np.clip(base[:, None] + np.arange(-1, 3)[None, :], 0, old_len - 1)     # BORDER_REPLICATE
    moved = np.moveaxis(img, axis, 0)
    out = np.einsum("nk,nk...->n...", w, moved[idx])
    return np.moveaxis(out, 0, axis)


def resize_image(image_data, size):
    """Letterbox (reference utils/utils.py:7-40): aspect-preserving bicubic resize to
    (int(w*scale), int(h*scale)) -- truncation, as written -- then centre-pad with 128.
    The bicubic is a float restatement of cv2.INTER_CUBIC (a=-0.75, replicate border);
    cv2 itself works in fixed point, so individual pixels may differ by 1 LSB: PARITY
    UNPINNED (opencv is not installable here)."""
    image_data = np.asarray(image_data)
    image_height, image_width = image_data.shape[:2]
    input_height, input_width = size
    scale = min(input_width / image_width, input_height / image_height)
    new_width, new_height = int(image_width * scale), int(image_height * scale)
    dx, dy = input_width - new_width, input_height - new_height
    img = image_data.astype(np.float64)
    if (new_height, new_width) != (image_height, image_width):
        img = _resize_axis_cubic(_resize_axis_cubic(img, new_width, 1), new_height, 0)
    img = np.clip(np.floor(img + 0.5), 0, 255).astype(np.uint8)
    top, left = dy // 2, dx // 2
    out = np.full((input_height, input_width) + image_data.shape[2:], 128, dtype=np.uint8)
    out[top:top + new_height, left:left + new_width] = img
    return out


def letterbox_gpu(image, size, out=None):
    """Device form of `resize_image(image, size) / 255.` (utils/utils.py:7-40 + inference.py:242): uint8 RGB
    [h,w,3] (numpy, or a torch tensor already on the GPU) -> fp32 cuda tensor [size_h,size_w,3], written
    into `out` if given (e.g. one image slot of a batch tensor).  Same pixels as the host path."""
    lib = _lib.load()
    if not isinstance(image, torch.Tensor):
        image = torch.from_numpy(np.array(image, dtype=np.uint8))       # a copy: PIL arrays are read-only
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise ValueError("letterbox_gpu: expected a uint8 [h,w,3] image, got %s %r" % (image.dtype, tuple(image.shape)))
    if not image.is_cuda:
        image = image.cuda()
    if image.stride(2) != 1 or image.stride(1) != 3:
        image = image.contiguous()
    h, w = int(image.shape[0]), int(image.shape[1])
    in_h, in_w = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty((in_h, in_w, 3), dtype=torch.float32, device=image.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (in_h, in_w, 3) or not out.is_contiguous():
        raise ValueError("letterbox_gpu: out must be a contiguous fp32 [%d,%d,3] tensor" % (in_h, in_w))
    _lib.check(lib.y3_letterbox_u8(image.data_ptr(), h, w, int(image.stride(0)), out.data_ptr(), in_h, in_w,
                                   _stream(image.device)))
    out._y3_keep = image              # the source must outlive the asynchronous launch
    return out


def get_head(output, anchors, num_classes, input_shape, calc_loss=False):
    """-> box_xy [B,gh,gw,A,2], box_wh [B,gh,gw,A,2], box_confidence [B,gh,gw,A,1],
    box_class_probs [B,gh,gw,A,C] (reference utils/utils.py:43-96)."""
    if calc_loss:
        raise NotImplementedError("calc_loss=True is the training branch (utils/utils.py:94-95): out of scope")
    lib = _lib.load()
    t, ld = _head_view(output)
    B, gh, gw, ch = t.shape
    anchors = np.asarray(anchors, dtype=np.float32).reshape(-1, 2)
    A = anchors.shape[0]
    if ch != A * (5 + num_classes):
        raise ValueError("output has %d channels, expected %d" % (ch, A * (5 + num_classes)))
    dev = t.device
    anc = torch.from_numpy(np.ascontiguousarray(anchors)).to(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    xy = torch.empty((B, gh, gw, A, 2), **f32)
    wh = torch.empty((B, gh, gw, A, 2), **f32)
    conf = torch.empty((B, gh, gw, A, 1), **f32)
    cls = torch.empty((B, gh, gw, A, num_classes), **f32)
    sx = float(np.floor(np.float32(input_shape[1]) / np.float32(gw)))
    sy = float(np.floor(np.float32(input_shape[0]) / np.float32(gh)))
    _lib.check(lib.y3_get_head(t.data_ptr(), B, gh, gw, ld, A, num_classes, anc.data_ptr(), sx, sy,
                               xy.data_ptr(), wh.data_ptr(), conf.data_ptr(), cls.data_ptr(), _stream(dev)))
    torch.cuda.current_stream(dev).synchronize()       # anc is freed on return
    return xy, wh, conf, cls


def correct_boxes(box_xy, box_wh, input_shape, image_shape):
    """-> boxes [..., 4] = (y_min, x_min, y_max, x_max) in source-image pixels, unclipped
    (reference utils/utils.py:100-129)."""
    lib = _lib.load()
    xy = box_xy.float().contiguous()
    wh = box_wh.float().contiguous()
    scale, off_h, off_w = letterbox_params(input_shape, image_shape)
    boxes = torch.empty(xy.shape[:-1] + (4,), dtype=torch.float32, device=xy.device)
    n = xy.numel() // 2
    _lib.check(lib.y3_correct_boxes(xy.data_ptr(), wh.data_ptr(), n, float(scale), float(off_h), float(off_w),
                                    boxes.data_ptr(), _stream(xy.device)))
    torch.cuda.current_stream(xy.device).synchronize()
    return boxes


def get_boxes_and_scores(output, anchors, num_classes, input_shape, image_shape):
    """-> boxes [B*gh*gw*A, 4], box_scores [B*gh*gw*A, C] (reference utils/utils.py:132-160);
    computed by the fused decode kernel, then expanded to the reference's dense layout."""
    lib = _lib.load()
    t, ld = _head_view(output)
    B, gh, gw, ch = t.shape
    anchors = np.asarray(anchors, dtype=np.float32).reshape(-1, 2)
    A = anchors.shape[0]
    if ch != A * (5 + num_classes):
        raise ValueError("output has %d channels, expected %d" % (ch, A * (5 + num_classes)))
    dev = t.device
    d = _lib.DecodeDesc()
    d.batch, d.num_scales, d.num_classes, d.num_anchors = B, 1, num_classes, A
    d.grid_h[0], d.grid_w[0], d.head_ld[0] = gh, gw, ld
    for a in range(A):
        d.anchors[0][a][0], d.anchors[0][a][1] = float(anchors[a][0]), float(anchors[a][1])
    d.stride_x[0] = float(np.floor(np.float32(input_shape[1]) / np.float32(gw)))
    d.stride_y[0] = float(np.floor(np.float32(input_shape[0]) / np.float32(gh)))
    d.score_threshold = float("inf")
    N = gh * gw * A
    d.boxes_per_image = N
    lb = np.zeros((B, 4), dtype=np.float32)
    lb[:, :3] = letterbox_params(input_shape, image_shape)
    lbt = torch.from_numpy(lb).to(dev)
    rec = torch.empty((B, N, 8), dtype=torch.float32, device=dev)
    cand = torch.empty((B, N), dtype=torch.int32, device=dev)
    counts = torch.zeros(B + B * num_classes, dtype=torch.int32, device=dev)
    st = _stream(dev)
    _lib.check(lib.y3_decode(ctypes.byref(d), t.data_ptr(), None, None, lbt.data_ptr(), rec.data_ptr(),
                             cand.data_ptr(), counts[:B].data_ptr(), counts[B:].data_ptr(), st))
    boxes = torch.empty((B * N, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B * N, num_classes), dtype=torch.float32, device=dev)
    _lib.check(lib.y3_expand_scores(rec.data_ptr(), B * N, num_classes, boxes.data_ptr(), scores.data_ptr(), st))
    torch.cuda.current_stream(dev).synchronize()
    return boxes, scores
