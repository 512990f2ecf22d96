"""Oracle (test infrastructure): python face of nms_ref.c plus a pure-numpy twin.

`non_max_suppression` mirrors tf.image.non_max_suppression(boxes, scores,
max_output_size, iou_threshold) as called at inference.py:148-152.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liby3oracle.so")
_lib = None


def build(force=False):
    """gcc the C restatement into oracle/_build/ (the checker, not the product)."""
    src = os.path.join(_HERE, "nms_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        lib.y3o_nms.restype = ctypes.c_int
        lib.y3o_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                ctypes.c_float, ctypes.c_void_p]
        lib.y3o_iou.restype = ctypes.c_float
        lib.y3o_iou.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


def non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    sel = np.empty(max(1, min(n, int(max_output_size))), dtype=np.int32)
    ns = _load().y3o_nms(boxes.ctypes.data, scores.ctypes.data, n, int(max_output_size),
                         ctypes.c_float(np.float32(iou_threshold)), sel.ctypes.data)
    if ns < 0:
        raise MemoryError("y3o_nms")
    return sel[:ns].copy()


def iou_np(bi, bj):
    """fp32 IoU, same operation order as nms_ref.c (numpy scalars keep fp32)."""
    f = np.float32
    bi = np.asarray(bi, dtype=f); bj = np.asarray(bj, dtype=f)
    ymin_i, xmin_i = min(bi[0], bi[2]), min(bi[1], bi[3])
    ymax_i, xmax_i = max(bi[0], bi[2]), max(bi[1], bi[3])
    ymin_j, xmin_j = min(bj[0], bj[2]), min(bj[1], bj[3])
    ymax_j, xmax_j = max(bj[0], bj[2]), max(bj[1], bj[3])
    area_i = f(f(ymax_i - ymin_i) * f(xmax_i - xmin_i))
    area_j = f(f(ymax_j - ymin_j) * f(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return f(0)
    ih = max(f(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f(0))
    iw = max(f(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f(0))
    inter = f(ih * iw)
    return f(inter / f(f(area_i + area_j) - inter))


def non_max_suppression_np(boxes, scores, max_output_size, iou_threshold):
    """Pure-python twin (small inputs only) used to cross-check the C build."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    thr = np.float32(iou_threshold)
    sel = []
    for i in order:
        if len(sel) >= max_output_size:
            break
        if all(not (iou_np(boxes[i], boxes[j]) > thr) for j in sel):
            sel.append(i)
    return np.asarray(sel, dtype=np.int32)
