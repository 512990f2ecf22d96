#!/usr/bin/env python
"""Writes tests/golden/tiny416_b1.npz: BASELINE config 1 (YOLOv3-tiny 416x416, batch 1) run through the CPU
oracle on seeded inputs -- x = default_rng(0).random((1,416,416,3)), weights = darknet_weights.random_params(seed=1),
thresholds 0.6 / 0.3 / 51.  The reference itself cannot run here (TensorFlow 1.12 / OpenCV are not
installed, nothing can be downloaded), so these vectors pin the ORACLE (regression fixture) and give the GPU tests a
committed target; they are not outputs of the reference.  Stored: a strided subsample of both head tensors, float64
sums of the full tensors, and the final boxes / scores / classes.  Usage: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import darknet as od  # noqa: E402
from oracle import decode as odec  # noqa: E402
from yolo_amd import config, inference  # noqa: E402
from yolo_amd import darknet_weights as dw  # noqa: E402

STRIDE = 7           # every 7th value of the flattened head tensor
SCORE_THR = 0.3      # lower than config's 0.6 so that random-init heads leave some detections


def compute():
    cfg = config.yolov3_tiny_cfg_path
    layers, out_index = od.parse_layers(cfg, 80)
    params = dw.random_params(layers, seed=1)
    x = np.random.default_rng(0).random((1, 416, 416, 3), dtype=np.float32)
    heads = od.forward(layers, out_index, params, x)
    anchors = inference.read_anchors(config.tiny_anchors_path)
    boxes, scores, classes = odec.predict(heads, anchors, 80, (416, 416), (375, 500), SCORE_THR, 0.3, 51)
    out = {"boxes": boxes.astype(np.float32), "scores": scores.astype(np.float32), "classes": classes.astype(np.int32)}
    for k, h in enumerate(heads):
        out["head%d_shape" % k] = np.array(h.shape, dtype=np.int32)
        out["head%d_sub" % k] = h.reshape(-1)[::STRIDE].astype(np.float32)
        out["head%d_sum" % k] = np.array([h.astype(np.float64).sum(), np.abs(h.astype(np.float64)).sum()])
    return x, params, layers, out_index, anchors, out


if __name__ == "__main__":
    out = compute()[-1]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny416_b1.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out["classes"]), "detections")
