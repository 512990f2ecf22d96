#!/usr/bin/env python
"""Per-launch timing report of one forward plan (HIP events), with each launch priced against
its own roofline: MFMA time at the dense peak and HBM time for its algorithmic bytes."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import build_model, config, planner  # noqa: E402
from yolo_amd import darknet_weights as dw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov3")
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--no-stem", action="store_true")
    ap.add_argument("--splitk", action="store_true")
    ap.add_argument("--no-resblock", action="store_true", help="residual blocks as two launches (1x1, 3x3 + shortcut)")
    ap.add_argument("--decode", action="store_true", help="time the step's launches with the decode fused into the head convolutions")
    ap.add_argument("--phase", type=int, default=-1, help="de-phasing offset in percent of a tile period (0 = off)")
    a = ap.parse_args()
    cfg = config.yolov3_tiny_cfg_path if a.model == "yolov3-tiny" else config.yolov3_cfg_path
    from yolo_amd import _lib
    if a.tile or a.phase >= 0:           # overrides exist only in libyolo3hip_exp.so (-DY3_EXPERIMENTS)
        _lib._lib = _lib.load_experimental()
        _lib.check(_lib.load().y3_set_conv_tile(a.tile))
        if a.phase >= 0:
            _lib.check(_lib.load().y3_set_conv_phase(a.phase))
    net = build_model.Network(cfg, 80, dtype=a.dtype, stem=not a.no_stem, splitk=a.splitk,
                              resblock=False if a.no_resblock else None)
    net.set_params(dw.random_params(net.layers, seed=0))
    x = torch.rand((a.batch, a.size, a.size, 3), device="cuda")
    pp = None
    if a.decode:
        from yolo_amd import inference
        anchors = inference.read_anchors(config.tiny_anchors_path if a.model == "yolov3-tiny" else config.anchors_path)
        pp = inference._postprocessor(anchors, config.num_classes, (a.size, a.size))

    def run(trace=None):
        if pp is not None:
            pp.detect(net, x, (a.size, a.size), chunk=a.chunk or None, trace=trace)
        else:
            net.forward(x, chunk=a.chunk or None, trace=trace)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    acc = {}
    for _ in range(a.reps):
        tr = []
        run(tr)
        torch.cuda.synchronize()
        for i, (op, e0, e1) in enumerate(tr):
            acc.setdefault(i, [op, []])[1].append(e0.elapsed_time(e1))
    es = 2 if a.dtype == "bf16" else 4
    peak = 2500e12 if a.dtype == "bf16" else 157.3e12
    nb = a.chunk or a.batch
    tot = tot_ideal = 0.0
    print("%3s %5s %-22s %9s %8s %8s %8s %7s" % ("op", "layer", "shape", "ms", "TFLOP/s", "GB/s", "ideal_ms", "eff"))
    for i in sorted(acc):
        op, ts = acc[i]
        ms = float(np.median(ts))
        flops, byt = planner.op_flops(op, nb), planner.op_bytes(op, nb, es)
        if op["kind"] == "conv":
            oh = op["out_hw"][0]
            shape = "%dx%d s%d %d->%d @%d%s%s" % (op["ksize"], op["ksize"], op["stride"], op["in_c"], op["out_c"], oh,
                                                  "+r" if op["res"] is not None else "", "+u" if op["ups"] else "")
        elif op["kind"] == "resblock":
            shape = "block 1x1+3x3+r %d->%d->%d @%d" % (op["c"], op["c"] // 2, op["c"], op["hw"][0])
        elif op["kind"] == "stem":
            shape = "stem 3->" + "->".join(str(c["out_c"]) for c in op["convs"]) + " @%d" % op["out_hw"][0]
        else:
            shape = op["kind"]
        ideal = max(flops / peak, byt / 6.3e12) * 1e3
        tot += ms; tot_ideal += ideal
        print("%3d %5d %-22s %9.4f %8.1f %8.0f %8.4f %6.1f%%" % (i, op["layer"], shape, ms, flops / ms / 1e9, byt / ms / 1e6,
                                                                ideal, 100 * ideal / ms if ms else 0))
    print("total %.3f ms   sum of per-launch rooflines %.3f ms (MFMA peak %.0f TF, HBM 6.3 TB/s achievable)" % (tot, tot_ideal, peak / 1e12))


if __name__ == "__main__":
    main()
