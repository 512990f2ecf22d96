#!/usr/bin/env python
"""A/B on the GPU: the greedy-round NMS kernels (y3_nms_bucketed, the product) against the all-pairs IoU bitmask + scan
form BASELINE's north_star sketches (y3_nms_bitmask, libyolo3hip_exp.so only), on what the bench network's decode
produces: the headline thresholds (score >= 0.6) and BASELINE config 5 (dense stress, score >= 0.001: every box of the
image is a candidate, three classes of random-init heads hold ~5 750 each).  Kept indices must agree bit for bit; the
times are HIP-event averages of the NMS entry point alone (bucket + its kernels), same stream, same inputs.

    python tools/nms_ab.py [--batch 64] [--size 608] [--reps 10] > profiles/r02c/nms_ab.json
"""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib, build_model, config, inference       # noqa: E402
from yolo_amd import darknet_weights as dw                        # noqa: E402
from yolo_amd import planner                                      # noqa: E402
from yolo_amd.postprocess import PostProcessor                    # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    B, S = args.batch, args.size
    cfg = config.yolov3_cfg_path
    layers, _ = planner.parse_cfg(cfg, 80)
    anchors = inference.read_anchors(config.anchors_path)
    out = {"batch": B, "size": S, "cases": []}
    with _lib.experimental() as lib:
        net = build_model.Network(cfg, 80, dtype="bf16")
        net.set_params(dw.random_params(layers, seed=0))
        x = torch.rand((B, S, S, 3), dtype=torch.float32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
        heads = net(x)
        stream = torch.cuda.current_stream().cuda_stream
        for name, thr in (("headline (score >= 0.6)", 0.6), ("config 5 dense stress (score >= 0.001)", 0.001)):
            pp = PostProcessor(anchors, 80, (S, S), thr, 0.3, 51)
            views, d, ws, (_, N, _) = pp.prepare(heads, (S, S))
            pp.launch(views, d, ws, (B, N, 3))            # decode: records, candidate lists, class counts
            torch.cuda.synchronize()
            C = 80
            cand_count, class_count = ws["counts"][:B], ws["counts"][B:]
            clc = class_count.cpu().numpy().reshape(B, C).astype(np.int64)
            words = int(max(1, (clc * ((clc + 63) // 64)).sum(axis=1).max()))
            mask = torch.empty((B * words,), dtype=torch.int64, device="cuda")
            status = torch.zeros((B,), dtype=torch.int32, device="cuda")
            k2 = torch.full_like(ws["kept_idx"], -1)
            c2 = torch.full_like(ws["kept_count"], -1)
            ws2 = torch.empty_like(ws["nms_ws"])

            def greedy():
                _lib.check(lib.y3_nms_bucketed(ws["records"].data_ptr(), ws["cand_idx"].data_ptr(), cand_count.data_ptr(),
                                               class_count.data_ptr(), B, N, C, 51, pp.iou_threshold, ws["nms_ws"].data_ptr(),
                                               ws["kept_idx"].data_ptr(), None, ws["kept_count"].data_ptr(), stream))

            def bitmask():
                _lib.check(lib.y3_nms_bitmask(ws["records"].data_ptr(), ws["cand_idx"].data_ptr(), cand_count.data_ptr(),
                                              class_count.data_ptr(), B, N, C, 51, pp.iou_threshold, ws2.data_ptr(),
                                              mask.data_ptr(), words, status.data_ptr(), k2.data_ptr(), c2.data_ptr(), stream))

            t_g, t_b = timed(greedy, args.reps), timed(bitmask, args.reps)
            kc, kc2 = ws["kept_count"].cpu().numpy(), c2.cpu().numpy()
            ki, ki2 = ws["kept_idx"].cpu().numpy(), k2.cpu().numpy()
            st = status.cpu().numpy()
            ok_img = st == 0                       # images whose masks fit and whose class lists could be bucketed
            same = bool((kc[ok_img] == kc2[ok_img]).all())
            for b in np.nonzero(ok_img)[0]:
                for c in range(C):
                    same = same and bool((ki[b, c, :kc[b, c]] == ki2[b, c, :kc[b, c]]).all())
            out["cases"].append({
                "case": name, "candidates_per_image_mean": float(cand_count.float().mean().item()),
                "largest_class_mean": float(clc.max(axis=1).mean()), "pairs_over_1024": int((clc > 1024).sum()),
                "iou_tests_bitmask": int((clc * (clc - 1) // 2).sum()), "mask_MB_per_image": words * 8 / 1e6,
                "kept_total": int(kc.sum()), "images_compared": int(ok_img.sum()),
                "images_not_bucketable (ties make the class lists longer than N; greedy falls back to its scan kernel, "
                "the bitmask form reports status 1)": int((st != 0).sum()),
                "kept_indices_identical": same,
                "greedy_ms": round(t_g, 4), "bitmask_ms": round(t_b, 4), "bitmask_over_greedy": round(t_b / t_g, 2)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
