#!/usr/bin/env python
"""PCIe-inclusive throughput: uint8 images in pinned host memory -> device (second stream) -> y3_letterbox_u8
(/255, fp32 NHWC) -> backbone + decode + NMS, double-buffered so that the upload of batch n+1 runs under the
compute of batch n.  bench.py times the path with inputs resident in HBM (the contract); this is the number
DESIGN.md quotes beside it."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import build_model, config, inference  # noqa: E402
from yolo_amd import darknet_weights as dw  # noqa: E402
from yolo_amd.utils.utils import letterbox_gpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--serial", action="store_true", help="upload, then compute, on one stream (no overlap)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, S = a.batch, a.size
    nodes = build_model.yolo(None, False, config.yolov3_cfg_path, config.num_classes, dtype="bf16", device=dev)
    nodes.net.set_params(dw.random_params(nodes.net.layers, seed=0))
    anchors = inference.read_anchors(config.anchors_path)
    host = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).pin_memory()
    u8 = [torch.empty((B, S, S, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    img = [torch.empty((B, S, S, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    main_s = torch.cuda.current_stream(dev)
    copy_s = main_s if a.serial else torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]

    def upload(k):
        i = k & 1
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(free[i])                       # the step that last read this buffer is done
            u8[i].copy_(host, non_blocking=True)
            for b in range(B):
                letterbox_gpu(u8[i][b], (S, S), out=img[i][b])
            ready[i].record(copy_s)

    def compute(k):
        i = k & 1
        main_s.wait_event(ready[i])
        heads = nodes(img[i])
        det = inference.predict_batch(heads, anchors, config.num_classes, (S, S), (S, S))
        free[i].record(main_s)
        return det

    for e in free:
        e.record(main_s)
    total = a.warmup + a.steps
    upload(0)
    t0 = None
    for k in range(total):
        if k == a.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if k + 1 < total:
            upload(k + 1)
        det = compute(k)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"mode": "serial" if a.serial else "overlapped", "images_per_sec": round(B * a.steps / el, 1),
                      "ms_per_step": round(1e3 * el / a.steps, 3), "batch": B, "size": S,
                      "host_bytes_per_step": int(host.numel()), "detections_image0": int(det.count[0].item())}))


if __name__ == "__main__":
    main()
