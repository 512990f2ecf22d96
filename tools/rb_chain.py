#!/usr/bin/env python
"""Experiment: a chain of N fused residual blocks (ping-pong buffers, as the arena lays them out) over 64 images,
run whole-batch or depth-first in sub-batches that keep x + out inside the 256 MiB Infinity Cache."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib  # noqa: E402
from yolo_amd.build_model import pack_conv  # noqa: E402

B, H, C, N = 64, int(os.environ.get("RB_H", 76)), int(os.environ.get("RB_C", 256)), int(os.environ.get("RB_N", 8))
lib = _lib.load()
rng = np.random.default_rng(0)
ws = []
for i in range(N):
    k1 = (rng.standard_normal((1, 1, C, C // 2)) * np.sqrt(2.0 / C)).astype(np.float32)
    k2 = (rng.standard_normal((3, 3, C // 2, C)) * np.sqrt(2.0 / (9 * C // 2)) * 0.2).astype(np.float32)
    ws.append(pack_conv(k1, np.zeros(C // 2, np.float32), "bf16", False, "cuda") + pack_conv(k2, np.zeros(C, np.float32), "bf16", False, "cuda"))
bufs = [torch.randn((B, H, H, C), device="cuda").to(torch.bfloat16), torch.empty((B, H, H, C), dtype=torch.bfloat16, device="cuda")]
zeros = torch.zeros(256, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(sub):
    d = _lib.ResblockDesc()
    d.dtype, d.batch, d.h, d.w, d.c, d.x_ld, d.out_ld = _lib.Y3_BF16, sub, H, H, C, C, C
    img = H * H * C * 2
    for s in range(0, B, sub):
        for i in range(N):
            w1, b1, _, k1p, n1p, w2, b2, _, k2p, n2p = ws[i]
            d.k1_pad, d.n1_pad, d.k2_pad, d.n2_pad, d.flags = k1p, n1p, k2p, n2p, 3 | (32 if i & 1 else 0)
            src, dst = bufs[i & 1], bufs[(i + 1) & 1]
            _lib.check(lib.y3_resblock_fused(ctypes.byref(d), src.data_ptr() + s * img, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                             b2.data_ptr(), dst.data_ptr() + s * img, zeros.data_ptr(), st))


for sub in (64, 32, 16, 64, 32, 16):
    run(sub)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run(sub)
    e1.record()
    torch.cuda.synchronize()
    print("sub-batch %2d: %.4f ms per chain of %d blocks (%.4f per block)" % (sub, e0.elapsed_time(e1) / 3, N, e0.elapsed_time(e1) / 3 / N))
