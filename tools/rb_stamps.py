#!/usr/bin/env python
"""Diagnostic (A/B build with -DY3_RB_STAMPS=1 only): where a tile of the fused residual-block kernel spends its cycles.
Runs one 76x76x256 block at batch 64 a few times and prints the per-segment s_memtime sums of wave 0, median over the
workgroups, in microseconds per tile at the measured shader clock."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from yolo_amd import _lib  # noqa: E402
from yolo_amd.build_model import pack_conv  # noqa: E402

B, H, W, C = 64, int(os.environ.get("RB_H", 76)), int(os.environ.get("RB_H", 76)), int(os.environ.get("RB_C", 256))
lib = _lib.load()
rng = np.random.default_rng(0)
k1 = (rng.standard_normal((1, 1, C, C // 2)) * np.sqrt(2.0 / C)).astype(np.float32)
k2 = (rng.standard_normal((3, 3, C // 2, C)) * np.sqrt(2.0 / (9 * C // 2))).astype(np.float32)
w1, b1, _, k1p, n1p = pack_conv(k1, np.zeros(C // 2, np.float32), "bf16", False, "cuda")
w2, b2, _, k2p, n2p = pack_conv(k2, np.zeros(C, np.float32), "bf16", False, "cuda")
x = torch.randn((B, H, W, C), device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
zeros = torch.zeros(256, dtype=torch.uint8, device="cuda")
d = _lib.ResblockDesc()
d.dtype, d.batch, d.h, d.w, d.c, d.x_ld, d.out_ld = _lib.Y3_BF16, B, H, W, C, C, C
d.k1_pad, d.n1_pad, d.k2_pad, d.n2_pad, d.flags = k1p, n1p, k2p, n2p, 3
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(5):
    e0.record()
    _lib.check(lib.y3_resblock_fused(ctypes.byref(d), x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                     out.data_ptr(), zeros.data_ptr(), st))
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
tiles = ctypes.c_int64(0)
lib.y3_resblock_tiles(ctypes.byref(d), ctypes.byref(tiles))
print("launch %.4f ms, %d tiles (%.2f per workgroup)" % (ms, tiles.value, tiles.value / 256.0))
if hasattr(lib, "y3_resblock_stamps"):
    buf = (ctypes.c_uint64 * (256 * 8))()
    assert lib.y3_resblock_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
    tot = a.sum(axis=1)
    clk = np.median(tot) / (ms * 1e3)            # cycles per microsecond ~ shader clock in MHz (s_memtime ticks at 100 MHz on some parts)
    names = ["W1 issue", "wait x + barrier", "phase 1 MFMA", "barrier + y write", "phase 2", "res loads + epilogue math", "next x + stores issue", "-"]
    per_tile = tiles.value / 256.0
    print("s_memtime ticks per us (median workgroup total / launch time): %.1f" % clk)
    for k in range(7):
        print("  %-28s %8.2f us/tile  (%4.1f %%)" % (names[k], np.median(a[:, k]) / clk / per_tile, 100 * np.median(a[:, k]) / np.median(tot)))
