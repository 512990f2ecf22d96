"""What the balanced split of conv_pp_kernel does on the bench's own layer shapes: time per launch without a workspace
(whole tiles, partial last round), with one (k-steps dealt evenly), and with every tail recomputing its head (test hook),
plus what the tails' polls saw (experimental build: taken / recomputed, polls needed).  GPU only; timing tool, no checks."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from yolo_amd import _lib, build_model  # noqa: E402

SHAPES = [(64, 76, 128, 256, 1, True), (64, 38, 256, 512, 1, True), (64, 19, 512, 1024, 1, True), (64, 38, 512, 1024, 2, False)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    st = torch.cuda.current_stream().cuda_stream
    with _lib.experimental() as lib:
        _lib.check(lib.y3_set_conv_tile(tile))
        for B, H, C, O, stride, res in SHAPES:
            Ho = H // stride
            rng = np.random.default_rng(1)
            k = (rng.standard_normal((3, 3, C, O)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
            wt, bt, fam, k_pad, n_pad = build_model.pack_conv(k, rng.standard_normal(O).astype(np.float32), "bf16", False, "cuda")
            x = torch.randn((B, H, H, C), device="cuda").to(torch.bfloat16)
            r = torch.randn((B, Ho, Ho, O), device="cuda").to(torch.bfloat16)
            out = torch.empty((B, Ho, Ho, O), device="cuda", dtype=torch.bfloat16)
            zeros = torch.zeros(256, dtype=torch.uint8, device="cuda")
            d = _lib.ConvDesc()
            d.dtype, d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = _lib.Y3_BF16, B, H, H, C, C
            d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld = Ho, Ho, O, O, O
            d.ksize, d.stride, d.k_pad, d.n_pad = 3, stride, k_pad, n_pad
            base = _lib.CONV_LEAKY | (_lib.CONV_RESIDUAL if res else 0)
            nbytes = ctypes.c_int64(0)
            d.flags = base
            _lib.check(lib.y3_conv_workspace_bytes(ctypes.byref(d), ctypes.byref(nbytes)))
            slabs = torch.empty(max(int(nbytes.value), 16), dtype=torch.uint8, device="cuda")
            flags = torch.zeros(256, dtype=torch.int32, device="cuda")
            rp = r.data_ptr() if res else None
            row = "%dx%d %d->%d s%d%s slabs %d MiB:" % (H, H, C, O, stride, "+r" if res else "", nbytes.value >> 20)
            for mode in ("plain", "ws", "ws_recompute"):
                d.flags = base | ((1 << 30) if mode == "ws_recompute" else 0)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
                for i in range(reps + 3):
                    if mode != "plain":
                        flags.zero_()
                    if i >= 3:
                        ev[2 * (i - 3)].record()
                    if mode == "plain":
                        rc = lib.y3_conv2d_fused(ctypes.byref(d), x.data_ptr(), wt.data_ptr(), bt.data_ptr(), rp, out.data_ptr(), zeros.data_ptr(), st)
                    else:
                        rc = lib.y3_conv2d_fused_ws(ctypes.byref(d), x.data_ptr(), wt.data_ptr(), bt.data_ptr(), rp, out.data_ptr(), zeros.data_ptr(),
                                                    slabs.data_ptr(), slabs.numel(), flags.data_ptr(), st)
                    _lib.check(rc)
                    if i >= 3:
                        ev[2 * (i - 3) + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(reps))
                row += "  %s %.4f ms" % (mode, ts[len(ts) // 2])
                if mode == "ws":
                    f = flags.cpu().numpy().astype(np.int64)
                    taken, rec, raised = int(((f & 255) == 3).sum()), int(((f & 255) == 5).sum()), int((f == 1).sum())
                    polls = f[(f & 255) == 3] >> 8
                    row += " (taken %d, recomputed %d, raised only %d, polls max %d)" % (taken, rec, raised, int(polls.max()) if len(polls) else 0)
            print(row, flush=True)


if __name__ == "__main__":
    main()
