"""What a plain device-to-device copy reaches on this GPU: the practical ceiling the HBM-bound kernels
(decode, stem, the Cout <= 128 convolutions) are held against in DESIGN.md beside the 8 TB/s nominal figure.

    python tools/hbm_copy_rate.py            # one JSON line: GB/s (read + written bytes) per buffer size
"""
import json

import torch


def rate(nbytes, reps=20):
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    a.random_(0, 255)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def fill_rate(nbytes, reps=20):
    a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    for _ in range(3):
        a.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        a.zero_()
    e1.record()
    torch.cuda.synchronize()
    return 1.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def read_rate(nbytes, reps=20):
    a = torch.ones(nbytes // 4, dtype=torch.float32, device="cuda")
    for _ in range(3):
        a.sum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        a.sum()
    e1.record()
    torch.cuda.synchronize()
    return 1.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


if __name__ == "__main__":
    out = {"device": torch.cuda.get_device_name(0), "unit": "GB/s"}
    for mb in (256, 1024, 4096):
        n = mb << 20
        out["%d MiB" % mb] = {"copy (read+write)": round(rate(n), 1), "fill (write)": round(fill_rate(n), 1),
                              "sum (read)": round(read_rate(n), 1)}
    print(json.dumps(out))
