"""Per-launch times of the bench plan with N MiB of unrelated device memory read and rewritten before every launch
(outside the timed pair): how much of a layer's speed inside the network is the Infinity Cache holding what the last
layers wrote.  Usage: pollute_probe.py <MiB> [balanced 0|1]."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from yolo_amd import _lib, build_model, config  # noqa: E402
from yolo_amd import darknet_weights as dw  # noqa: E402


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    balanced = len(sys.argv) > 2 and sys.argv[2] == "1"
    net = build_model.Network(config.yolov3_cfg_path, 80, dtype="bf16", balanced=balanced)
    net.set_params(dw.random_params(net.layers, seed=0))
    B, S = 64, 608
    x = torch.rand((B, S, S, 3), device="cuda")
    cp = net.plan_for(B, S, S)
    heads = [torch.zeros((B, hd["h"], hd["w"], hd["ld"]), dtype=torch.float32, device="cuda") for hd in cp.plan.heads]
    env = {"in": x.data_ptr(), "stream": torch.cuda.current_stream().cuda_stream, "heads": [h.data_ptr() for h in heads]}
    junk = torch.zeros(max(mib, 1) << 18, dtype=torch.int32, device="cuda")
    acc = {}
    for rep in range(7):
        cp.begin_pass()
        evs = []
        for i, (op, launch) in enumerate(zip(cp.plan.ops, cp.launchers)):
            if mib:
                junk.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(launch(env))
            e1.record()
            evs.append((i, op, e0, e1))
        torch.cuda.synchronize()
        if rep >= 2:
            for i, op, e0, e1 in evs:
                acc.setdefault(i, [op, []])[1].append(e0.elapsed_time(e1))
    tot = 0.0
    groups = {}
    for i in sorted(acc):
        op, ts = acc[i]
        ms = float(np.median(ts))
        tot += ms
        if op["kind"] == "conv":
            key = "%dx%d s%d %d->%d @%d%s" % (op["ksize"], op["ksize"], op["stride"], op["in_c"], op["out_c"], op["out_hw"][0], "+r" if op["res"] is not None else "")
            groups.setdefault(key, []).append(ms)
    for k, v in groups.items():
        if len(v) >= 3:
            print("%-28s n=%2d  mean %.4f ms" % (k, len(v), sum(v) / len(v)))
    if balanced and cp._ws_flags is not None:       # experimental build (YOLO3HIP_LIB=.../libyolo3hip_exp.so): what the polls saw
        f = cp._ws_flags.cpu().numpy().astype(np.int64).reshape(-1, 256)
        for k, row in enumerate(f):
            taken, rec, raised = int(((row & 255) == 3).sum()), int(((row & 255) == 5).sum()), int((row == 1).sum())
            polls = row[(row & 255) == 3] >> 8
            print("ws launch %2d: taken %3d recomputed %3d raised-only %3d polls max %d mean %.1f" % (k, taken, rec, raised, int(polls.max()) if len(polls) else 0, float(polls.mean()) if len(polls) else 0))
    print("pollute %d MiB balanced %d: total %.3f ms" % (mib, balanced, tot))


if __name__ == "__main__":
    main()
