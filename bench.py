#!/usr/bin/env python
"""Headline benchmark: YOLOv3-608 end-to-end (backbone + decode + per-class NMS) images/sec.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: 64 images of 608x608x3 per GPU (BASELINE.json configs[3]), seeded
random-init Darknet-53 weights, bf16 compute.  Ranks shard the global batch by image
(no data-path collective); rank 0 prints ONE JSON line.  `roofline` prices the dominant
kernel (the bf16 implicit-GEMM convolution variant most of the step's time goes to) against the dense bf16 MFMA peak
from HIP-event timings taken inside the timed region; `cpu_baseline` times the numpy/C
oracle (a CPU restatement, NOT TensorFlow) on the host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from yolo_amd import build_model, config, inference, planner, sharding  # noqa: E402
from yolo_amd import darknet_weights as dw  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md chip table
F32_MFMA_PEAK_TFLOPS = 157.3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="yolov3", choices=["yolov3", "yolov3-tiny"])
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=0, help="sub-batch the plan runs at a time (0 = whole batch)")
    ap.add_argument("--score-threshold", type=float, default=config.score_threshold)
    ap.add_argument("--max-boxes", type=int, default=config.max_boxes, help="per-class NMS cap (experiments; the reference's 51 is the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=24, help="timed images of the CPU baseline (plus 1 warm-up)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--unfused-decode", action="store_true", help="A/B: write the fp32 heads and run y3_decode on them instead of "
                    "decoding inside the head convolutions' epilogue")
    ap.add_argument("--event-every", type=int, default=4,
                    help="timed steps that carry HIP events around the dominant kernel's launches: every Nth (an event "
                         "pair costs ~3 us of stream time; 25 pairs on every step slowed the measured step by 1.6 %%)")
    ap.add_argument("--splitk", action="store_true", help="split launches with few output tiles over k (small-batch latency mode)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one HIP graph (yolo_amd.graph.GraphedDetector): the serving mode at small "
                         "batch; per-kernel events, hence the roofline object, are not available in it")
    return ap.parse_args()


def conv_flops(op, batch):
    return planner.op_flops(op, batch)


def cpu_baseline(model_cfg, size, n_images, score_threshold):
    """Oracle (oracle/: numpy + C restatement of the reference path) on the host cores, batch 1,
    FPS as inference.py:329-331 defines it (first image excluded)."""
    from oracle import darknet as od, decode as odec
    layers, out_index = od.parse_layers(model_cfg, config.num_classes)
    params = dw.random_params(layers, seed=0)
    anchors = inference.read_anchors(config.tiny_anchors_path if "tiny" in model_cfg else config.anchors_path)
    rng = np.random.default_rng(1234)
    times = []
    for i in range(n_images + 1):
        x = rng.random((1, size, size, 3), dtype=np.float32)
        t0 = time.time()
        heads = od.forward(layers, out_index, params, x)
        odec.predict(heads, anchors, config.num_classes, (size, size), (size, size), score_threshold,
                     config.nms_threshold, config.max_boxes)
        times.append(time.time() - t0)
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    return {"value": round(n_images / sum(times[1:]), 3), "unit": "images/sec", "cores": int(cores), "kind": "port",
            "sample": "%d images of %dx%d, batch 1, after 1 warm-up image; numpy(BLAS)+C oracle, not TensorFlow"
                      % (n_images, size, size)}


def _closed(name):
    """A kernel name as the plan carries it (template arguments cut short) with its bracket closed."""
    return name + ">" * max(0, name.count("<") - name.count(">"))


def spawn_ranks(n):
    """`python bench.py --gpus N` started without a launcher: start N fresh child ranks under torch.distributed.run
    (one process per GPU, RCCL), relay rank 0's JSON line and exit with the launcher's code.  This process never
    touches a GPU (torch.cuda.device_count() does not initialise one), and nothing is re-executed in place."""
    import socket
    import subprocess
    rehearsal = os.environ.get("YOLO_AMD_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if not rehearsal and have < n:
        sys.exit("bench.py: --gpus %d but only %d GPU(s) visible" % (n, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a one-GPU box (YOLO_AMD_BENCH_REHEARSAL=1): every rank uses device 0 and the
    # collectives run over gloo; the numbers mean nothing, the code path (barriers, max-over-ranks, gather) is real.
    rehearsal = os.environ.get("YOLO_AMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one process per GPU: this rank's kernels, arena and events all live on its own device (checked before any launch)
    assert torch.cuda.current_device() == local_rank, "rank %d bound to cuda:%d, expected cuda:%d" % (
        rank, torch.cuda.current_device(), local_rank)
    assert rehearsal or world <= torch.cuda.device_count(), "%d ranks but %d visible GPUs" % (world, torch.cuda.device_count())

    cfg = config.yolov3_tiny_cfg_path if args.model == "yolov3-tiny" else config.yolov3_cfg_path
    anchors = inference.read_anchors(config.tiny_anchors_path if args.model == "yolov3-tiny" else config.anchors_path)
    config.score_threshold = args.score_threshold
    config.max_boxes = args.max_boxes
    B, S = args.batch, args.size

    nodes = build_model.yolo(None, False, cfg, config.num_classes, dtype=args.dtype, device=dev, splitk=args.splitk)
    net = nodes.net
    net.set_params(dw.random_params(net.layers, seed=0))          # every rank: same seeded weights
    lo, hi = sharding.shard_range(world * B, rank, world)         # this rank's images of the global batch
    gen = torch.Generator(device="cpu").manual_seed(100 + rank)
    images = torch.rand((hi - lo, S, S, 3), generator=gen, dtype=torch.float32).to(dev)
    chunk = args.chunk or None

    graphed = None
    if args.graph:
        from yolo_amd.graph import GraphedDetector
        args.no_kernel_events = True
        graphed = GraphedDetector(net, anchors, config.num_classes, (S, S), batch=hi - lo, score_threshold=config.score_threshold,
                                  iou_threshold=config.nms_threshold, max_boxes=config.max_boxes)
        graphed.images.copy_(images)
        graphed.capture((S, S))

    pp = inference._postprocessor(anchors, config.num_classes, (S, S))

    def step(trace=None, trace_filter=None):
        # backbone -> decode -> NMS -> pack.  PostProcessor.detect runs each scale's decode inside its head convolution's
        # epilogue when every head launch qualifies (bf16, 256 / 192-pixel tiles; --unfused-decode turns it off), else
        # the three fp32 heads are written, and y3_decode reads them back.
        if graphed is not None:
            return graphed()
        if args.unfused_decode:
            heads = net.forward(images, chunk=chunk, trace=trace, trace_filter=trace_filter)
            det = None
        full = trace is not None and trace_filter is None
        if full:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n0 = len(trace)
        if args.unfused_decode:
            if full:
                e0.record()
            det = inference.predict_batch(heads, anchors, config.num_classes, (S, S), (S, S))
            if full:
                e1.record()
                trace.append(({"kind": "post", "kernel": "decode+nms+pack"}, e0, e1))
            return det
        if full:
            e0.record()
        det, _ = pp.detect(net, images, (S, S), chunk=chunk, trace=trace, trace_filter=trace_filter)
        if full:                                      # full trace (warm-up only): what follows the last backbone launch
            e1.record()
            trace.append(({"kind": "post", "kernel": "nms+pack (decode is in the head convolutions)", "after": trace[-1][2] if len(trace) > n0 else e0},
                          e0, e1))
        return det

    # Warm-up.  Its last step is traced in full (un-timed) to find the kernel most of the step goes to and the
    # per-stage split; the timed steps then carry HIP events around that kernel's launches only, because an
    # event pair costs ~3 us of stream time and 150 of them per step would slow the measured step by 4 %.
    warm_trace = None
    for w in range(max(args.warmup, 1)):
        last = w == max(args.warmup, 1) - 1 and not args.no_kernel_events
        warm_trace = [] if last else None
        det = step(warm_trace)
    torch.cuda.synchronize()
    dom_kernel = None
    if warm_trace:
        acc = {}
        for op, e0, e1 in warm_trace:
            if op["kind"] in ("conv", "stem", "resblock"):
                acc[op["kernel"]] = acc.get(op["kernel"], 0.0) + e0.elapsed_time(e1)
        dom_kernel = max(acc, key=acc.get)
    only_dom = (lambda op: op.get("kernel") == dom_kernel) if dom_kernel else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    traces = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr = [] if (not args.no_kernel_events and i % max(args.event_every, 1) == 0) else None
        det = step(tr, only_dom)
        if tr is not None:
            traces.append(tr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    counts = sharding.gather_counts(det.count)                    # the one collective, off the timed path
    assert counts.numel() == world * B, "gathered %d counts, expected %d ranks x %d images" % (counts.numel(), world, B)
    total_images = world * B * args.steps
    value = total_images / elapsed

    if rank == 0:
        plan = net.plan_for(chunk or (hi - lo), S, S).plan
        gflop_img = planner.conv_gflops(net.layers, S, S)
        out = {
            "metric": "images/sec (whole node) YOLOv3-608 end-to-end incl. NMS; conv MFMA util %",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s %dx%d, batch %d per GPU, %s, seeded random-init Darknet weights, backbone + "
                                   "decode + per-class NMS (score>=%.3g, iou>%.3g, max %d/class)"
                                   % (args.model, S, S, B, args.dtype, config.score_threshold, config.nms_threshold,
                                      config.max_boxes),
                       "global_batch": world * B, "parallelism": "dp%d (image-sharded, no data-path collective)" % world,
                       "chunk": chunk or B, "launch": "hip graph replay" if args.graph else "eager", "conv_gflop_per_image": round(gflop_img, 3),
                       "mean_detections_per_image": round(float(counts.float().mean().item()), 2)},
            # what the collective layer saw (the driver's scaling run can check that RCCL really spanned N ranks)
            "distributed": {"backend": dist.get_backend() if world > 1 else "none",
                            "world": dist.get_world_size() if world > 1 else 1,
                            "ranks_reporting": counts.numel() // B, "device_of_rank0": torch.cuda.current_device()},
            "conv_mfma_util_pct_end_to_end": round(100.0 * gflop_img * value / world / 1e3 /
                                                   (BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS), 2),
        }
        if traces:
            nb = chunk or (hi - lo)
            per_kernel = {}          # rocprof kernel name -> [ms, flops, launches]
            conv_ms = other_ms = 0.0
            for tr in traces:
                for op, e0, e1 in tr:
                    ms = e0.elapsed_time(e1)
                    if op["kind"] in ("conv", "stem", "resblock"):
                        conv_ms += ms
                        k = per_kernel.setdefault(op["kernel"], [0.0, 0.0, 0])
                        k[0] += ms; k[1] += conv_flops(op, nb); k[2] += 1
                    else:
                        other_ms += ms
            peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
            dom = max(per_kernel, key=lambda k: per_kernel[k][0])          # the kernel most time goes to
            dom_ms, dom_flops, dom_n = per_kernel[dom]
            achieved = dom_flops / (dom_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(achieved / peak, 4), "traffic": None,
                               "kernel": _closed(dom), "launches_per_step": dom_n // len(traces),
                               "avg_launch_ms": round(dom_ms / dom_n, 4),
                               "avg_gflop_per_launch": round(dom_flops / dom_n / 1e9, 3),
                               "share_of_step": round(dom_ms / len(traces) / (1e3 * elapsed / args.steps), 3),
                               "timed_steps_with_events": len(traces)}
            try:            # HBM bytes per launch of that kernel, from the committed PMC passes (profiles/)
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                for name, byt in tj["hbm_bytes_per_launch"].items():      # listed by total time: the first match is the main variant
                    if name.startswith(dom):
                        out["roofline"]["traffic"] = byt
                        out["roofline"]["traffic_source"] = tj["source"]
                        break
            except (OSError, ValueError, KeyError):
                pass
            w_conv = sum(e0.elapsed_time(e1) for op, e0, e1 in warm_trace if op["kind"] in ("conv", "stem", "resblock"))
            w_post = sum((op["after"].elapsed_time(e1) if "after" in op else e0.elapsed_time(e1))
                         for op, e0, e1 in warm_trace if op["kind"] == "post")
            w_other = sum(e0.elapsed_time(e1) for op, e0, e1 in warm_trace if op["kind"] not in ("conv", "stem", "resblock", "post"))
            out["stage_ms_per_step"] = {"conv_all": round(w_conv, 3), "other_layers": round(w_other, 3),
                                        "decode_nms_pack": round(w_post, 3),
                                        "note": "from one fully traced warm-up step, whose ~150 event pairs add ~3 us "
                                                "of stream time each: the stages sum to more than ms_per_step"}
            wk = {}
            for op, e0, e1 in warm_trace:
                if op["kind"] in ("conv", "stem", "resblock"):
                    v = wk.setdefault(op["kernel"], [0.0, 0.0, 0])
                    v[0] += e0.elapsed_time(e1); v[1] += conv_flops(op, nb); v[2] += 1
            out["kernels"] = {_closed(k): {"ms_per_step": round(v[0], 3), "launches_per_step": v[2],
                                        "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1)} for k, v in wk.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, S, args.cpu_images, args.score_threshold)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
