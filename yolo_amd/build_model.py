"""Darknet cfg -> network, and the Darknet weight loader: the MI355X counterpart of the
reference's build_model.py (same public names and argument order):

    unique_config_sections(cfg_file)                                  build_model.py:12
    yolo(input_images, is_training, config_path, num_classes)         build_model.py:36
    load_weights(var_list, weights_file)                              build_model.py:273

The reference returns lazy TF tensors that `sess.run` evaluates (inference.py:277).
Here `yolo()` returns `OutputNodes`: a list-like of lazy head nodes bound to a `Network`;
calling it (`nodes(images)` / `nodes.run(images)`) launches the fused HIP plan and yields
the head tensors [B, g, g, 3*(num_classes+5)] (fp32, coarse -> fine), exactly the tensors
the reference's `yolo()` hands to `predict()`.

torch is used for device memory and streams only; all arithmetic is in libyolo3hip.so.
"""
import ctypes
import itertools
import os
import weakref

import numpy as np
import torch

from . import _lib, config, planner
from . import darknet_weights as dw
from .planner import unique_config_sections  # noqa: F401  (re-exported: reference API)

_NETWORKS = weakref.WeakSet()   # what tf.global_variables() is to the reference's load_weights call: the networks that
                                # are still alive (a dropped Network frees its device weights and plan arenas)
_ORDER = weakref.WeakKeyDictionary()   # creation order, for a stable global_variables() listing


_COUNTER = itertools.count()


def global_variables():
    """Stand-in for tf.global_variables() in `load_weights(tf.global_variables(), path)`
    (inference.py:255): the networks built so far."""
    return sorted(_NETWORKS, key=lambda n: _ORDER.get(n, 0))


def _dtype_code(dtype):
    return _lib.Y3_BF16 if dtype == "bf16" else _lib.Y3_F32


def pack_conv(kernel, bias, dtype, in_f32, device):
    """Folded HWIO kernel + bias -> (weights, bias, family, k_pad, n_pad) in the layout
    y3_conv_weight_layout prescribes (include/yolo3hip.h)."""
    kh, kw, ci, co = kernel.shape
    fam, k_pad, n_pad = _lib.conv_weight_layout(_dtype_code(dtype), ci, co, kh, in_f32=in_f32)
    bias_p = np.zeros(n_pad, dtype=np.float32)
    bias_p[:co] = bias
    if fam == 0:      # implicit GEMM: [n_pad][k_pad], k = (kh*ks + kw)*Cin + c
        wp = np.zeros((n_pad, k_pad), dtype=np.float32)
        wp[:co, :kh * kw * ci] = np.transpose(kernel, (3, 0, 1, 2)).reshape(co, -1)
        wt = torch.from_numpy(wp)
        if dtype == "bf16":
            wt = wt.to(torch.bfloat16)
    else:             # direct: fp32 [k][n_pad]
        wp = np.zeros((kh * kw * ci, n_pad), dtype=np.float32)
        wp[:, :co] = kernel.reshape(-1, co)
        wt = torch.from_numpy(wp)
    return wt.to(device).contiguous(), torch.from_numpy(bias_p).to(device), fam, k_pad, n_pad


class _CompiledPlan(object):
    """A planner.Plan bound to device memory: an arena, and one prepared launcher per op."""

    def __init__(self, net, plan):
        self.plan = plan
        self.arena = torch.empty(max(plan.arena_bytes, 256), dtype=torch.uint8, device=net.device)
        self._splitk_ws = None          # one fp32 partial-sum buffer shared by the plan's split-K launches (they run in order)
        need = 0
        if net.splitk:
            for op in plan.ops:
                if op["kind"] == "conv":
                    need = max(need, self._splitk_bytes(net, op))
        if need:
            self._splitk_ws = torch.empty(need, dtype=torch.uint8, device=net.device)
        # y3_conv2d_fused_ws (the 256 x 256-tile kernel's balanced split): fp32 slabs shared by the plan's launches (they
        # run in order on one stream) and one block of flags per launch, all zeroed by ONE memset at the start of a pass
        self._ws_slab_bytes, self._ws_ops, self._ws_slabs, self._ws_flags = 0, 0, None, None
        self._ws_ptrs = [0, 0]          # slabs, flags (read by the launch closures at launch time)
        self.head_ops = {}              # head index -> what a fused head-decode launch of that op needs (_prepare)
        self._decode = {}               # PostProcessor id -> (DecodeDesc for this plan's batch, fused launchers or None)
        self.launchers = [self._prepare(net, op) for op in plan.ops]
        if self._ws_ops:
            self._ws_slabs = torch.empty(self._ws_slab_bytes, dtype=torch.uint8, device=net.device)
            self._ws_flags = torch.zeros(self._ws_ops * 256, dtype=torch.int32, device=net.device)
            self._ws_ptrs[0], self._ws_ptrs[1] = self._ws_slabs.data_ptr(), self._ws_flags.data_ptr()

    def begin_pass(self):
        """Before the launches of one pass through the plan: the hand-off flags of y3_conv2d_fused_ws start at zero."""
        if self._ws_flags is not None:
            self._ws_flags.zero_()

    def decode_launchers(self, pp):
        """Launch list of this plan with the YOLO decode fused into the head convolutions (y3_conv2d_head_decode), for
        the thresholds / anchors of PostProcessor `pp`, or None when a head launch does not qualify (small batches,
        fp32 mode, 2-anchor heads ...): then the caller runs the plain plan and y3_decode.  Returns (desc, launchers)."""
        key = id(pp)
        hit = self._decode.get(key)
        if hit is None:
            lib = _lib.load()
            heads = self.plan.heads
            d = pp.describe(self.plan.batch, [(h["h"], h["w"], h["ld"]) for h in heads])
            ok = len(self.head_ops) == len(heads) and all(
                lib.y3_conv_head_decode_ok(self.head_ops[k]["dref"], ctypes.byref(d), k) for k in range(len(heads)))
            launchers = None
            if ok:
                launchers = list(self.launchers)
                fn = lib.y3_conv2d_head_decode
                for k, ho in self.head_ops.items():
                    def launch(env, _ho=ho, _k=k, _d=d, _dref=ctypes.byref(d)):
                        dec = env["decode"]
                        return fn(_ho["dref"], _ho["x"](env), _ho["w"], _ho["b"], _ho["y"](env) if dec["store_heads"] else None,
                                  _ho["z"], _dref, _k, dec["letterbox"], dec["records"], dec["cand_idx"], dec["cand_count"],
                                  dec["class_count"], env["stream"])
                    launchers[ho["index"]] = launch
            hit = (d, launchers, pp)      # (pp kept so that id(pp) stays unique while the entry lives)
            self._decode[key] = hit
        return hit[0], hit[1]

    def splitk_workspace(self, net, nbytes):
        assert self._splitk_ws is not None and self._splitk_ws.numel() >= nbytes
        return self._splitk_ws

    def _splitk_bytes(self, net, op):
        """Workspace y3_conv_splitk_plan asks for this conv op (0 = not split).  Needs the packed layout only."""
        lib = _lib.load()
        w, bias, fam, k_pad, n_pad = net._packed[op["layer"]]
        d = _lib.ConvDesc()
        d.dtype, d.batch = _dtype_code(self.plan.dtype), self.plan.batch
        d.in_h, d.in_w = op["in_hw"]
        d.in_c = op["in_c"]
        d.out_h, d.out_w = op["out_hw"]
        d.out_c, d.ksize, d.stride, d.k_pad, d.n_pad = op["out_c"], op["ksize"], op["stride"], k_pad, n_pad
        d.flags = _lib.CONV_IN_F32 if op["in_f32"] else 0
        splits, ws_bytes = ctypes.c_int32(1), ctypes.c_int64(0)
        _lib.check(lib.y3_conv_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(ws_bytes)))
        return int(ws_bytes.value) if splits.value > 1 else 0

    def _ptr(self, ref, es):
        """-> (kind, base_offset_bytes)"""
        buf = ref["buf"]
        if buf == "in":
            return ("in", ref["off"] * 4)
        if buf[0] == "arena":
            return ("arena", buf[1] + ref["off"] * es)
        return (buf, ref["off"] * 4)             # ('head', k), fp32

    def _prepare(self, net, op):
        lib = _lib.load()
        plan = self.plan
        es = 2 if plan.dtype == "bf16" else 4
        code = _dtype_code(plan.dtype)
        arena_base = self.arena.data_ptr()
        B = plan.batch

        def resolver(ref, esize=es):
            kind, off = self._ptr(ref, esize)
            if kind == "arena":
                p = arena_base + off
                return lambda env: p
            if kind == "in":
                return lambda env: env["in"] + off
            k = kind[1]
            return lambda env: env["heads"][k] + off

        kind = op["kind"]
        if kind == "conv":
            w, bias, fam, k_pad, n_pad = net._packed[op["layer"]]
            d = _lib.ConvDesc()
            d.dtype, d.batch = code, B
            d.in_h, d.in_w = op["in_hw"]
            d.in_c, d.in_ld = op["in_c"], op["x"]["ld"]
            d.out_h, d.out_w = op["out_hw"]
            d.out_c, d.out_ld = op["out_c"], op["y"]["ld"]
            d.res_ld = op["res"]["ld"] if op["res"] is not None else 0
            d.ksize, d.stride, d.k_pad, d.n_pad = op["ksize"], op["stride"], k_pad, n_pad
            flags = 0
            if op["leaky"]: flags |= _lib.CONV_LEAKY
            if op["res"] is not None: flags |= _lib.CONV_RESIDUAL
            if op["ups"]: flags |= _lib.CONV_UPS2X
            if op["out_f32"]: flags |= _lib.CONV_OUT_F32
            if op["in_f32"]: flags |= _lib.CONV_IN_F32
            if op.get("reverse"): flags |= _lib.CONV_REVERSE
            if op.get("pool"): flags |= _lib.CONV_POOL2
            d.flags = flags
            dref = ctypes.byref(d)
            name = ctypes.create_string_buffer(96)
            _lib.check(lib.y3_conv_kernel_name(dref, name, 96))
            op["kernel"] = name.value.decode()         # as rocprofv3 names it (benches, profiles)
            x = resolver(op["x"], 4 if op["in_f32"] else es)
            y = resolver(op["y"], 4 if op["out_f32"] else es)
            r = resolver(op["res"]) if op["res"] is not None else (lambda env: None)
            wp, bp, zp = w.data_ptr(), bias.data_ptr(), net._zeros.data_ptr()
            fn = lib.y3_conv2d_fused
            if net.splitk:              # small-batch mode: launches with few tiles and a long k are split over k
                splits, ws_bytes = ctypes.c_int32(1), ctypes.c_int64(0)
                _lib.check(lib.y3_conv_splitk_plan(dref, ctypes.byref(splits), ctypes.byref(ws_bytes)))
                if splits.value > 1:
                    op["splits"] = splits.value
                    ws = self.splitk_workspace(net, ws_bytes.value)
                    wsp, wsn, fk = ws.data_ptr(), ws.numel(), lib.y3_conv2d_splitk

                    def launch_split(env, _keep=(d, w, bias, ws)):
                        return fk(dref, x(env), wp, bp, r(env), y(env), zp, wsp, wsn, env["stream"])
                    return launch_split

            if op["out_f32"] and op["y"]["buf"][0] == "head" and not net.splitk:
                self.head_ops[op["y"]["buf"][1]] = dict(dref=dref, x=x, y=y, w=wp, b=bp, z=zp, index=plan.ops.index(op), keep=(d, w, bias))

            if net.balanced and not net.splitk:
                nbytes = ctypes.c_int64(0)
                _lib.check(lib.y3_conv_workspace_bytes(dref, ctypes.byref(nbytes)))
                if nbytes.value > 0:        # several rounds of 256 x 256 tiles with a partial last one: deal the k-steps evenly
                    k_ws, ptrs, nb_ws, fws = self._ws_ops, self._ws_ptrs, int(nbytes.value), lib.y3_conv2d_fused_ws
                    self._ws_ops += 1
                    self._ws_slab_bytes = max(self._ws_slab_bytes, nb_ws)
                    op["balanced"] = True
                    op["kernel"] = "conv_pp_split_kernel"

                    def launch_ws(env, _keep=(d, w, bias)):
                        return fws(dref, x(env), wp, bp, r(env), y(env), zp, ptrs[0], nb_ws, ptrs[1] + k_ws * 1024, env["stream"])
                    return launch_ws

            def launch(env, _keep=(d, w, bias)):
                return fn(dref, x(env), wp, bp, r(env), y(env), zp, env["stream"])
            return launch
        if kind == "resblock":
            i1, i3, _ = op["layers"]
            w1, b1, fam1, k1_pad, n1_pad = net._packed[i1]
            w2, b2, fam2, k2_pad, n2_pad = net._packed[i3]
            assert fam1 == 0 and fam2 == 0, "resblock: unexpected weight layouts"
            d = _lib.ResblockDesc()
            d.dtype, d.batch = code, B
            d.h, d.w = op["hw"]
            d.c, d.x_ld, d.out_ld = op["c"], op["x"]["ld"], op["y"]["ld"]
            d.k1_pad, d.n1_pad, d.k2_pad, d.n2_pad = k1_pad, n1_pad, k2_pad, n2_pad
            d.flags = (1 if op["leaky"][0] else 0) | (2 if op["leaky"][1] else 0) | (_lib.CONV_REVERSE if op.get("reverse") else 0)
            dref = ctypes.byref(d)
            tiles = ctypes.c_int64(0)
            _lib.check(lib.y3_resblock_tiles(dref, ctypes.byref(tiles)))
            op["kernel"] = {256: "resblock_kernel<128, 2, 4", 128: "resblock128_kernel", 64: "resblock64_kernel"}[op["c"]]
            assert tiles.value == op["tiles"], "planner and library disagree on the block's tiling"
            x, y = resolver(op["x"]), resolver(op["y"])
            ptrs = (w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), net._zeros.data_ptr())
            fn = lib.y3_resblock_fused

            def launch(env, _keep=(d, w1, b1, w2, b2)):
                return fn(dref, x(env), ptrs[0], ptrs[1], ptrs[2], ptrs[3], y(env), ptrs[4], env["stream"])
            return launch
        if kind == "stem":
            i0, i1, i2 = op["layers"]
            w0, b0, fam0, _, n0_pad = net._packed[i0]
            w1, b1, fam1, k1_pad, _ = net._packed[i1]
            assert fam0 == 1 and fam1 == 0, "stem: unexpected weight layouts"
            d = _lib.StemDesc()
            d.batch = B
            d.in_h, d.in_w = op["in_hw"]
            d.c0, d.c1, d.c2 = 32, 64, (32 if i2 is not None else 0)
            d.n0_pad, d.k1_pad, d.out1_ld = n0_pad, k1_pad, op["y"]["ld"]
            w2 = b2 = None
            y2 = lambda env: None
            if i2 is not None:
                w2, b2, fam2, k2_pad, _ = net._packed[i2]
                assert fam2 == 0
                d.k2_pad, d.out2_ld = k2_pad, op["y2"]["ld"]
                y2 = resolver(op["y2"])
            d.flags = sum(1 << j for j, lk in enumerate(op["leaky"]) if lk)
            dref = ctypes.byref(d)
            op["kernel"] = "conv_stem_kernel"
            x, y = resolver(op["x"], 4), resolver(op["y"])
            ptrs = (w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                    w2.data_ptr() if w2 is not None else None, b2.data_ptr() if b2 is not None else None)
            fn = lib.y3_conv_stem

            def launch(env, _keep=(d, w0, b0, w1, b1, w2, b2)):
                return fn(dref, x(env), ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5], y(env), y2(env), env["stream"])
            return launch
        if kind == "padin":
            x, y = resolver(op["x"], 4), resolver(op["y"])
            args = (op["pixels"], op["c"])
            return lambda env: lib.y3_pad_input(code, x(env), y(env), *args, env["stream"])
        if kind == "maxpool":
            x, y = resolver(op["x"]), resolver(op["y"])
            h, w_ = op["in_hw"]
            args = (B, h, w_, op["c"], op["x"]["ld"], op["y"]["ld"], op["size"], op["stride"])
            return lambda env: lib.y3_maxpool2d(code, x(env), y(env), *args, env["stream"])
        if kind == "upsample":
            x, y = resolver(op["x"]), resolver(op["y"])
            h, w_ = op["in_hw"]
            args = (B, h, w_, op["c"], op["x"]["ld"], op["y"]["ld"])
            return lambda env: lib.y3_upsample2x(code, x(env), y(env), *args, env["stream"])
        if kind == "copy":
            x, y = resolver(op["x"]), resolver(op["y"])
            args = (op["pixels"], op["c"], op["x"]["ld"], op["y"]["ld"])
            return lambda env: lib.y3_copy_channels(code, x(env), y(env), *args, env["stream"])
        if kind == "add":
            a, b, y = resolver(op["a"]), resolver(op["b"]), resolver(op["y"])
            args = (op["pixels"], op["c"], op["a"]["ld"], op["b"]["ld"], op["y"]["ld"])
            return lambda env: lib.y3_shortcut_add(code, a(env), b(env), y(env), *args, env["stream"])
        raise ValueError("unknown op kind %r" % kind)


class Network(object):
    """The compiled detector: cfg layers + packed device weights + cached launch plans."""

    def __init__(self, config_path, num_classes, dtype=None, device=None, fuse=True, pad_input=False, stem=True,
                 snake=None, splitk=False, resblock=None, balanced=None):
        self.config_path = config_path
        self.num_classes = num_classes
        self.dtype = dtype or config.compute_dtype
        if self.dtype not in ("bf16", "fp32"):
            raise ValueError("dtype must be 'bf16' or 'fp32'")
        self.fuse = fuse
        # pad_input: stage the 3-channel image zero-padded to one 16-byte chunk so the first conv runs on
        # the MFMA kernel.  Measured slower than the direct kernel at 608 (1.05 vs 0.87 ms), so off by default.
        self.pad_input = pad_input
        self.stem = stem            # fuse Darknet-53's first three convolutions into one launch (bf16 mode)
        # consecutive convolutions walk their tiles in opposite directions (cache reuse); YOLO_AMD_SNAKE=0 for A/B runs
        self.snake = (os.environ.get("YOLO_AMD_SNAKE", "1") != "0") if snake is None else bool(snake)
        # split-K for launches with few output tiles (small batches: latency mode).  Off by default: split sums are
        # added in another order, so results are no longer bit-identical across batch sizes / chunking.
        self.splitk = bool(splitk)
        # y3_conv2d_fused_ws for the launches that leave a partial round of 256 x 256 tiles: same bits, but measured 16 - 20 %
        # SLOWER per layer inside the network (profiles/r03l), so off unless asked for (YOLO_AMD_BALANCED=1: A/B)
        self.balanced = (os.environ.get("YOLO_AMD_BALANCED", "0") == "1") if balanced is None else bool(balanced)
        # residual blocks (1x1 -> 3x3 -> shortcut) as one launch: None = where the planner finds it pays (bf16, large
        # launches), False = never; YOLO_AMD_RESBLOCK=0 for A/B runs.  Same bits either way.
        if resblock is None and os.environ.get("YOLO_AMD_RESBLOCK") == "0":
            resblock = False
        self.resblock = resblock
        self.layers, self.out_index = planner.parse_cfg(config_path, num_classes, config.num_anchors_per_scale)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.params = None
        self._packed = None
        self._plans = {}
        self._zeros = None
        _lib.load()                 # fail now, loudly, if the HIP library is not built

    # ---- weights ---------------------------------------------------------------------------
    def set_params(self, params):
        """params: {layer: {'kernel' HWIO, 'gamma','beta','mean','var' | 'bias'}} (fp32 numpy)."""
        packed = {}
        first_conv = next(i for i, L in enumerate(self.layers) if L["type"] == "conv")
        for li, L in enumerate(self.layers):
            if L["type"] != "conv":
                continue
            kernel, bias = dw.fold_batch_norm(params[li])
            kh, kw, ci, co = kernel.shape
            assert (kh, ci, co) == (L["size"], L["cin"], L["filters"]), "layer %d: kernel shape %r" % (li, kernel.shape)
            epc = 8 if self.dtype == "bf16" else 4
            if li == first_conv and self.pad_input and ci < epc:
                kernel = np.concatenate([kernel, np.zeros((kh, kw, epc - ci, co), dtype=np.float32)], axis=2)
                packed[li] = pack_conv(kernel, bias, self.dtype, False, self.device)
            else:
                packed[li] = pack_conv(kernel, bias, self.dtype, li == first_conv, self.device)
        self.params = params
        self._packed = packed
        self._zeros = torch.zeros(256, dtype=torch.uint8, device=self.device)
        self._plans = {}

    def load_weights(self, weights_file):
        self.set_params(dw.read_weights(weights_file, self.layers))

    # ---- execution -------------------------------------------------------------------------
    def plan_for(self, batch, height, width):
        key = (batch, height, width)
        cp = self._plans.get(key)
        if cp is None:
            if self._packed is None:
                raise RuntimeError("weights are not loaded: call load_weights(net, path) first")
            plan = planner.build_plan(self.layers, self.out_index, batch, height, width, self.dtype, fuse=self.fuse,
                                      pad_input=self.pad_input, stem=self.stem, snake=self.snake,
                                      resblock=(None if self.dtype == "bf16" else False) if self.resblock is None else self.resblock)
            cp = _CompiledPlan(self, plan)
            self._plans[key] = cp
        return cp

    def forward(self, images, chunk=None, trace=None, trace_filter=None, decode=None):
        """images: float32 NHWC [B,H,W,3] on this network's device, values as the reference
        feeds them (image/255, inference.py:242).  Returns the head tensors (fp32,
        [B,g,g,3*(C+5)] views, coarse -> fine).  `chunk` splits the batch into sub-batches
        that run back to back through the whole plan.  `trace`, if a list, receives
        (op, start_event, end_event) for every launch (HIP events on the launch stream), or only for
        the launches trace_filter(op) selects (an event pair costs ~3 us of stream time).
        `decode` (PostProcessor.detect's side of the fused head decode): dict(pp=..., ws=..., store_heads=bool); when
        every head launch of the plan qualifies, the head convolutions decode their own scale into ws (records,
        candidates, counts; counts zeroed by the caller) and decode["done"] is set; the heads are returned only with
        store_heads (None otherwise).  When not, the plain plan runs and decode["done"] stays False."""
        if not isinstance(images, torch.Tensor):
            images = torch.as_tensor(np.asarray(images), dtype=torch.float32)
        if images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError("images must be NHWC [B,H,W,3], got %r" % (tuple(images.shape),))
        images = images.to(device=self.device, dtype=torch.float32).contiguous()
        B, H, W, _ = images.shape
        step = B if not chunk else max(1, min(int(chunk), B))
        first = self.plan_for(step, H, W)
        chunks = [(b0, min(step, B - b0)) for b0 in range(0, B, step)]
        plans = {nb: (first if nb == step else self.plan_for(nb, H, W)) for _, nb in chunks}
        fused = None
        if decode is not None:
            decode["done"] = False
            fused = {nb: cp.decode_launchers(decode["pp"]) for nb, cp in plans.items()}
            if any(v[1] is None for v in fused.values()):
                fused = None
        want_heads = fused is None or decode["store_heads"]
        heads = [torch.empty((B, hd["h"], hd["w"], hd["ld"]), dtype=torch.float32, device=self.device)
                 for hd in first.plan.heads] if want_heads else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        in_base = images.data_ptr()
        for b0, nb in chunks:
            cp = plans[nb]
            env = {"in": in_base + b0 * H * W * 3 * 4, "stream": stream,
                   "heads": [h.data_ptr() + b0 * h.stride(0) * 4 for h in heads] if want_heads else [0] * len(first.plan.heads)}
            launchers = cp.launchers
            cp.begin_pass()
            if fused is not None:
                ws, N, C = decode["ws"], fused[nb][0].boxes_per_image, fused[nb][0].num_classes
                env["decode"] = dict(store_heads=bool(decode["store_heads"]),
                                     letterbox=ws["letterbox"].data_ptr() + b0 * 16,
                                     records=ws["records"].data_ptr() + b0 * N * 32,
                                     cand_idx=ws["cand_idx"].data_ptr() + b0 * N * 4,
                                     cand_count=ws["counts"].data_ptr() + b0 * 4,
                                     class_count=ws["counts"].data_ptr() + (B + b0 * C) * 4)
                launchers = fused[nb][1]
            if trace is None:
                for launch in launchers:
                    rc = launch(env)
                    if rc:
                        _lib.check(rc)
            else:
                for op, launch in zip(cp.plan.ops, launchers):
                    if trace_filter is not None and not trace_filter(op):
                        rc = launch(env)
                        if rc:
                            _lib.check(rc)
                        continue
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = launch(env)
                    e1.record()
                    if rc:
                        _lib.check(rc)
                    trace.append((op, e0, e1))
        self._keepalive = images
        if fused is not None:
            decode["done"] = True
        if not want_heads:
            return None
        return [h[..., :hd["c"]] for h, hd in zip(heads, first.plan.heads)]

    __call__ = forward


class HeadNode(object):
    """Lazy stand-in for one output tensor of the reference's yolo()."""

    def __init__(self, net, index):
        self.net, self.index = net, index

    def __repr__(self):
        return "<HeadNode scale_%d of %s>" % (self.index + 1, self.net.config_path)


class OutputNodes(list):
    """What yolo() returns: list of lazy head nodes; call it to evaluate them."""

    def __init__(self, net):
        super(OutputNodes, self).__init__(HeadNode(net, k) for k in range(len(net.out_index)))
        self.net = net

    def run(self, images, chunk=None):
        return self.net.forward(images, chunk=chunk)

    __call__ = run


def yolo(input_images, is_training, config_path, num_classes, dtype=None, device=None, splitk=False):
    """Builds the network described by a Darknet cfg (reference build_model.py:36-270).
    input_images is accepted for signature parity and may be None (the reference passes a
    placeholder); is_training must be False (inference path only)."""
    if is_training:
        raise ValueError("yolo_amd implements the inference path only: is_training must be False")
    net = Network(config_path, num_classes, dtype=dtype, device=device, splitk=splitk)
    _NETWORKS.add(net)
    _ORDER[net] = next(_COUNTER)
    return OutputNodes(net)


def reset_default_graph():
    """tf.reset_default_graph() for this package: forget every network yolo() has built (they are freed once the
    caller drops its own references)."""
    _NETWORKS.clear()


class _LoadOp(object):
    """The reference returns tf.assign ops for sess.run; here the load has already happened."""

    def __init__(self, net, path):
        self.net, self.path = net, path

    def run(self):
        return None


def load_weights(var_list, weights_file):
    """Loads a Darknet .weights file (reference build_model.py:273-339).  var_list may be a
    Network, the OutputNodes of yolo(), or global_variables().  A missing file raises
    FileNotFoundError: nothing is fetched from the network (the reference shells out to wget)."""
    if isinstance(var_list, Network):
        nets = [var_list]
    elif isinstance(var_list, OutputNodes):
        nets = [var_list.net]
    else:
        nets = [v.net if isinstance(v, (OutputNodes, HeadNode)) else v for v in var_list]
    if not nets:
        raise ValueError("load_weights: no network to load into")
    ops = []
    for net in nets:
        print("loading weights from {} ...............".format(weights_file))
        net.load_weights(weights_file)
        ops.append(_LoadOp(net, weights_file))
    print("weights loaded successfully!!")
    return ops
