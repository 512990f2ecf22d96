"""Fused decode -> per-class NMS -> pack pipeline (the device side of the reference's
predict(), inference.py:88-167) with its workspace.  Three launches per batch, no host
synchronisation; results stay on the GPU as padded per-image tensors.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def anchor_masks(num_output_layers):
    """inference.py:106-107."""
    return [[6, 7, 8], [3, 4, 5], [0, 1, 2]] if num_output_layers == 3 else [[3, 4, 5], [0, 1, 2]]


def letterbox_params(input_shape, image_shape):
    """fp32 scalars of correct_boxes (utils/utils.py:111-116): scale = min(input/image);
    new = round_half_even(image*scale); offset = input - new.  -> (scale, off_h, off_w)."""
    f = np.float32
    inp = np.array(input_shape, dtype=f)
    img = np.array(image_shape, dtype=f)
    if inp.shape != (2,) or img.shape != (2,) or not (np.all(np.isfinite(img)) and np.all(img > 0) and np.all(inp > 0)):
        # the reference divides by whatever it is fed (utils/utils.py:114) and returns inf / NaN boxes; an image has
        # a positive height and width, anything else is a caller's bug
        raise ValueError("letterbox_params: input_shape %r / image_shape %r must be positive (h, w)"
                         % (tuple(np.ravel(input_shape)), tuple(np.ravel(image_shape))))
    scale = np.min(inp / img).astype(f)
    new_shape = np.round(img * scale).astype(f)
    off = (inp - new_shape).astype(f)
    return f(scale), f(off[0]), f(off[1])


def _head_view(t):
    """-> (tensor, ld): a float32 cuda tensor [B,gh,gw,C] whose cells are ld floats apart."""
    if t.dtype != torch.float32:
        t = t.float()
    B, gh, gw, C = t.shape
    ld = t.stride(2)
    if t.stride(3) != 1 or ld < C or t.stride(1) != gw * ld or t.stride(0) != gh * gw * ld:
        t = t.contiguous()
        ld = C
    return t, ld


class Detections(object):
    """Padded per-image results of one batch (all on the GPU):
    boxes [B,cap,4] (y_min,x_min,y_max,x_max), scores [B,cap], classes [B,cap] int32,
    count [B] int32 -- rows [0,count[b]) are valid, class-major, score-descending in a class."""

    def __init__(self, boxes, scores, classes, count, kept_idx, kept_count, kept_rank, records, cand_count):
        self.boxes, self.scores, self.classes, self.count = boxes, scores, classes, count
        self.kept_idx, self.kept_count, self.kept_rank = kept_idx, kept_count, kept_rank
        self.records, self.cand_count = records, cand_count

    def image(self, b):
        """(boxes[M,4], scores[M], classes[M]) of image b -- synchronises (reads count)."""
        m = int(self.count[b].item())
        return self.boxes[b, :m], self.scores[b, :m], self.classes[b, :m]


class PostProcessor(object):
    def __init__(self, anchors, num_classes, input_shape, score_threshold=0.6, iou_threshold=0.3, max_boxes=51,
                 with_rank=False):
        self.anchors = np.asarray(anchors, dtype=np.float32).reshape(-1, 2)
        self.num_classes = int(num_classes)
        self.input_shape = (int(input_shape[0]), int(input_shape[1]))
        self.score_threshold = float(np.float32(score_threshold))
        self.iou_threshold = float(np.float32(iou_threshold))
        self.max_boxes = int(max_boxes)
        self.with_rank = with_rank
        self._ws = {}
        _lib.load()

    def _workspace(self, B, N, device):
        key = (B, N, str(device))
        ws = self._ws.get(key)
        if ws is None:
            C, K = self.num_classes, self.max_boxes
            i32 = dict(dtype=torch.int32, device=device)
            f32 = dict(dtype=torch.float32, device=device)
            ws = dict(records=torch.empty((B, N, 8), **f32), cand_idx=torch.empty((B, N), **i32),
                      counts=torch.zeros(B + B * C, **i32), kept_idx=torch.empty((B, C, K), **i32),
                      kept_count=torch.empty((B, C), **i32),
                      kept_rank=torch.empty((B, C, K), **i32) if self.with_rank else None,
                      out_boxes=torch.empty((B, C * K, 4), **f32), out_scores=torch.empty((B, C * K), **f32),
                      out_classes=torch.empty((B, C * K), **i32), out_count=torch.empty(B, **i32),
                      letterbox=torch.empty((B, 4), **f32),
                      nms_ws=torch.empty(int(_lib.load().y3_nms_workspace_ints(B, N)), **i32))
            self._ws[key] = ws
        return ws

    def describe(self, B, grids):
        """y3_decode_desc for a batch of B images and head grids [(gh, gw, ld), ...] (coarse first)."""
        L = len(grids)
        masks = anchor_masks(L)
        C = self.num_classes
        d = _lib.DecodeDesc()
        d.batch, d.num_scales, d.num_classes = B, L, C
        A = len(masks[0])
        d.num_anchors = A
        N = 0
        for l, (gh, gw, ld) in enumerate(grids):
            d.grid_h[l], d.grid_w[l], d.head_ld[l] = gh, gw, ld
            for a, ai in enumerate(masks[l]):
                d.anchors[l][a][0] = float(self.anchors[ai][0])
                d.anchors[l][a][1] = float(self.anchors[ai][1])
            # input_shape[::-1] // grid_shape[::-1]  (utils/utils.py:90)
            d.stride_x[l] = float(np.floor(np.float32(self.input_shape[1]) / np.float32(gw)))
            d.stride_y[l] = float(np.floor(np.float32(self.input_shape[0]) / np.float32(gh)))
            N += gh * gw * A
        d.score_threshold = self.score_threshold
        d.boxes_per_image = N
        return d

    def detect(self, net, images, image_shapes, chunk=None, want_heads=False, trace=None, trace_filter=None):
        """The whole step on a build_model.Network: backbone -> decode -> NMS -> pack, with the decode of each scale
        fused into its head convolution's epilogue where the plan allows (bf16, launches on the 256 / 192-pixel tiles:
        the fp32 heads are then never read back, and not even written unless want_heads), else backbone + y3_decode.
        Same records, same detections either way (tests/test_gpu_e2e.py).  Returns (Detections, heads or None)."""
        images = images if isinstance(images, torch.Tensor) else torch.as_tensor(np.asarray(images), dtype=torch.float32)
        B, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        step = B if not chunk else max(1, min(int(chunk), B))
        hds = net.plan_for(step, H, W).plan.heads
        grids = [(h["h"], h["w"], h["ld"]) for h in hds]
        d = self.describe(B, grids)
        N, L = d.boxes_per_image, len(grids)
        ws = self._workspace(B, N, net.device)
        self.set_image_shapes(ws, B, image_shapes)
        ws["counts"].zero_()
        dec = dict(pp=self, ws=ws, store_heads=bool(want_heads))
        heads = net.forward(images, chunk=chunk, trace=trace, trace_filter=trace_filter, decode=dec)
        if dec["done"]:
            return self._finish(d, ws, (B, N, L), net.device), heads
        views = [_head_view(h) for h in heads]
        return self.launch(views, d, ws, (B, N, L)), heads

    def prepare(self, heads, image_shapes):
        """Host side of a call: validates the head tensors, builds the decode descriptor, makes sure the
        workspace exists and uploads the per-image letterbox scalars when the image shapes changed.  Nothing
        here may run under HIP graph capture (allocation, host -> device copy); `launch` is what a graph holds."""
        views = [_head_view(h) for h in heads]
        B = views[0][0].shape[0]
        device = views[0][0].device
        L = len(views)
        A = len(anchor_masks(L)[0])
        for l, (t, ld) in enumerate(views):
            if t.shape[3] != A * (5 + self.num_classes):
                raise ValueError("head %d has %d channels, expected %d" % (l, t.shape[3], A * (5 + self.num_classes)))
        d = self.describe(B, [(t.shape[1], t.shape[2], ld) for t, ld in views])
        N = d.boxes_per_image
        ws = self._workspace(B, N, device)
        self.set_image_shapes(ws, B, image_shapes)
        return views, d, ws, (B, N, L)

    def set_image_shapes(self, ws, B, image_shapes):
        """(h, w) for all images or a [B,2] array (inference.py:218) -> letterbox scalars on the device."""
        shapes = np.asarray(image_shapes, dtype=np.float32)
        if shapes.ndim == 1:
            shapes = np.broadcast_to(shapes, (B, 2))
        lb_key = shapes.tobytes()
        if ws.get("letterbox_key") != lb_key:        # re-upload only when the image shapes change
            lb = np.zeros((B, 4), dtype=np.float32)
            for b in range(B):
                lb[b, :3] = letterbox_params(self.input_shape, shapes[b])
            ws["letterbox"].copy_(torch.from_numpy(lb), non_blocking=False)
            ws["letterbox_key"] = lb_key

    def launch(self, views, d, ws, dims):
        """Device side: counter reset, decode, NMS (4 launches), pack -- asynchronous on the current stream,
        no allocation, no host copy: capturable in a HIP graph."""
        lib = _lib.load()
        B, N, L = dims
        C = self.num_classes
        device = views[0][0].device
        ws["counts"].zero_()
        stream = torch.cuda.current_stream(device).cuda_stream
        hp = [v[0].data_ptr() for v in views] + [None] * (3 - L)
        cand_count = ws["counts"][:B]
        class_count = ws["counts"][B:]
        _lib.check(lib.y3_decode(ctypes.byref(d), hp[0], hp[1], hp[2], ws["letterbox"].data_ptr(),
                                 ws["records"].data_ptr(), ws["cand_idx"].data_ptr(), cand_count.data_ptr(),
                                 class_count.data_ptr(), stream))
        self._keep = views
        return self._finish(d, ws, dims, device)

    def _finish(self, d, ws, dims, device):
        """NMS + pack on the decoded records of ws (from y3_decode or from the head convolutions' fused decode)."""
        lib = _lib.load()
        B, N, L = dims
        C = self.num_classes
        stream = torch.cuda.current_stream(device).cuda_stream
        cand_count = ws["counts"][:B]
        class_count = ws["counts"][B:]
        rank_ptr = ws["kept_rank"].data_ptr() if self.with_rank else None
        _lib.check(lib.y3_nms_bucketed(ws["records"].data_ptr(), ws["cand_idx"].data_ptr(), cand_count.data_ptr(),
                                       class_count.data_ptr(), B, N, C, self.max_boxes, self.iou_threshold,
                                       ws["nms_ws"].data_ptr(), ws["kept_idx"].data_ptr(), rank_ptr,
                                       ws["kept_count"].data_ptr(), stream))
        _lib.check(lib.y3_pack_outputs(ws["records"].data_ptr(), ws["kept_idx"].data_ptr(), ws["kept_count"].data_ptr(),
                                       B, N, C, self.max_boxes, ws["out_boxes"].data_ptr(), ws["out_scores"].data_ptr(),
                                       ws["out_classes"].data_ptr(), ws["out_count"].data_ptr(), stream))
        return Detections(ws["out_boxes"], ws["out_scores"], ws["out_classes"], ws["out_count"], ws["kept_idx"],
                          ws["kept_count"], ws["kept_rank"], ws["records"], cand_count)

    def __call__(self, heads, image_shapes):
        """heads: list of 2 or 3 head tensors [B,g,g,A*(5+C)] (coarse first);
        image_shapes: (h, w) for all images or a [B,2] array (inference.py:218)."""
        return self.launch(*self.prepare(heads, image_shapes))
