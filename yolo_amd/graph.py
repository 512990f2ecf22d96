"""One HIP graph for a whole step: backbone (73 launches for YOLOv3) + decode + NMS + pack.

The reference runs one image per `sess.run` (inference.py:243,277).  At batch 1 a step is ~80 kernel launches
of a few microseconds each, so the host's launch cost, not the GPU, sets the latency; captured once with
`torch.cuda.graph` (hipGraph underneath) and replayed per image, the step is a single submission.  The input
lives in a static buffer (`GraphedDetector.images`, which `utils.letterbox_gpu(..., out=...)` can fill in
place); results are the same padded `Detections` tensors the eager path returns -- same kernels, same bits.
"""
import numpy as np
import torch

from .postprocess import PostProcessor


class GraphedDetector(object):
    def __init__(self, net, anchors, num_classes, input_shape, batch=1, score_threshold=0.6, iou_threshold=0.3,
                 max_boxes=51):
        self.net = net
        self.batch = int(batch)
        self.input_shape = (int(input_shape[0]), int(input_shape[1]))
        self.pp = PostProcessor(anchors, num_classes, self.input_shape, score_threshold, iou_threshold, max_boxes)
        self.images = torch.zeros((self.batch,) + self.input_shape + (3,), dtype=torch.float32, device=net.device)
        self.graph = None
        self.heads = None
        self.det = None
        self._ws = None

    def _eager(self, image_shapes):
        # backbone -> decode -> NMS -> pack; the decode runs inside the head convolutions where the plan allows
        # (postprocess.PostProcessor.detect); the head tensors are kept (self.heads) as before
        return self.pp.detect(self.net, self.images, image_shapes, want_heads=True)

    def capture(self, image_shapes=None):
        """Warm up on a side stream (first launches set kernel attributes and allocate the workspace, neither
        of which may happen during capture), then record one step."""
        shapes = self.input_shape if image_shapes is None else image_shapes
        side = torch.cuda.Stream(device=self.net.device)
        side.wait_stream(torch.cuda.current_stream(self.net.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._eager(shapes)
        torch.cuda.current_stream(self.net.device).wait_stream(side)
        torch.cuda.synchronize(self.net.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            det, heads = self._eager(shapes)           # workspace exists, shapes unchanged: no allocation, no copy
        ws = self.pp._workspace(self.batch, int(det.records.shape[1]), self.net.device)
        self.graph, self.heads, self.det, self._ws = graph, heads, det, ws
        return self

    def __call__(self, image_shapes=None):
        """Replays the step on whatever `self.images` holds now.  image_shapes: (h, w) or [B,2] source sizes
        (uploaded before the replay when they changed).  Returns postprocess.Detections (static buffers: they
        are overwritten by the next call)."""
        if self.graph is None:
            self.capture(image_shapes)
        if image_shapes is not None:
            self.pp.set_image_shapes(self._ws, self.batch, image_shapes)
        self.graph.replay()
        return self.det

    def image(self, b=0):
        """(boxes [M,4], scores [M], classes [M]) of image b as numpy arrays -- synchronises."""
        m = int(self.det.count[b].item())
        return (self.det.boxes[b, :m].cpu().numpy(), self.det.scores[b, :m].cpu().numpy(),
                self.det.classes[b, :m].cpu().numpy())
