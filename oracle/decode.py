"""Oracle (test infrastructure): head decode + predict, numpy fp32.

Restates utils/utils.py:43-160 (get_head, correct_boxes, get_boxes_and_scores)
and inference.py:88-167 (predict) of the reference, op for op and in the same
order, in float32.  PARITY UNPINNED (see oracle/__init__.py).
"""
import numpy as np

from . import nms as _nms

F32 = np.float32


def _sigmoid(x):
    # tf.nn.sigmoid in fp32: 1 / (1 + exp(-x))
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def get_head(output, anchors, num_classes, input_shape, calc_loss=False):
    """utils/utils.py:43-96."""
    output = np.asarray(output, dtype=F32)
    anchors = np.asarray(anchors, dtype=F32)
    num_anchors = anchors.shape[0]
    anchors_tensor = anchors.reshape(1, 1, 1, num_anchors, 2)                       # :59-60
    gh, gw = output.shape[1:3]                                                      # :63
    mx, my = np.meshgrid(np.arange(gw, dtype=np.int32), np.arange(gh, dtype=np.int32))   # :68-70
    xy_offset = np.concatenate([mx.reshape(-1, 1), my.reshape(-1, 1)], axis=-1)     # :71-73
    xy_offset = xy_offset.reshape(gh, gw, 1, 2).astype(F32)                         # :74-75
    output = output.reshape(-1, gh, gw, num_anchors, num_classes + 5)               # :79
    box_xy = _sigmoid(output[..., :2])                                              # :84
    box_wh = np.exp(output[..., 2:4]).astype(F32)                                   # :85
    box_conf = _sigmoid(output[..., 4:5])                                           # :86
    box_cls = _sigmoid(output[..., 5:])                                             # :87
    # input_shape[::-1] // float(grid_shape[::-1]) : floor division, (w, h) order   # :90
    stride = np.floor(np.array(input_shape[::-1], dtype=F32) / np.array([gw, gh], dtype=F32)).astype(F32)
    box_xy = ((box_xy + xy_offset) * stride).astype(F32)                            # :90-91
    box_wh = (box_wh * anchors_tensor).astype(F32)                                  # :92
    if calc_loss:
        return xy_offset, output, box_xy, box_wh
    return box_xy, box_wh, box_conf, box_cls


def letterbox_params(input_shape, image_shape):
    """The scalars of correct_boxes (utils/utils.py:111-116) in fp32:
    scale, offset_h, offset_w.  tf.round == round-half-to-even == np.round."""
    inp = np.array(input_shape, dtype=F32)
    img = np.array(image_shape, dtype=F32)
    scale = np.min(inp / img).astype(F32)
    new_shape = np.round(img * scale).astype(F32)
    offset = (inp - new_shape).astype(F32)
    return scale, offset[0], offset[1]


def correct_boxes(box_xy, box_wh, input_shape, image_shape):
    """utils/utils.py:100-129 -> [..., (y_min, x_min, y_max, x_max)], no clipping."""
    box_xy = np.asarray(box_xy, dtype=F32)
    box_wh = np.asarray(box_wh, dtype=F32)
    scale, off_h, off_w = letterbox_params(input_shape, image_shape)
    mins = (box_xy - box_wh / F32(2.0)).astype(F32)                                 # :118
    maxes = (box_xy + box_wh / F32(2.0)).astype(F32)                                # :119
    boxes = np.concatenate([
        (mins[..., 1:2] - off_h / F32(2.0)) / scale,                                # :123
        (mins[..., 0:1] - off_w / F32(2.0)) / scale,                                # :124
        (maxes[..., 1:2] - off_h / F32(2.0)) / scale,                               # :125
        (maxes[..., 0:1] - off_w / F32(2.0)) / scale,                               # :126
    ], axis=-1).astype(F32)
    return boxes


def get_boxes_and_scores(output, anchors, num_classes, input_shape, image_shape):
    """utils/utils.py:132-160 -> boxes [N,4], box_scores [N,num_classes]."""
    xy, wh, conf, cls = get_head(output, anchors, num_classes, input_shape)
    boxes = correct_boxes(xy, wh, input_shape, image_shape).reshape(-1, 4)          # :147-148
    best = np.max(cls, axis=-1, keepdims=True)                                      # :150
    mask = (cls >= best).astype(F32)                                                # :151
    class_scores = (cls * mask).astype(F32)                                         # :153
    box_scores = (conf * class_scores).astype(F32)                                  # :154
    return boxes, box_scores.reshape(-1, num_classes)                               # :159


def anchor_masks(num_output_layers):
    """inference.py:106-107."""
    return [[6, 7, 8], [3, 4, 5], [0, 1, 2]] if num_output_layers == 3 else [[3, 4, 5], [0, 1, 2]]


def pooled_boxes_and_scores(output_nodes, anchors, num_classes, input_shape, image_shape):
    """inference.py:110-125: per-scale decode, concatenated coarse -> fine."""
    anchors = np.asarray(anchors, dtype=F32)
    masks = anchor_masks(len(output_nodes))
    boxes, scores = [], []
    for l, node in enumerate(output_nodes):
        b, s = get_boxes_and_scores(node, anchors[masks[l]], num_classes, input_shape, image_shape)
        boxes.append(b)
        scores.append(s)
    return np.concatenate(boxes, axis=0), np.concatenate(scores, axis=0)


def select(boxes, box_scores, score_threshold=0.6, iou_threshold=0.3, max_boxes=51, return_index=False):
    """inference.py:127-167: per-class order-preserving mask, greedy NMS, gather,
    class-major concat.  With return_index also returns, per kept row, the index
    into the class-masked array (tf nms_index) and into the pooled box array."""
    num_classes = box_scores.shape[1]
    mask = box_scores >= F32(score_threshold)                                       # :127
    out_b, out_s, out_c, out_rank, out_pool = [], [], [], [], []
    for c in range(num_classes):                                                    # :136
        sel = np.nonzero(mask[:, c])[0]                                             # boolean_mask keeps order
        cb = np.ascontiguousarray(boxes[sel])                                       # :143
        cs = np.ascontiguousarray(box_scores[sel, c])                               # :144
        keep = _nms.non_max_suppression(cb, cs, max_boxes, iou_threshold)           # :148-152
        out_b.append(cb[keep])                                                      # :153
        out_s.append(cs[keep])                                                      # :155
        out_c.append(np.full(len(keep), c, dtype=np.int32))                         # :157
        out_rank.append(keep.astype(np.int32))
        out_pool.append(sel[keep].astype(np.int32))
    res = (np.concatenate(out_b, axis=0).reshape(-1, 4).astype(F32),
           np.concatenate(out_s, axis=0).astype(F32),
           np.concatenate(out_c, axis=0))
    if return_index:
        return res + (np.concatenate(out_rank), np.concatenate(out_pool))
    return res


def predict(output_nodes, anchors, num_classes, input_shape, image_shape,
            score_threshold=0.6, iou_threshold=0.3, max_boxes=51):
    """inference.py:88-167 at batch 1 (the only batch size the reference runs,
    inference.py:243).  Thresholds default to config.py:3,35,36."""
    boxes, scores = pooled_boxes_and_scores(output_nodes, anchors, num_classes, input_shape, image_shape)
    return select(boxes, scores, score_threshold, iou_threshold, max_boxes)


def label_lines(out_boxes, out_scores, out_classes, image_hw):
    """inference.py:291-308: rows iterated in reverse; left,top,right,bottom,class,score
    with floor(v+0.5) and clipping to the image."""
    h, w = image_hw
    lines = []
    for i in reversed(range(len(out_classes))):
        top, left, bottom, right = out_boxes[i]
        top = max(0, int(np.floor(top + 0.5).astype(np.int32)))
        left = max(0, int(np.floor(left + 0.5).astype(np.int32)))
        bottom = min(h, int(np.floor(bottom + 0.5).astype(np.int32)))
        right = min(w, int(np.floor(right + 0.5).astype(np.int32)))
        lines.append("%d,%d,%d,%d,%d,%s" % (left, top, right, bottom, int(out_classes[i]), str(out_scores[i])))
    return lines
