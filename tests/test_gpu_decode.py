"""-m gpu: fused decode kernel and the dense helper kernels against the oracle's decode.
Floating point: boxes within 1e-5 relative (+1e-4 px absolute), scores within 2e-6 absolute --
the only differences are the last ulps of expf / division between the GPU and numpy.  Class
masks must be identical wherever the top-2 class probabilities are not within 1e-6."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import decode as odec
from yolo_amd import inference
from yolo_amd.postprocess import PostProcessor
from yolo_amd.utils import utils as yu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
COCO = np.array([(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)], dtype=np.float32)
TINY = np.array([(10, 14), (23, 27), (37, 58), (81, 82), (135, 169), (344, 319)], dtype=np.float32)


def _heads(rng, B, grids, scale=1.5):
    return [(rng.standard_normal((B, g, g, 255)) * scale).astype(np.float32) for g in grids]


def _mask_to_dense(rec, C=80):
    m = rec[..., 5:8].copy().view(np.uint32)
    out = np.zeros(rec.shape[:-1] + (C,), dtype=bool)
    for c in range(C):
        out[..., c] = ((m[..., c >> 5] >> np.uint32(c & 31)) & 1) != 0
    return out


def _oracle_class_mask(heads_b, anchors, size):
    """utils/utils.py:150-151 through the oracle: mask = (cls >= max_c cls) on its fp32 sigmoids, pooled coarse -> fine."""
    out = []
    for l, h in enumerate(heads_b):
        cls = odec.get_head(h, anchors[odec.anchor_masks(len(heads_b))[l]], 80, (size, size))[3]
        out.append((cls >= cls.max(axis=-1, keepdims=True)).reshape(-1, 80))
    return np.concatenate(out)


def _robust_zones(heads_b):
    """Where the class mask does not depend on the last ulp of expf: `must` = classes tied with the maximum for
    any correct sigmoid (same logit as the largest, or both saturated: t > 18.1 gives 1/(1+e) == 1.0f whatever the
    last bits of e), `mustnt` = classes whose float64 sigmoid is more than 1e-6 (relative) below the maximum."""
    t = np.concatenate([h.reshape(-1, 85)[:, 5:] for h in heads_b]).astype(np.float64)
    tmax = t.max(axis=1, keepdims=True)
    with np.errstate(over="ignore"):
        p = 1.0 / (1.0 + np.exp(-t))
    pmax = p.max(axis=1, keepdims=True)
    must = (t == tmax) | ((t > 18.1) & (tmax > 18.1)) | ((t < -104.0) & (tmax < -104.0))
    mustnt = p < pmax * (1.0 - 1e-6)
    return must, mustnt


@pytest.mark.parametrize("grids,anchors,size,image_shape,ld,scale", [
    ((13, 26, 52), COCO, 416, (416, 416), 255, 1.5),
    ((19, 38, 76), COCO, 608, (375, 500), 255, 1.5),
    ((13, 26), TINY, 416, (720, 1280), 255, 1.5),
    # cells 256 floats apart = what the network hands y3_decode: the lane-per-box LDS-DMA kernel, directly against the oracle
    ((19, 38, 76), COCO, 608, (375, 500), 256, 1.5),
    ((13, 26), TINY, 416, (720, 1280), 256, 1.5),
    ((13, 26, 52), COCO, 416, (300, 500), 256, 12.0),     # saturated sigmoids: many classes tie at 1.0
    ((13, 26, 52), COCO, 416, (300, 500), 255, 12.0),
])
def test_fused_decode_matches_oracle(grids, anchors, size, image_shape, ld, scale):
    rng = np.random.default_rng(size + len(grids) + ld + int(scale))
    B = 3
    full = [(rng.standard_normal((B, g, g, ld)) * scale).astype(np.float32) for g in grids]
    if scale > 10:
        full[0][0, 0, 0, 5:85] = -110.0                      # every class underflows: all tie at sigmoid == 0
        full[0][0, 0, 0, 4] = 3.0
        full[1][1, 2, 3, 90:170] = 30.0                      # every class saturates: all tie at 1
        full[1][1, 2, 3, 89] = 3.0
    heads = [f[..., :255] for f in full]
    pp = PostProcessor(anchors, 80, (size, size), score_threshold=0.3)
    dev_heads = [torch.from_numpy(f).cuda()[..., :255] for f in full]
    det = pp(dev_heads, image_shape)
    torch.cuda.synchronize()
    assert pp.prepare(dev_heads, image_shape)[1].head_ld[0] == ld          # the layout under test reached the library
    rec = det.records.cpu().numpy()
    cand_count = det.cand_count.cpu().numpy()
    loose = total = 0
    for b in range(B):
        hb = [h[b:b + 1] for h in heads]
        with np.errstate(over="ignore"):
            boxes, scores = odec.pooled_boxes_and_scores(hb, anchors, 80, (size, size), image_shape)
            want_mask = _oracle_class_mask(hb, anchors, size)
        np.testing.assert_allclose(rec[b, :, :4], boxes, rtol=1e-5, atol=1e-4)
        best = scores.max(axis=1)
        np.testing.assert_allclose(rec[b, :, 4], best, rtol=0, atol=2e-6)
        got_mask = _mask_to_dense(rec[b])
        must, mustnt = _robust_zones(hb)
        assert got_mask[must].all() and not got_mask[mustnt].any()
        np.testing.assert_array_equal(want_mask[must | mustnt], got_mask[must | mustnt])
        # elsewhere (sigmoids within 1e-6 of the largest but not provably tied) the two expf implementations may
        # round differently: counted, and required to be rare
        loose += int((want_mask != got_mask).sum())
        total += want_mask.size
        n_or = int((best >= np.float32(0.3)).sum())
        assert abs(int(cand_count[b]) - n_or) <= 2          # borderline scores may flip by an ulp
    print("decode ld=%d scale=%g: %d of %d class-mask bits differ from the oracle (near-ties only)" % (ld, scale, loose, total))
    assert loose <= 1e-4 * total
    if scale > 10:
        p0 = 0                                              # scale 0, cell (0,0), anchor 0: all 80 classes tie at 0
        assert _mask_to_dense(rec[0])[p0].all() and rec[0, p0, 4] == 0.0
        p1 = 13 * 13 * 3 + (2 * 26 + 3) * 3 + 1             # scale 1, cell (2,3), anchor 1: all tie at 1
        assert _mask_to_dense(rec[1])[p1].all() and rec[1, p1, 4] == np.float32(1.0) / (np.float32(1.0) + np.exp(np.float32(-3.0)))


def test_decode_known_answers_on_gpu():
    gold = json.load(open(os.path.join(GOLD, "decode_cases.json")))
    for case in gold["cases"]:
        g = case["grid"]
        out = torch.zeros((1, g, g, 255), dtype=torch.float32, device="cuda")
        anchors = np.array([case["anchor_wh"]] * 3, dtype=np.float32)
        boxes, scores = yu.get_boxes_and_scores(out, anchors, 80, tuple(case["input_shape"]), tuple(case["image_shape"]))
        gx, gy = case["cell_xy"]
        p = (gy * g + gx) * 3
        np.testing.assert_array_equal(boxes[p].cpu().numpy(), np.float32(case["box_yxyx"]))
        assert torch.all(scores[p] == case["score_every_class"])         # 80-way tie: every class keeps the score
        xy, wh, conf, cls = yu.get_head(out, anchors, 80, tuple(case["input_shape"]))
        np.testing.assert_array_equal(xy[0, gy, gx, 0].cpu().numpy(), np.float32(case["box_xy"]))
        np.testing.assert_array_equal(wh[0, gy, gx, 0].cpu().numpy(), np.float32(case["box_wh"]))
        assert float(conf[0, gy, gx, 0, 0]) == case["conf"] and torch.all(cls == case["cls"])


def test_dense_helpers_match_oracle():
    rng = np.random.default_rng(77)
    out = (rng.standard_normal((2, 13, 13, 255)) * 1.5).astype(np.float32)
    anchors = COCO[[6, 7, 8]]
    t = torch.from_numpy(out).cuda()
    xy, wh, conf, cls = yu.get_head(t, anchors, 80, (416, 416))
    oxy, owh, oconf, ocls = odec.get_head(out, anchors, 80, (416, 416))
    assert tuple(xy.shape) == oxy.shape and tuple(cls.shape) == ocls.shape and tuple(conf.shape) == oconf.shape
    np.testing.assert_allclose(xy.cpu().numpy(), oxy, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(wh.cpu().numpy(), owh, rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(conf.cpu().numpy(), oconf, rtol=0, atol=2e-7)
    np.testing.assert_allclose(cls.cpu().numpy(), ocls, rtol=0, atol=2e-7)
    boxes = yu.correct_boxes(torch.from_numpy(oxy).cuda(), torch.from_numpy(owh).cuda(), (416, 416), (300, 500))
    np.testing.assert_allclose(boxes.cpu().numpy(), odec.correct_boxes(oxy, owh, (416, 416), (300, 500)), rtol=1e-6, atol=1e-4)
    b, s = yu.get_boxes_and_scores(t, anchors, 80, (416, 416), (300, 500))
    ob, os_ = odec.get_boxes_and_scores(out, anchors, 80, (416, 416), (300, 500))
    assert tuple(b.shape) == ob.shape and tuple(s.shape) == os_.shape
    np.testing.assert_allclose(b.cpu().numpy(), ob, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(s.cpu().numpy(), os_, rtol=0, atol=2e-6)


def test_head_with_plain_contiguous_layout_and_bad_channels():
    out = torch.zeros((1, 13, 13, 255), device="cuda")
    pp = PostProcessor(COCO, 80, (416, 416))
    with pytest.raises(ValueError):
        pp([out[..., :250]], (416, 416))
    with pytest.raises(ValueError):
        inference.predict([out, out], TINY, 20, (416, 416), (416, 416))


@pytest.mark.parametrize("grids,anchors,size,scale", [((19, 38, 76), COCO, 608, 1.5), ((13, 26), TINY, 416, 1.5),
                                                      ((13, 26, 52), COCO, 416, 12.0), ((7, 9, 21), COCO, 224, 60.0)])
def test_lane_per_box_decode_equals_16_lane_decode(grids, anchors, size, scale):
    """The network hands y3_decode heads whose cells are 256 floats apart: that layout takes the lane-per-box
    kernel (LDS-DMA staging, exact sigmoid only near the class maximum).  It must reproduce the
    16-lanes-per-box kernel bit for bit -- records, candidate set and per-class counts -- including
    saturated sigmoids (scale 12: many classes tie at 1.0) and underflowing ones (scale 60)."""
    from yolo_amd import _lib
    rng = np.random.default_rng(int(size + scale))
    B = 2
    full = [torch.from_numpy((rng.standard_normal((B, g, g, 256)) * scale).astype(np.float32)).cuda() for g in grids]
    if scale > 50:
        full[0][0, 0, 0, 5:85] = -95.0                       # every class underflows: all tie at sigmoid = 0
        full[1][1, 2, 3, 90:170] = 30.0                      # every class saturates: all tie at 1
    heads = [t[..., :255] for t in full]
    outs = []
    with _lib.experimental() as lib:          # libyolo3hip_exp.so: the product library has no variant switch
        for variant in (1, 0):
            lib.y3_decode_variant(variant)
            pp = PostProcessor(anchors, 80, (size, size), score_threshold=0.3)
            det = pp(heads, (size - 7, size + 40))
            torch.cuda.synchronize()
            ws = list(pp._ws.values())[0]
            n = det.cand_count.cpu().numpy()
            cand = [np.sort(ws["cand_idx"][b, :n[b]].cpu().numpy()) for b in range(B)]
            outs.append((det.records.cpu().numpy().copy(), n.copy(), cand, ws["counts"].cpu().numpy().copy(),
                         det.kept_idx.cpu().numpy().copy(), det.kept_count.cpu().numpy().copy()))
    a, b_ = outs
    np.testing.assert_array_equal(a[0].view(np.uint32), b_[0].view(np.uint32))
    np.testing.assert_array_equal(a[1], b_[1])
    for x, y in zip(a[2], b_[2]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[3], b_[3])
    np.testing.assert_array_equal(a[5], b_[5])
