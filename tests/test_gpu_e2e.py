"""-m gpu: whole hot path, cfg -> heads -> decode -> NMS, against the oracle.

Tolerances (BASELINE.json north_star): head tensors <= 1e-3 max-abs in fp32 mode and
<= 2e-2 relative (max-abs error / max-abs reference, per head tensor) in bf16 mode; NMS
bit-exact given the HIP-decoded boxes.  PARITY UNPINNED w.r.t. TensorFlow itself (see
oracle/__init__.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import darknet as od
from oracle import decode as odec
from yolo_amd import build_model, config, inference
from yolo_amd import darknet_weights as dw

pytestmark = pytest.mark.gpu


def _net(cfg, dtype, seed, tmp_path=None):
    layers, out_index = od.parse_layers(cfg, 80)
    params = dw.random_params(layers, seed=seed)
    nodes = build_model.yolo(None, False, cfg, 80, dtype=dtype)
    if tmp_path is not None:           # through the Darknet file format, like the reference
        path = os.path.join(str(tmp_path), "w.weights")
        dw.write_weights(path, nodes.net.layers, params)
        build_model.load_weights(nodes, path)
    else:
        nodes.net.set_params(params)
    return nodes, layers, out_index, params


def _dense_from_records(rec, C=80):
    m = rec[:, 5:8].copy().view(np.uint32)
    scores = np.zeros((rec.shape[0], C), dtype=np.float32)
    for c in range(C):
        scores[:, c] = np.where((m[:, c >> 5] >> np.uint32(c & 31)) & 1, rec[:, 4], 0)
    return rec[:, :4], scores


def bf16_error_figures(h, r):
    """Error figures of one bf16-mode head tensor against an oracle tensor.  `norm` is the north_star's "2e-2
    relative" as this suite reads it (max-abs error / max-abs reference); the other two are element-wise, so that
    the leniency of the normalised reading is on record: `elem_max` = max |err| / max(|ref|, 1) and
    `frac_over` = share of elements with |err| > 2e-2 * max(|ref|, 1).  Measured: the
    element-wise reading does NOT hold end to end (bf16 storage through 75 layers: elem_max ~5e-2, 2-4 percent of
    the elements over 2e-2), neither against the fp32 oracle nor against its bf16-storage model; launch by launch
    (tests/test_gpu_layerwise.py) every kernel is within one bf16 ulp of that model."""
    h = np.asarray(h, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    err = np.abs(h - r)
    den = np.maximum(np.abs(r), 1.0)
    return {"norm": float(err.max() / np.abs(r).max()), "elem_max": float((err / den).max()),
            "frac_over": float((err > 2e-2 * den).mean())}


def _check_nms_exact(det, b, thr):
    boxes, scores = _dense_from_records(det.records[b].cpu().numpy())
    ob, os_, oc = odec.select(boxes, scores, thr, config.nms_threshold, config.max_boxes)
    m = int(det.count[b].item())
    assert m == len(oc)
    np.testing.assert_array_equal(det.classes[b, :m].cpu().numpy(), oc)
    np.testing.assert_array_equal(det.boxes[b, :m].cpu().numpy(), ob)
    np.testing.assert_array_equal(det.scores[b, :m].cpu().numpy(), os_)
    return m


def test_config2_tiny416_b32_fp32(tmp_path):
    """BASELINE config 2: YOLOv3-tiny 416 batch 32 fp32: conv + decode + NMS parity."""
    nodes, layers, oi, params = _net(config.yolov3_tiny_cfg_path, "fp32", 1, tmp_path)
    x = np.random.default_rng(1).random((32, 416, 416, 3), dtype=np.float32)
    heads = nodes(torch.from_numpy(x))
    torch.cuda.synchronize()
    ref = od.forward(layers, oi, params, x)
    for h, r in zip(heads, ref):
        assert tuple(h.shape) == r.shape
        assert np.abs(h.cpu().numpy() - r).max() <= 1e-3
    anchors = inference.read_anchors(config.tiny_anchors_path)
    det = inference.predict_batch(heads, anchors, 80, (416, 416), (416, 416))
    total = sum(_check_nms_exact(det, b, config.score_threshold) for b in range(32))
    assert total > 0
    # decode of the HIP heads vs decode of the oracle heads (fp32 tolerance)
    boxes, scores = odec.pooled_boxes_and_scores([r[:1] for r in ref], anchors, 80, (416, 416), (416, 416))
    rec0 = det.records[0].cpu().numpy()
    np.testing.assert_allclose(rec0[:, :4], boxes, rtol=1e-3, atol=0.5)
    np.testing.assert_allclose(rec0[:, 4], scores.max(axis=1), rtol=0, atol=2e-3)


def test_reference_style_predict_batch1():
    nodes, layers, oi, params = _net(config.yolov3_tiny_cfg_path, "fp32", 2)
    x = np.random.default_rng(2).random((1, 416, 416, 3), dtype=np.float32)
    heads = nodes(torch.from_numpy(x))
    anchors = inference.read_anchors(config.tiny_anchors_path)
    boxes, scores, classes = inference.predict(heads, anchors, 80, (416, 416), [375, 500])
    assert boxes.dtype == torch.float32 and classes.dtype == torch.int32 and boxes.shape[1] == 4
    assert boxes.shape[0] == scores.shape[0] == classes.shape[0] > 0
    assert torch.all(classes[1:] >= classes[:-1])
    lines = inference.label_lines(boxes.cpu().numpy(), scores.cpu().numpy(), classes.cpu().numpy(), (375, 500))
    want = odec.label_lines(boxes.cpu().numpy(), scores.cpu().numpy(), classes.cpu().numpy(), (375, 500))
    assert lines == want


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_yolov3_416_heads(dtype):
    """Darknet-53 416 (config 3 at batch 2 so the oracle finishes in seconds)."""
    nodes, layers, oi, params = _net(config.yolov3_cfg_path, dtype, 3)
    x = np.random.default_rng(3).random((2, 416, 416, 3), dtype=np.float32)
    heads = nodes(torch.from_numpy(x))
    torch.cuda.synchronize()
    ref = od.forward(layers, oi, params, x)
    for h, r in zip(heads, ref):
        err = np.abs(h.cpu().numpy() - r).max()
        if dtype == "fp32":
            assert err <= 1e-3
        else:
            assert err / np.abs(r).max() <= 2e-2
            print("yolov3-416 bf16 head %s vs fp32 oracle: %s" % (tuple(r.shape), bf16_error_figures(h.cpu().numpy(), r)))
    if dtype == "bf16":
        # against the oracle's bf16-storage model the figures are the same size: end to end the two runs carry
        # independent realisations of the same rounding noise (tests/test_gpu_layerwise.py is the launch-by-launch
        # check that the kernels themselves are within one bf16 ulp of that model)
        for h, r in zip(heads, od.forward_bf16_model(layers, oi, params, x)):
            fig = bf16_error_figures(h.cpu().numpy(), r)
            print("yolov3-416 bf16 head %s vs bf16-storage oracle: %s" % (tuple(r.shape), fig))
            assert fig["norm"] <= 2e-2
    anchors = inference.read_anchors(config.anchors_path)
    det = inference.predict_batch(heads, anchors, 80, (416, 416), (416, 416))
    for b in range(2):
        assert _check_nms_exact(det, b, config.score_threshold) > 0


def test_unfused_plan_equals_fused_plan():
    """Fallback kernels (upsample / add) give the same heads as the fused epilogues."""
    layers, oi = od.parse_layers(config.yolov3_cfg_path, 80)
    params = dw.random_params(layers, seed=4)
    x = torch.from_numpy(np.random.default_rng(4).random((1, 224, 224, 3), dtype=np.float32))
    outs = []
    for fuse in (True, False):
        net = build_model.Network(config.yolov3_cfg_path, 80, dtype="fp32", fuse=fuse)
        net.set_params(params)
        outs.append([h.cpu().numpy() for h in net(x)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)          # same arithmetic, same order: bit-identical


def test_full_size_properties_608_bf16():
    """BASELINE full size (608, bf16, batch 16 here): size-independent properties --
    an image's result does not depend on its batch position or on chunking, and the
    per-image NMS output equals the oracle's NMS of the same decoded boxes."""
    nodes, layers, oi, params = _net(config.yolov3_cfg_path, "bf16", 5)
    rng = np.random.default_rng(5)
    x = rng.random((16, 608, 608, 3), dtype=np.float32)
    x[9] = x[0]
    xt = torch.from_numpy(x).cuda()
    heads = [h.clone() for h in nodes(xt)]
    heads_chunked = [h.clone() for h in nodes(xt, chunk=5)]
    for a, b in zip(heads, heads_chunked):
        assert torch.equal(a, b)
        assert torch.equal(a[0], a[9])
    assert [tuple(h.shape) for h in heads] == [(16, 19, 19, 255), (16, 38, 38, 255), (16, 76, 76, 255)]
    # batch 16 picks the big tiles (256x256, 192x256, 256x128 ring, 256x64), batch 1 the small ones: every variant sums
    # k in the same order, so the same image gives the same bits -- and batch 1 is checked against the oracle
    one = [h.clone() for h in nodes(xt[:1])]
    ref = od.forward(layers, oi, params, x[:1])
    for a, o, r in zip(heads, one, ref):
        assert torch.equal(a[:1], o)
        assert np.abs(o.cpu().numpy() - r).max() / np.abs(r).max() <= 2e-2
    anchors = inference.read_anchors(config.anchors_path)
    det = inference.predict_batch(heads, anchors, 80, (608, 608), (608, 608))
    m0 = _check_nms_exact(det, 0, config.score_threshold)
    m9 = _check_nms_exact(det, 9, config.score_threshold)
    assert m0 == m9 > 0 and torch.equal(det.boxes[0, :m0], det.boxes[9, :m9])


def test_headline_shape_608_bf16_batch64():
    """The bench configuration itself (YOLOv3-608, bf16, 64 images: a 2.27 GB arena, buffer offsets near 2^31 in
    the register epilogue).  Image 0 is repeated at position 63 and image 1 at 37: heads and detections must be bit-equal
    between the copies and equal to the batch-1 runs, whose heads are checked against the oracle (normalised and
    element-wise bf16 figures); NMS of images 0, 37 and 63 equals the oracle's NMS of the same decoded boxes."""
    nodes, layers, oi, params = _net(config.yolov3_cfg_path, "bf16", 15)
    gen = torch.Generator(device="cpu").manual_seed(15)
    x = torch.rand((64, 608, 608, 3), generator=gen, dtype=torch.float32)
    x[63] = x[0]
    x[37] = x[1]
    xt = x.cuda()
    heads = [h.clone() for h in nodes(xt)]
    assert [tuple(h.shape) for h in heads] == [(64, 19, 19, 255), (64, 38, 38, 255), (64, 76, 76, 255)]
    for h in heads:
        assert torch.equal(h[0], h[63]) and torch.equal(h[1], h[37])
        assert bool(torch.isfinite(h).all())
    anchors = inference.read_anchors(config.anchors_path)
    det = inference.predict_batch(heads, anchors, 80, (608, 608), (608, 608))
    counts = det.count.cpu().numpy().copy()
    det_rows = {b: tuple(t.clone() for t in det.image(b)) for b in (0, 1, 37, 63)}
    for b in (0, 37, 63):
        assert _check_nms_exact(det, b, config.score_threshold) == counts[b]
    assert counts[0] == counts[63] > 0 and counts[1] == counts[37]
    for a, b in ((0, 63), (1, 37)):
        for u, v in zip(det_rows[a], det_rows[b]):
            assert torch.equal(u, v)
    for b in (0, 1):                                   # the same image alone: same bits (heads and detections)
        one = [h.clone() for h in nodes(xt[b:b + 1])]
        for h, o in zip(heads, one):
            assert torch.equal(h[b:b + 1], o)
        d1 = inference.predict_batch(one, anchors, 80, (608, 608), (608, 608))
        for u, v in zip(d1.image(0), det_rows[b]):
            assert torch.equal(u, v)
    ref = od.forward(layers, oi, params, x[:1].numpy())
    for h, r in zip(heads, ref):
        fig = bf16_error_figures(h[:1].cpu().numpy(), r)
        print("yolov3-608 b64 bf16 head %s vs fp32 oracle: %s" % (tuple(r.shape), fig))
        assert fig["norm"] <= 2e-2
    for h, r in zip(heads, od.forward_bf16_model(layers, oi, params, x[:1].numpy())):
        fig = bf16_error_figures(h[:1].cpu().numpy(), r)
        print("yolov3-608 b64 bf16 head %s vs bf16-storage oracle: %s" % (tuple(r.shape), fig))
        assert fig["norm"] <= 2e-2


def test_dense_stress_config5():
    """BASELINE config 5: 608, low score threshold -> ~10k+ candidates per image into NMS."""
    rng = np.random.default_rng(7)
    B = 2
    heads = []
    for g in (19, 38, 76):
        t = np.empty((B, g, g, 3, 85), dtype=np.float32)
        t[..., 0:2] = rng.standard_normal((B, g, g, 3, 2))
        t[..., 2:4] = rng.standard_normal((B, g, g, 3, 2)) * 0.5
        t[..., 4] = rng.standard_normal((B, g, g, 3)) * 2
        t[..., 5:] = rng.standard_normal((B, g, g, 3, 80)) * 2 - 2
        heads.append(torch.from_numpy(t.reshape(B, g, g, 255)).cuda())
    anchors = inference.read_anchors(config.anchors_path)
    old = config.score_threshold
    config.score_threshold = 0.001
    try:
        det = inference.predict_batch(heads, anchors, 80, (608, 608), (608, 608))
        counts = det.cand_count.cpu().numpy()
        assert np.all(counts > 8000)
        for b in range(B):
            m = _check_nms_exact(det, b, 0.001)
            assert m > 1000
    finally:
        config.score_threshold = old


def test_missing_weights_raise_and_nothing_is_fetched(tmp_path):
    nodes = build_model.yolo(None, False, config.yolov3_tiny_cfg_path, 80)
    with pytest.raises(FileNotFoundError):
        build_model.load_weights(nodes, str(tmp_path / "yolov3.weights"))
    with pytest.raises(RuntimeError):
        nodes(torch.zeros((1, 416, 416, 3)))
    with pytest.raises(ValueError):
        build_model.yolo(None, True, config.yolov3_tiny_cfg_path, 80)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_direct_first_layer_equals_padded_mfma_first_layer(dtype):
    """The first conv runs either on the direct kernel (fp32 image, Cin=3) or, by default, on the
    MFMA kernel over the zero-padded staged image: both must satisfy the head tolerance."""
    layers, oi = od.parse_layers(config.yolov3_tiny_cfg_path, 80)
    params = dw.random_params(layers, seed=6)
    x = np.random.default_rng(6).random((2, 416, 416, 3), dtype=np.float32)
    ref = od.forward(layers, oi, params, x)
    for pad in (True, False):
        net = build_model.Network(config.yolov3_tiny_cfg_path, 80, dtype=dtype, pad_input=pad)
        net.set_params(params)
        for h, r in zip(net(torch.from_numpy(x)), ref):
            err = np.abs(h.cpu().numpy() - r).max()
            assert err <= 1e-3 if dtype == "fp32" else err / np.abs(r).max() <= 2e-2


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
def test_run_inference_image_dir_contract(tmp_path, graph):
    """inference.py:170-331 end to end: a directory of images -> labels/<stem>.txt + images/<name>, through the
    GPU letterbox, the network, decode and NMS; checked against the oracle run on the same files (fp32 mode:
    coordinates may move by one pixel where floor(v + 0.5) sits on a boundary, scores within 2e-3)."""
    import argparse
    from PIL import Image
    from oracle import letterbox as olb
    rng = np.random.default_rng(5)
    img_dir, out_dir = tmp_path / "imgs", tmp_path / "out"
    img_dir.mkdir()
    shapes = [(120, 160), (200, 150), (96, 96)]
    for i, hw in enumerate(shapes):
        Image.fromarray(rng.integers(0, 256, hw + (3,), dtype=np.uint8), mode="RGB").save(str(img_dir / ("im%d.png" % i)))
    cfg = config.yolov3_tiny_cfg_path
    layers, oi = od.parse_layers(cfg, 80)
    params = dw.random_params(layers, seed=3)
    wpath = str(tmp_path / "tiny.weights")
    from yolo_amd import planner
    dw.write_weights(wpath, planner.parse_cfg(cfg, 80)[0], params)
    args = argparse.Namespace(cfg=cfg, weights=wpath, input_shape=416, dtype="fp32", darknet_model="1", graph=graph)
    results = inference.run_inference(str(img_dir), str(out_dir), args)
    anchors = inference.read_anchors(config.tiny_anchors_path)
    assert sorted(os.listdir(str(out_dir / "images"))) == ["im0.png", "im1.png", "im2.png"]
    found = 0
    for i, hw in enumerate(shapes):
        img = np.asarray(Image.open(str(img_dir / ("im%d.png" % i))).convert("RGB"))
        x = olb.network_input(img, (416, 416))
        heads = od.forward(layers, oi, params, x)
        ob, os_, oc = odec.predict(heads, anchors, 80, (416, 416), hw, config.score_threshold, config.nms_threshold,
                                   config.max_boxes)
        want = odec.label_lines(ob, os_, oc, hw)
        got = open(str(out_dir / "labels" / ("im%d.txt" % i))).read().splitlines()
        assert len(got) == len(want) == len(results[i][2])
        for g, w in zip(got, want):
            gv, wv = g.split(","), w.split(",")
            assert gv[4] == wv[4]
            assert all(abs(int(a) - int(b)) <= 1 for a, b in zip(gv[:4], wv[:4]))
            assert abs(float(gv[5]) - float(wv[5])) <= 2e-3
        found += len(got)
        drawn = np.asarray(Image.open(str(out_dir / "images" / ("im%d.png" % i))).convert("RGB"))
        assert drawn.shape == img.shape and (len(got) == 0) == bool((drawn == img).all())    # boxes are drawn (inference.py:314-321)
    assert found > 0


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_run_inference_batched_writes_the_same_files(tmp_path, dtype):
    """`--batch N` runs the directory in chunks of N images per step (last chunk shorter; with --graph it replays the
    full-batch graph): every label file and every annotated image is byte-identical to the one-image-per-step run
    (--no_splitk: split-K adds partial sums in another order), because a batch is B independent batch-1 runs and the
    batched plan gives an image the same bits at every batch size."""
    import argparse
    import filecmp
    from PIL import Image
    from yolo_amd import planner
    rng = np.random.default_rng(8)
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    shapes = [(120, 160), (200, 150), (96, 96), (64, 300), (333, 222)]
    for i, hw in enumerate(shapes):
        Image.fromarray(rng.integers(0, 256, hw + (3,), dtype=np.uint8), mode="RGB").save(str(img_dir / ("im%d.png" % i)))
    cfg = config.yolov3_tiny_cfg_path
    layers, _ = od.parse_layers(cfg, 80)
    wpath = str(tmp_path / "tiny.weights")
    dw.write_weights(wpath, planner.parse_cfg(cfg, 80)[0], dw.random_params(layers, seed=3))
    runs = {"b1": dict(batch=1, graph=False), "b2": dict(batch=2, graph=False), "b4g": dict(batch=4, graph=True),
            "b8": dict(batch=8, graph=False)}
    found = {}
    for name, kw in runs.items():
        args = argparse.Namespace(cfg=cfg, weights=wpath, input_shape=416, dtype=dtype, darknet_model="1", no_splitk=True, **kw)
        res = inference.run_inference(str(img_dir), str(tmp_path / name), args)
        found[name] = [len(r[2]) for r in res]
    assert sum(found["b1"]) > 0
    for name in ("b2", "b4g", "b8"):
        assert found[name] == found["b1"]
        for sub in ("labels", "images"):
            files = sorted(os.listdir(str(tmp_path / "b1" / sub)))
            assert len(files) == len(shapes) and files == sorted(os.listdir(str(tmp_path / name / sub)))
            match, mismatch, errors = filecmp.cmpfiles(str(tmp_path / "b1" / sub), str(tmp_path / name / sub), files, shallow=False)
            assert not mismatch and not errors, (name, sub, mismatch, errors)


@pytest.mark.parametrize("size,batch,chunk,thr", [(608, 24, None, 0.6), (416, 120, 64, 0.3), (608, 32, None, 0.001)])
def test_head_decode_fused_into_the_head_convolutions(size, batch, chunk, thr):
    """PostProcessor.detect runs each scale's decode inside its head convolution's epilogue (y3_conv2d_head_decode) when
    every head launch qualifies: the records, the candidate SET, the class counts, the heads (when asked for) and the
    final detections are bit-identical to backbone + y3_decode, chunked batches and the dense-stress threshold
    included; without want_heads the head tensors are not produced at all.  Small launches fall back."""
    from yolo_amd.postprocess import PostProcessor
    cfg = config.yolov3_cfg_path
    layers, _ = od.parse_layers(cfg, 80)
    net = build_model.Network(cfg, 80, dtype="bf16")
    net.set_params(dw.random_params(layers, seed=21))
    anchors = inference.read_anchors(config.anchors_path)
    x = torch.rand((batch, size, size, 3), dtype=torch.float32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(size + batch))
    shapes = np.stack([np.array([300 + 7 * i, 500 - 3 * i]) for i in range(batch)])      # every image has a positive size
    assert shapes.min() > 0
    plain = PostProcessor(anchors, 80, (size, size), thr, 0.3, 51)
    heads = net.forward(x, chunk=chunk)
    want = plain(heads, shapes)
    w_rec, w_cnt = want.records.clone(), plain._workspace(batch, want.records.shape[1], x.device)["counts"].clone()
    w_cand = plain._workspace(batch, want.records.shape[1], x.device)["cand_idx"].clone()
    w_out = [t.clone() for t in (want.boxes, want.scores, want.classes, want.count)]
    fused = PostProcessor(anchors, 80, (size, size), thr, 0.3, 51)
    for want_heads in (True, False):
        dec = dict(pp=fused, ws=None, store_heads=want_heads)
        got, gh = fused.detect(net, x, shapes, chunk=chunk, want_heads=want_heads)
        step = batch if not chunk else chunk
        assert all(net.plan_for(nb, size, size).decode_launchers(fused)[1] is not None for nb in {step, batch % step or step}), "fused path not taken"
        ws = fused._workspace(batch, got.records.shape[1], x.device)
        torch.testing.assert_close(got.records, w_rec, rtol=0, atol=0)
        torch.testing.assert_close(ws["counts"], w_cnt, rtol=0, atol=0)
        cc = w_cnt[:batch].cpu().numpy()
        a, b = ws["cand_idx"].cpu().numpy(), w_cand.cpu().numpy()
        for i in range(batch):
            assert sorted(a[i, :cc[i]]) == sorted(b[i, :cc[i]])
        torch.testing.assert_close(got.count, w_out[3], rtol=0, atol=0)
        for i, m in enumerate(w_out[3].cpu().numpy()):             # (the padded tails are uninitialised memory)
            for g, w in zip((got.boxes, got.scores, got.classes), w_out[:3]):
                torch.testing.assert_close(g[i, :m], w[i, :m], rtol=0, atol=0)
        # the bench path's OWN output against the oracle (not only against its un-fused twin): NMS + pack of detect()'s
        # records equal the oracle's select() on the same decoded boxes, bit for bit ...
        for b in sorted({0, batch // 2, batch - 1}):
            _check_nms_exact(got, b, thr)
        if want_heads:
            for g, w in zip(gh, heads):
                torch.testing.assert_close(g, w, rtol=0, atol=0)
            # ... and its records of image 0 equal the oracle's decode (utils/utils.py:43-160) of the head values
            # the same launch wrote (fp32 arithmetic: boxes rtol 1e-5, scores 2e-6 as in tests/test_gpu_decode.py)
            ob, osc = odec.pooled_boxes_and_scores([g[:1, ..., :255].cpu().numpy() for g in gh], anchors, 80, (size, size),
                                                   tuple(int(v) for v in shapes[0]))
            rec0 = got.records[0].cpu().numpy()
            np.testing.assert_allclose(rec0[:, :4], ob, rtol=1e-5, atol=1e-3)
            np.testing.assert_allclose(rec0[:, 4], osc.max(axis=1), rtol=0, atol=2e-6)
        else:
            assert gh is None
    assert int(w_out[3].sum()) > 0
    # a launch too small for the big tiles: detect falls back to y3_decode, same detections
    small = x[:2]
    d1, _ = fused.detect(net, small, shapes[:2])
    assert net.plan_for(2, size, size).decode_launchers(fused)[1] is None
    b1, c1 = d1.boxes.clone(), d1.count.clone()
    d2 = plain(net.forward(small), shapes[:2])
    torch.testing.assert_close(c1, d2.count, rtol=0, atol=0)
    for i, m in enumerate(c1.cpu().numpy()):
        torch.testing.assert_close(b1[i, :m], d2.boxes[i, :m], rtol=0, atol=0)


@pytest.mark.gpu
def test_frozen_graph_named_nodes():
    """predict_from_pb.py:221-251 against the node-name shim: fetching scale_1..3 by name with input_image fed
    returns exactly what the network's own call returns, and predict() on them gives the same detections."""
    from yolo_amd import frozen_graph
    from yolo_amd.build_model import Network
    cfg = config.yolov3_cfg_path
    net = Network(cfg, 80, dtype="fp32")
    net.set_params(dw.random_params(od.parse_layers(cfg, 80)[0], seed=11))
    x = np.random.default_rng(4).random((1, 96, 96, 3), dtype=np.float32)
    g = frozen_graph.FrozenGraph(net)
    nodes = [g.get_tensor_by_name("scale_%d:0" % k) for k in (1, 2, 3)]
    got = [t.clone() for t in g.run(nodes, {g.get_tensor_by_name("input_image:0"): x, "input_shape:0": (96, 96)})]
    want = net(torch.from_numpy(x))
    layers, oi = od.parse_layers(cfg, 80)
    ref = od.forward(layers, oi, dw.random_params(layers, seed=11), x)           # the oracle, not the shim's own back end
    for a, b, r in zip(got, want, ref):
        assert a.shape == b.shape and torch.equal(a[..., :255], b[..., :255])
        assert tuple(a[..., :255].shape) == r.shape and np.abs(a[..., :255].cpu().numpy() - r).max() <= 1e-3
    assert [t.shape[1] for t in got] == [3, 6, 12]
    one = g.run("scale_2:0", {"input_image:0": x[0]})              # [S,S,3] gets its batch dimension
    assert torch.equal(one[..., :255], want[1][..., :255])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_yolov3_spp_heads(dtype):
    """cfg stem 'yolov3-spp': max-pool 5/9/13 stride 1 + four-way zero-copy concat, against the oracle."""
    nodes, layers, oi, params = _net(config.yolov3_spp_cfg_path, dtype, 9)
    x = np.random.default_rng(9).random((2, 160, 160, 3), dtype=np.float32)
    heads = nodes(torch.from_numpy(x))
    torch.cuda.synchronize()
    ref = od.forward(layers, oi, params, x)
    for h, r in zip(heads, ref):
        err = np.abs(h.cpu().numpy() - r).max()
        assert err <= (1e-3 if dtype == "fp32" else 2e-2 * np.abs(r).max())


def _synthetic_heads(rng, B, grids, obj_mean):
    heads = []
    for g in grids:
        t = np.empty((B, g, g, 3, 85), dtype=np.float32)
        t[..., 0:2] = rng.standard_normal((B, g, g, 3, 2))
        t[..., 2:4] = rng.standard_normal((B, g, g, 3, 2)) * 0.5
        t[..., 4] = rng.standard_normal((B, g, g, 3)) * 2 + obj_mean
        t[..., 5:] = rng.standard_normal((B, g, g, 3, 80)) * 2 - 2
        heads.append(t.reshape(B, g, g, 255))
    return heads


def test_nothing_above_threshold_gives_empty_results(tmp_path):
    """Edge case: no box reaches score_threshold (the reference's boolean_mask is then empty for every class,
    inference.py:143-152): empty [0,4] / [0] / [0] tensors of the right dtypes, an empty label file, and an
    image written back unchanged."""
    heads = [torch.from_numpy(h).cuda() for h in _synthetic_heads(np.random.default_rng(8), 2, (13, 26), -30.0)]
    anchors = inference.read_anchors(config.tiny_anchors_path)
    res = inference.predict(heads, anchors, 80, (416, 416), [300, 400])
    assert len(res) == 2
    for boxes, scores, classes in res:
        assert tuple(boxes.shape) == (0, 4) and tuple(scores.shape) == (0,) and tuple(classes.shape) == (0,)
        assert boxes.dtype == torch.float32 and scores.dtype == torch.float32 and classes.dtype == torch.int32
    det = inference.predict_batch(heads, anchors, 80, (416, 416), [300, 400])
    assert det.count.cpu().tolist() == [0, 0] and det.cand_count.cpu().tolist() == [0, 0]
    assert inference.label_lines(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32), (300, 400)) == []
    img = np.random.default_rng(8).integers(0, 256, (30, 40, 3), dtype=np.uint8)
    drawn = inference.draw_detections(img, np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32),
                                      ["x"] * 80)
    np.testing.assert_array_equal(np.asarray(drawn), img)


def test_ragged_image_shapes_equal_batch1_runs():
    """SURVEY 8a batch note: a batch of B images with B different source shapes returns B triples, each equal
    (bit for bit: same kernels, per-image letterbox scalars) to the batch-1 run of that image, and to the oracle's
    NMS of the decoded boxes."""
    rng = np.random.default_rng(9)
    hn = _synthetic_heads(rng, 3, (19, 38, 76), 0.0)
    shapes = np.array([[480, 640], [608, 608], [1080, 1920]], dtype=np.int32)
    anchors = inference.read_anchors(config.anchors_path)
    heads = [torch.from_numpy(h).cuda() for h in hn]
    got = inference.predict(heads, anchors, 80, (608, 608), shapes)
    assert len(got) == 3
    total = 0
    for b in range(3):
        one = inference.predict([torch.from_numpy(h[b:b + 1]).cuda() for h in hn], anchors, 80, (608, 608), shapes[b])
        for a, c in zip(got[b], one):
            assert torch.equal(a, c)
        ob, os_, oc = odec.predict([h[b:b + 1] for h in hn], anchors, 80, (608, 608), tuple(int(v) for v in shapes[b]),
                                   config.score_threshold, config.nms_threshold, config.max_boxes)
        assert len(oc) == got[b][2].shape[0]
        np.testing.assert_array_equal(got[b][2].cpu().numpy(), oc)
        np.testing.assert_allclose(got[b][0].cpu().numpy(), ob, rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(got[b][1].cpu().numpy(), os_, rtol=0, atol=2e-6)
        total += len(oc)
    assert total > 100


def test_snake_tile_order_is_only_a_schedule():
    """Y3_CONV_REVERSE (consecutive convolutions walk their tiles in opposite directions) changes which workgroup
    computes which tile, nothing else: heads are bit-identical with and without it."""
    layers, oi = od.parse_layers(config.yolov3_cfg_path, 80)
    params = dw.random_params(layers, seed=6)
    x = torch.from_numpy(np.random.default_rng(6).random((3, 320, 320, 3), dtype=np.float32))
    outs = []
    for snake in (True, False):
        net = build_model.Network(config.yolov3_cfg_path, 80, dtype="bf16", snake=snake)
        net.set_params(params)
        plan = net.plan_for(3, 320, 320).plan
        assert any(op.get("reverse") for op in plan.ops) == snake
        outs.append([h.cpu().numpy() for h in net(x)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_balanced_split_in_the_plan_is_only_a_schedule():
    """Network(balanced=True): the launches that leave a partial round of 256 x 256 tiles go through y3_conv2d_fused_ws
    (one slab buffer for the plan, one block of flags per launch, zeroed once per pass) -- heads bit-identical to the
    default plan's, on two passes in a row (the flags of the first must not leak into the second)."""
    layers, oi = od.parse_layers(config.yolov3_cfg_path, 80)
    params = dw.random_params(layers, seed=8)
    B = 22          # 497 tiles of 256 x 256 at 76 x 76: 1.94 rounds on 256 workgroups, the library's choice there
    x = torch.from_numpy(np.random.default_rng(8).random((B, 608, 608, 3), dtype=np.float32))
    outs = []
    for balanced in (False, True, True):
        if len(outs) < 2:
            net = build_model.Network(config.yolov3_cfg_path, 80, dtype="bf16", balanced=balanced)
            net.set_params(params)
        plan = net.plan_for(B, 608, 608).plan
        assert any(op.get("balanced") for op in plan.ops) == balanced
        outs.append([h.cpu().numpy() for h in net(x)])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


def test_graphed_detector_equals_eager_path():
    """graph.GraphedDetector: one HIP graph per step (backbone + decode + NMS + pack), replayed on a static input
    buffer.  Same kernels as the eager path -> identical detections, also after the input and the source image
    shape change between replays."""
    from yolo_amd.graph import GraphedDetector
    nodes, layers, oi, params = _net(config.yolov3_tiny_cfg_path, "fp32", 12)
    anchors = inference.read_anchors(config.tiny_anchors_path)
    old = config.score_threshold
    config.score_threshold = 0.05           # random-init heads: make sure there is something to compare
    try:
        _graphed_vs_eager(GraphedDetector, nodes, anchors)
    finally:
        config.score_threshold = old


def _graphed_vs_eager(GraphedDetector, nodes, anchors):
    gd = GraphedDetector(nodes.net, anchors, 80, (416, 416), batch=2, score_threshold=config.score_threshold,
                         iou_threshold=config.nms_threshold, max_boxes=config.max_boxes)
    rng = np.random.default_rng(12)
    found = 0
    for shapes in ([375, 500], [[416, 416], [720, 1280]], [375, 500]):
        x = torch.from_numpy(rng.random((2, 416, 416, 3), dtype=np.float32)).cuda()
        gd.images.copy_(x)
        det = gd(shapes)
        counts = det.count.cpu().tolist()
        want = inference.predict(nodes(x), anchors, 80, (416, 416), shapes)
        for b in range(2):
            m = counts[b]
            assert m == want[b][2].shape[0]
            assert torch.equal(det.boxes[b, :m], want[b][0]) and torch.equal(det.scores[b, :m], want[b][1])
            assert torch.equal(det.classes[b, :m], want[b][2])
            found += m
        bx, sc, cl = gd.image(1)
        assert bx.shape == (counts[1], 4) and cl.dtype == np.int32
    assert found > 0 and gd.graph is not None


@pytest.mark.parametrize("dtype,size", [("fp32", 224), ("bf16", 608)])
def test_splitk_small_batch_mode(dtype, size):
    """Network(splitk=True): at batch 1 the launches with few output tiles and a long k are split over k
    (y3_conv2d_splitk: fp32 partial sums, fixed-order finish).  Same tolerance against the oracle as the
    un-split path, results reproducible run to run, and close to the un-split result."""
    layers, oi = od.parse_layers(config.yolov3_cfg_path, 80)
    params = dw.random_params(layers, seed=21)
    x = np.random.default_rng(21).random((1, size, size, 3), dtype=np.float32)
    xt = torch.from_numpy(x).cuda()
    outs = {}
    for sk in (True, False):
        net = build_model.Network(config.yolov3_cfg_path, 80, dtype=dtype, splitk=sk)
        net.set_params(params)
        plan = net.plan_for(1, size, size).plan
        n_split = sum(1 for op in plan.ops if op.get("splits", 1) > 1)
        assert (n_split > 10) == sk
        a = [h.clone() for h in net(xt)]
        b = [h.clone() for h in net(xt)]
        for u, v in zip(a, b):
            assert torch.equal(u, v)                 # deterministic: partial sums are added in a fixed order
        outs[sk] = [h.cpu().numpy() for h in a]
    ref = od.forward(layers, oi, params, x)
    for s_, u_, r in zip(outs[True], outs[False], ref):
        if dtype == "fp32":
            assert np.abs(s_ - r).max() <= 1e-3 and np.abs(s_ - u_).max() <= 1e-4
        else:
            assert np.abs(s_ - r).max() / np.abs(r).max() <= 2e-2
            assert np.abs(s_ - u_).max() / np.abs(r).max() <= 1e-2


def test_config1_against_committed_fixture():
    """BASELINE config 1 (YOLOv3-tiny 416, batch 1, fp32) against the committed fixture tests/golden/tiny416_b1.npz
    (oracle output; generating script beside it): heads within 1e-3, and the detections -- class, box within half a
    pixel, score within 2e-3 -- for at least 98 % of the fixture's boxes (a 1e-3 head difference may flip a
    near-threshold box or an IoU sitting on 0.3)."""
    import importlib.util
    gold = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gold, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = np.load(os.path.join(gold, "tiny416_b1.npz"))
    layers, oi = od.parse_layers(config.yolov3_tiny_cfg_path, 80)
    params = dw.random_params(layers, seed=1)
    x = np.random.default_rng(0).random((1, 416, 416, 3), dtype=np.float32)
    net = build_model.Network(config.yolov3_tiny_cfg_path, 80, dtype="fp32")
    net.set_params(params)
    heads = net(torch.from_numpy(x))
    for k, h in enumerate(heads):
        hn = h.cpu().numpy()
        assert list(hn.shape) == list(want["head%d_shape" % k])
        assert np.abs(hn.reshape(-1)[::mg.STRIDE] - want["head%d_sub" % k]).max() <= 1e-3
        assert abs(hn.astype(np.float64).sum() - want["head%d_sum" % k][0]) <= 1e-4 * want["head%d_sum" % k][1]
    anchors = inference.read_anchors(config.tiny_anchors_path)
    old = config.score_threshold
    config.score_threshold = mg.SCORE_THR
    try:
        boxes, scores, classes = (t.cpu().numpy() for t in inference.predict(heads, anchors, 80, (416, 416), (375, 500)))
    finally:
        config.score_threshold = old
    wb, ws_, wc = want["boxes"], want["scores"], want["classes"]
    assert abs(len(classes) - len(wc)) <= 3
    hit = 0
    for i in range(len(wc)):
        m = (classes == wc[i]) & (np.abs(boxes - wb[i]).max(axis=1) <= 0.5) & (np.abs(scores - ws_[i]) <= 2e-3)
        hit += bool(m.any())
    assert hit >= 0.98 * len(wc)
