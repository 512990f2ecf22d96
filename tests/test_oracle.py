"""CPU tests that pin the oracle (the reference has no tests of its own, so the pins are
hand-derived known answers, an independent torch-CPU implementation, and Darknet's
published FLOP counts)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import darknet as od
from oracle import decode as odec
from oracle import nms as onms
from yolo_amd import config, planner
from yolo_amd import darknet_weights as dw

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = "/root/reference"


def test_flops_match_darknet():
    for cfg, size, want in ((config.yolov3_cfg_path, 608, 140.692), (config.yolov3_cfg_path, 416, 65.864),
                            (config.yolov3_cfg_path, 320, 38.973), (config.yolov3_tiny_cfg_path, 416, 5.565)):
        layers, _ = od.parse_layers(cfg, 80)
        assert abs(od.conv_gflops(layers, size, size) - want) < 5e-4
        mine, _ = planner.parse_cfg(cfg, 80)
        assert abs(planner.conv_gflops(mine, size, size) - want) < 5e-4


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted")
@pytest.mark.parametrize("name", ["yolov3", "yolov3-tiny"])
def test_packaged_cfg_equals_reference_cfg(name):
    mine, oi = od.parse_layers(os.path.join(os.path.dirname(config.yolov3_cfg_path), name + ".cfg"), 80)
    ref, oi_ref = od.parse_layers(os.path.join(REF, "darknet_data", name + ".cfg"), 80)
    assert oi == oi_ref and len(mine) == len(ref)
    for a, b in zip(mine, ref):
        assert a == b
    prod, oi_p = planner.parse_cfg(os.path.join(REF, "darknet_data", name + ".cfg"), 80)
    assert oi_p == oi_ref
    for a, b in zip(prod, ref):
        for k in ("type", "cin", "filters", "size", "stride", "bn", "activation", "ids", "src"):
            assert a.get(k) == b.get(k)


def test_unique_config_sections():
    s = od.unique_config_sections(config.yolov3_tiny_cfg_path).read()
    assert "[net_0]" in s and "[convolutional_12]" in s and "[yolo_1]" in s and "[route_1]" in s
    assert s == planner.unique_config_sections(config.yolov3_tiny_cfg_path).read()


def test_unknown_cfg_stem_raises_like_reference(tmp_path):
    p = tmp_path / "mynet.cfg"
    p.write_text(open(config.yolov3_tiny_cfg_path).read())
    with pytest.raises(NameError):
        od.parse_layers(str(p), 80)
    with pytest.raises(NameError):
        planner.parse_cfg(str(p), 80)
    with pytest.raises(AssertionError):
        planner.parse_cfg(str(tmp_path / "mynet.txt"), 80)


def _torch_conv(x, k, stride, same):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    kt = torch.from_numpy(k).permute(3, 2, 0, 1).double()
    if same and k.shape[0] == 3:
        xt = F.pad(xt, (1, 1, 1, 1))
    return F.conv2d(xt, kt, stride=stride).permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("ks,stride", [(3, 1), (1, 1), (3, 2)])
def test_conv_against_torch(ks, stride):
    rng = np.random.default_rng(ks * 10 + stride)
    x = rng.standard_normal((2, 10, 10, 5)).astype(np.float32)
    k = rng.standard_normal((ks, ks, 5, 7)).astype(np.float32)
    if stride == 2:     # build_model.py:151-154: pad top/left then VALID
        got = od.conv2d(np.pad(x, ((0, 0), (1, 0), (1, 0), (0, 0))), k, 2, False)
        want = _torch_conv(np.pad(x, ((0, 0), (1, 0), (1, 0), (0, 0))), k, 2, False)
        assert got.shape == (2, 5, 5, 7)
    else:
        got = od.conv2d(x, k, 1, True)
        want = _torch_conv(x, k, 1, True)
        assert got.shape == (2, 10, 10, 7)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_maxpool_same_against_torch():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 13, 13, 4)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    got = od.max_pool_same(x, 2, 1)       # yolov3-tiny.cfg:93-95: pads bottom/right only
    want = F.max_pool2d(F.pad(xt, (0, 1, 0, 1), value=float("-inf")), 2, 1).permute(0, 2, 3, 1).numpy()
    assert got.shape == (1, 13, 13, 4)
    np.testing.assert_array_equal(got, want)
    x2 = rng.standard_normal((1, 8, 8, 4)).astype(np.float32)
    got2 = od.max_pool_same(x2, 2, 2)
    want2 = F.max_pool2d(torch.from_numpy(x2).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(got2, want2)


def test_upsample_against_torch():
    x = np.random.default_rng(4).standard_normal((2, 5, 5, 3)).astype(np.float32)
    want = F.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(od.upsample2x_nearest(x), want)


def test_batch_norm_against_torch():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 4, 4, 6)).astype(np.float32)
    p = dict(gamma=rng.uniform(.5, 1.5, 6).astype(np.float32), beta=rng.standard_normal(6).astype(np.float32),
             mean=rng.standard_normal(6).astype(np.float32), var=rng.uniform(.5, 1.5, 6).astype(np.float32))
    want = F.batch_norm(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(p["mean"]), torch.from_numpy(p["var"]),
                        torch.from_numpy(p["gamma"]), torch.from_numpy(p["beta"]), False, 0.0, 1e-5).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(od.batch_norm_inference(x, p), want, rtol=1e-5, atol=1e-6)


def test_weights_roundtrip(tmp_path):
    layers, _ = planner.parse_cfg(config.yolov3_tiny_cfg_path, 80)
    params = dw.random_params(layers, seed=3)
    path = str(tmp_path / "t.weights")
    dw.write_weights(path, layers, params)
    n_floats = sum(o * i * k * k + (4 * o if bn else o) for _, o, i, k, bn in dw.conv_param_shapes(layers))
    assert os.path.getsize(path) == 20 + 4 * n_floats and n_floats == 8858734      # 8.85 M params
    olayers, _ = od.parse_layers(config.yolov3_tiny_cfg_path, 80)
    got, consumed, total = od.read_darknet_weights(path, olayers)        # oracle reader
    assert consumed == total
    mine = dw.read_weights(path, layers)                                 # product reader
    for li in params:
        for k in params[li]:
            np.testing.assert_array_equal(got[li][k], params[li][k])
            np.testing.assert_array_equal(mine[li][k], params[li][k])
    with pytest.raises(FileNotFoundError):
        dw.read_weights(str(tmp_path / "missing.weights"), layers)
    with open(path, "r+b") as fh:
        fh.truncate(1000)
    with pytest.raises(ValueError):
        dw.read_weights(path, layers)


def test_fold_batch_norm_equals_unfolded():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 6, 6, 8)).astype(np.float32)
    p = dict(kernel=rng.standard_normal((3, 3, 8, 5)).astype(np.float32), gamma=rng.uniform(.5, 1.5, 5).astype(np.float32),
             beta=rng.standard_normal(5).astype(np.float32), mean=rng.standard_normal(5).astype(np.float32),
             var=rng.uniform(.5, 1.5, 5).astype(np.float32))
    want = od.batch_norm_inference(od.conv2d(x, p["kernel"], 1, True), p)
    k, b = dw.fold_batch_norm(p)
    np.testing.assert_allclose(od.conv2d(x, k, 1, True) + b, want, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- decode known answers
def test_decode_known_answers():
    gold = json.load(open(os.path.join(GOLD, "decode_cases.json")))
    for case in gold["cases"]:
        g = case["grid"]
        out = np.zeros((1, g, g, 255), dtype=np.float32)
        anchors = np.array([case["anchor_wh"]] * 3, dtype=np.float32)
        xy, wh, conf, cls = odec.get_head(out, anchors, 80, tuple(case["input_shape"]))
        gx, gy = case["cell_xy"]
        np.testing.assert_array_equal(xy[0, gy, gx, 0], np.float32(case["box_xy"]))
        np.testing.assert_array_equal(wh[0, gy, gx, 0], np.float32(case["box_wh"]))
        assert conf[0, gy, gx, 0, 0] == np.float32(case["conf"]) and np.all(cls == np.float32(case["cls"]))
        boxes, scores = odec.get_boxes_and_scores(out, anchors, 80, tuple(case["input_shape"]), tuple(case["image_shape"]))
        p = (gy * g + gx) * 3
        np.testing.assert_array_equal(boxes[p], np.float32(case["box_yxyx"]))
        assert np.all(scores[p] == np.float32(case["score_every_class"]))      # all 80 classes tie (utils.py:151)
    for lb in gold["letterbox"]:
        s, oh, ow = odec.letterbox_params(tuple(lb["input_shape"]), tuple(lb["image_shape"]))
        assert s == np.float32(lb["scale"]) and [oh, ow] == lb["offset_hw"], lb["name"]


def test_product_letterbox_params_known_answers_and_bad_shapes():
    """yolo_amd.postprocess.letterbox_params (host side of correct_boxes, utils/utils.py:111-116): the golden
    known answers, and a ValueError for an image without a positive height and width (the reference would divide by
    it and hand back inf / NaN boxes)."""
    from yolo_amd.postprocess import letterbox_params
    gold = json.load(open(os.path.join(GOLD, "decode_cases.json")))
    for lb in gold["letterbox"]:
        s, oh, ow = letterbox_params(tuple(lb["input_shape"]), tuple(lb["image_shape"]))
        assert s == np.float32(lb["scale"]) and [oh, ow] == lb["offset_hw"], lb["name"]
    for bad in ((0, 500), (300, -95), (-1, -1), (float("nan"), 10), (float("inf"), 10), (5,), (1, 2, 3)):
        with pytest.raises(ValueError):
            letterbox_params((608, 608), bad)
    with pytest.raises(ValueError):
        letterbox_params((0, 608), (10, 10))


def test_decode_only_argmax_class_scores():
    rng = np.random.default_rng(7)
    out = rng.standard_normal((1, 13, 13, 255)).astype(np.float32)
    anchors = np.array([[116, 90], [156, 198], [373, 326]], dtype=np.float32)
    boxes, scores = odec.get_boxes_and_scores(out, anchors, 80, (416, 416), (416, 416))
    assert boxes.shape == (507, 4) and scores.shape == (507, 80)
    assert np.all((scores > 0).sum(axis=1) == 1)
    t = out.reshape(1, 13, 13, 3, 85)
    np.testing.assert_array_equal(scores.argmax(1), t[..., 5:].reshape(-1, 80).argmax(1))


# ---------------------------------------------------------------- NMS known answers
def test_nms_known_answers():
    gold = json.load(open(os.path.join(GOLD, "nms_cases.json")))
    for case in gold["cases"]:
        b = np.array(case["boxes"], dtype=np.float32).reshape(-1, 4)
        s = np.array(case["scores"], dtype=np.float32)
        for fn in (onms.non_max_suppression, onms.non_max_suppression_np):
            got = fn(b, s, case["max_out"], case["thr"])
            assert list(got) == case["expect"], (case["name"], fn.__name__, list(got))


def test_nms_more_than_max_boxes_survivors():
    n = 60
    b = np.array([[0, 2 * i, 1, 2 * i + 1] for i in range(n)], dtype=np.float32)
    s = np.linspace(0.1, 0.9, n).astype(np.float32)
    got = onms.non_max_suppression(b, s, 51, 0.3)
    assert list(got) == list(range(n - 1, n - 52, -1))


def test_nms_c_equals_numpy_twin_on_random_boxes():
    rng = np.random.default_rng(8)
    for trial in range(5):
        n = 300
        yx = rng.uniform(0, 100, (n, 2)); hw = rng.uniform(5, 40, (n, 2))
        b = np.concatenate([yx, yx + hw], axis=1).astype(np.float32)
        s = rng.choice(np.linspace(0.1, 1, 50), n).astype(np.float32)       # many exact ties
        np.testing.assert_array_equal(onms.non_max_suppression(b, s, 51, 0.3), onms.non_max_suppression_np(b, s, 51, 0.3))


def test_select_contract():
    rng = np.random.default_rng(9)
    n = 200
    yx = rng.uniform(0, 100, (n, 2)); hw = rng.uniform(5, 40, (n, 2))
    boxes = np.concatenate([yx, yx + hw], axis=1).astype(np.float32)
    scores = np.zeros((n, 80), dtype=np.float32)
    scores[np.arange(n), rng.integers(0, 80, n)] = rng.uniform(0.3, 1, n).astype(np.float32)
    b, s, c, rank, pool = odec.select(boxes, scores, 0.6, 0.3, 51, return_index=True)
    assert np.all(np.diff(c) >= 0)                                   # class-major (inference.py:163-165)
    for cls in np.unique(c):
        assert np.all(np.diff(s[c == cls]) <= 0)                     # NMS selection order within a class
    assert np.all(s >= np.float32(0.6))                              # inclusive threshold (inference.py:127)
    np.testing.assert_array_equal(boxes[pool], b)
    lines = odec.label_lines(b, s, c, (100, 100))
    assert len(lines) == len(c) and lines[0].split(",")[4] == str(int(c[-1]))    # reversed order (inference.py:291)


def test_oracle_reproduces_committed_config1_fixture():
    """tests/golden/tiny416_b1.npz (made by tests/golden/make_golden.py from this oracle): BASELINE config 1.
    Guards the oracle against drift: head tensors (strided subsample + sums) and the final detections."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    got = mg.compute()[-1]
    want = np.load(os.path.join(GOLD, "tiny416_b1.npz"))
    for k in (0, 1):
        np.testing.assert_array_equal(got["head%d_shape" % k], want["head%d_shape" % k])
        np.testing.assert_allclose(got["head%d_sub" % k], want["head%d_sub" % k], rtol=0, atol=2e-5)     # BLAS summation order
        np.testing.assert_allclose(got["head%d_sum" % k], want["head%d_sum" % k], rtol=1e-6)
    assert len(got["classes"]) == len(want["classes"]) > 100
    np.testing.assert_array_equal(got["classes"], want["classes"])
    np.testing.assert_allclose(got["boxes"], want["boxes"], rtol=0, atol=1e-2)
    np.testing.assert_allclose(got["scores"], want["scores"], rtol=0, atol=1e-5)


def test_bf16_storage_model_is_close_to_fp32_graph():
    """oracle.darknet.forward_bf16_model (bf16 weights + activations, fp32 sums: what the build's bf16 mode computes)
    stays within the north_star's 2e-2 (max-abs / max-abs) of the fp32 graph, and differs from it (it is a model of
    bf16 storage, not a copy of `forward`)."""
    from yolo_amd import config, darknet_weights as dw
    layers, oi = od.parse_layers(config.yolov3_tiny_cfg_path, 80)
    params = dw.random_params(layers, seed=3)
    x = np.random.default_rng(3).random((1, 96, 96, 3), dtype=np.float32)
    ref = od.forward(layers, oi, params, x)
    mod = od.forward_bf16_model(layers, oi, params, x)
    for r, m in zip(ref, mod):
        err = np.abs(r - m).max() / np.abs(r).max()
        assert 0 < err <= 2e-2
    a = np.array([1.0, 1.00390625, 1.005859375, -3.1415927, 0.0], dtype=np.float32)
    np.testing.assert_array_equal(od.bf16_round(a), np.array([1.0, 1.0, 1.0078125, -3.140625, 0.0], dtype=np.float32))


def test_one_by_one_stride_two_is_refused():
    """the reference pads top/left for any stride > 1 (build_model.py:151-154); a 1x1 stride-2 conv is not
    implemented that way, so the planner refuses it instead of computing something else."""
    import pytest
    from yolo_amd import planner
    layers = [dict(type="conv", conv_idx=0, cin=8, filters=8, size=1, stride=2, pad=0, bn=True, activation="leaky", same=False)]
    with pytest.raises(ValueError, match="1x1 conv with stride"):
        planner.infer_shapes(layers, 16, 16, 8)


def test_weights_file_with_trailing_floats_warns(tmp_path):
    """a .weights file written for a larger network loads its prefix -- with a warning, not silently."""
    import pytest
    from yolo_amd import config, darknet_weights as dw, planner
    layers = planner.parse_cfg(config.yolov3_tiny_cfg_path, 80)[0]
    params = dw.random_params(layers, seed=0)
    path = str(tmp_path / "w.weights")
    dw.write_weights(path, layers, params)
    with open(path, "ab") as fp:
        np.zeros(7, dtype=np.float32).tofile(fp)
    with pytest.warns(RuntimeWarning, match="floats consumed"):
        dw.read_weights(path, layers)


def test_global_variables_tracks_live_networks_only():
    import gc
    from yolo_amd import build_model
    build_model.reset_default_graph()
    assert build_model.global_variables() == []
