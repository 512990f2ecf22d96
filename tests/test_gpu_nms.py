"""-m gpu: per-class NMS kernel.  Given identical decoded boxes the kept indices must be
BIT-EXACT those of the oracle's sequential greedy NMS (north_star), including ties,
degenerate boxes, the per-class cap and dense / single-class stress inputs."""
import json
import os

import numpy as np
import pytest

import gpu_util
from oracle import decode as odec
from oracle import nms as onms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _one_class(boxes, scores, C=80, cls=3, n_pool=None):
    n = len(scores)
    N = max(n_pool or n, 1)
    dense = np.zeros((N, C), dtype=np.float32)
    bx = np.zeros((N, 4), dtype=np.float32)
    if n:
        dense[:n, cls] = scores
        bx[:n] = boxes
    return bx, dense


def test_nms_known_answers_on_gpu():
    gold = json.load(open(os.path.join(GOLD, "nms_cases.json")))
    for case in gold["cases"]:
        b = np.array(case["boxes"], dtype=np.float32).reshape(-1, 4)
        s = np.array(case["scores"], dtype=np.float32)
        bx, dense = _one_class(b, s, n_pool=8)
        rec = gpu_util.records_from_dense(bx, dense, 0.05)
        out = gpu_util.run_nms([rec], 80, case["max_out"], case["thr"])
        k = int(out["kept_count"][0, 3])
        assert list(out["kept_idx"][0, 3, :k]) == case["expect"], case["name"]
        assert list(out["kept_rank"][0, 3, :k]) == case["expect"], case["name"]     # single class: rank == index
        assert out["kept_count"][0].sum() == k and out["count"][0] == k


def _random_scene(rng, n, C, ties=False, span=400.0):
    yx = rng.uniform(0, span, (n, 2)); hw = rng.uniform(10, 120, (n, 2))
    boxes = np.concatenate([yx, yx + hw], axis=1).astype(np.float32)
    dense = np.zeros((n, C), dtype=np.float32)
    sc = rng.choice(np.linspace(0.05, 1, 40), n) if ties else rng.uniform(0.01, 1, n)
    dense[np.arange(n), rng.integers(0, C, n)] = sc.astype(np.float32)
    return boxes, dense


@pytest.mark.parametrize("n,C,thr_s,ties", [(500, 80, 0.3, False), (3000, 80, 0.001, True), (2535, 20, 0.6, True),
                                             (10647, 80, 0.001, False)])
def test_nms_random_scenes_bit_exact(n, C, thr_s, ties):
    rng = np.random.default_rng(n)
    scenes = [_random_scene(rng, n, C, ties) for _ in range(3)]
    recs = [gpu_util.records_from_dense(b, d, thr_s) for b, d in scenes]
    out = gpu_util.run_nms(recs, C, 51, 0.3)
    for i, (boxes, dense) in enumerate(scenes):
        ob, os_, oc, orank, opool = odec.select(boxes, dense, thr_s, 0.3, 51, return_index=True)
        m = int(out["count"][i])
        assert m == len(oc)
        np.testing.assert_array_equal(out["classes"][i, :m], oc)
        np.testing.assert_array_equal(out["boxes"][i, :m], ob)
        np.testing.assert_array_equal(out["scores"][i, :m], os_)
        got_pool = np.concatenate([out["kept_idx"][i, c, :out["kept_count"][i, c]] for c in range(C)])
        got_rank = np.concatenate([out["kept_rank"][i, c, :out["kept_count"][i, c]] for c in range(C)])
        np.testing.assert_array_equal(got_pool, opool)
        np.testing.assert_array_equal(got_rank, orank)               # == tf nms_index


def test_nms_single_class_stress_and_cap():
    """Worst case of BASELINE config 5: every survivor in one class (22743 boxes at 608)."""
    rng = np.random.default_rng(5)
    n = 22743
    yx = rng.uniform(0, 600, (n, 2)); hw = rng.uniform(4, 60, (n, 2))
    boxes = np.concatenate([yx, yx + hw], axis=1).astype(np.float32)
    scores = rng.uniform(0.01, 1, n).astype(np.float32)
    bx, dense = _one_class(boxes, scores, cls=79)
    out = gpu_util.run_nms([gpu_util.records_from_dense(bx, dense, 0.001)], 80, 51, 0.3)
    want = onms.non_max_suppression(boxes, scores, 51, 0.3)
    assert out["kept_count"][0, 79] == 51 == len(want)
    np.testing.assert_array_equal(out["kept_idx"][0, 79], want)
    # heavy overlap: few survivors, the scan must run through all 22743 candidates
    yx2 = rng.uniform(0, 30, (n, 2)); hw2 = rng.uniform(200, 260, (n, 2))
    boxes2 = np.concatenate([yx2, yx2 + hw2], axis=1).astype(np.float32)
    bx2, dense2 = _one_class(boxes2, scores, cls=0)
    out2 = gpu_util.run_nms([gpu_util.records_from_dense(bx2, dense2, 0.001)], 80, 51, 0.3)
    want2 = onms.non_max_suppression(boxes2, scores, 51, 0.3)
    k = int(out2["kept_count"][0, 0])
    assert k == len(want2) and k < 51
    np.testing.assert_array_equal(out2["kept_idx"][0, 0, :k], want2)


def test_nms_multi_class_ties_share_a_box():
    """A box whose class probabilities tie keeps its score in every tied class (utils.py:151)
    and is therefore a candidate of several per-class NMS runs."""
    boxes = np.array([[0, 0, 10, 10], [0, 1, 10, 11], [50, 50, 60, 60]], dtype=np.float32)
    dense = np.zeros((3, 80), dtype=np.float32)
    dense[0, [2, 7]] = 0.9
    dense[1, 7] = 0.8
    dense[2, 2] = 0.7
    out = gpu_util.run_nms([gpu_util.records_from_dense(boxes, dense, 0.6)], 80, 51, 0.3)
    ob, os_, oc = odec.select(boxes, dense, 0.6, 0.3, 51)
    m = int(out["count"][0])
    assert m == len(oc) == 3
    np.testing.assert_array_equal(out["classes"][0, :m], oc)
    np.testing.assert_array_equal(out["boxes"][0, :m], ob)


def test_nms_crowded_classes_lds_cached_path():
    """Dense scenes put thousands of boxes in a few classes (BASELINE config 5: three classes of ~5 750 at 608).
    Pairs of 1 025..6 144 candidates take nms_kernel's LDS-cached sweep; this pins it -- score ties, rank
    (tf nms_index) and a class just over the cache limit included -- against the oracle and, through run_nms,
    against the plain y3_nms scan."""
    rng = np.random.default_rng(77)
    C, n = 80, 18000
    sizes = {3: 5776, 40: 1300, 79: 6200, 7: 2048, 12: 900}          # 6200 > 6144: stays on the un-cached scan
    scenes = []
    for img in range(2):
        yx = rng.uniform(0, 560, (n, 2)); hw = rng.uniform(8, 90, (n, 2))
        boxes = np.concatenate([yx, yx + hw], axis=1).astype(np.float32)
        dense = np.zeros((n, C), dtype=np.float32)
        cls = np.concatenate([np.full(k, c) for c, k in sizes.items()])
        cls = np.concatenate([cls, rng.integers(0, C, n - len(cls))])
        rng.shuffle(cls)
        sc = rng.choice(np.linspace(0.05, 1, 400), n) if img == 0 else rng.uniform(0.01, 1, n)
        dense[np.arange(n), cls] = sc.astype(np.float32)
        scenes.append((boxes, dense))
    recs = [gpu_util.records_from_dense(b, d, 0.001) for b, d in scenes]
    out = gpu_util.run_nms(recs, C, 51, 0.3)
    for i, (boxes, dense) in enumerate(scenes):
        ob, os_, oc, orank, opool = odec.select(boxes, dense, 0.001, 0.3, 51, return_index=True)
        m = int(out["count"][i])
        assert m == len(oc)
        np.testing.assert_array_equal(out["classes"][i, :m], oc)
        np.testing.assert_array_equal(out["boxes"][i, :m], ob)
        np.testing.assert_array_equal(out["scores"][i, :m], os_)
        got_pool = np.concatenate([out["kept_idx"][i, c, :out["kept_count"][i, c]] for c in range(C)])
        got_rank = np.concatenate([out["kept_rank"][i, c, :out["kept_count"][i, c]] for c in range(C)])
        np.testing.assert_array_equal(got_pool, opool)
        np.testing.assert_array_equal(got_rank, orank)


@pytest.mark.parametrize("n,C,thr_s,ties", [(500, 80, 0.3, False), (3000, 80, 0.001, True), (2535, 3, 0.05, True),
                                             (10647, 80, 0.001, False), (22743, 4, 0.001, True)])
def test_bitmask_nms_keeps_the_same_indices(n, C, thr_s, ties):
    """The all-pairs IoU bitmask + scan form of NMS (north_star's sketch; experimental flavour only, kept for the A/B
    of DESIGN.md 3.5) keeps exactly the indices of the greedy kernels, ties and crowded classes (thousands of
    candidates per pair: many 64-wide column blocks, multi-pass bitonic sorts) included."""
    rng = np.random.default_rng(n + C)
    scenes = [_random_scene(rng, n, C, ties) for _ in range(2)]
    recs = [gpu_util.records_from_dense(b, d, thr_s) for b, d in scenes]
    want = gpu_util.run_nms(recs, C, 51, 0.3, with_rank=False)
    got = gpu_util.run_nms_bitmask(recs, C, 51, 0.3)
    assert (got["status"] == 0).all()
    np.testing.assert_array_equal(got["kept_count"], want["kept_count"])
    assert want["kept_count"].sum() > 0
    for b in range(len(scenes)):
        for c in range(C):
            k = want["kept_count"][b, c]
            np.testing.assert_array_equal(got["kept_idx"][b, c, :k], want["kept_idx"][b, c, :k])
    # masks that do not fit are reported, not overrun
    small = gpu_util.run_nms_bitmask(recs, C, 51, 0.3, mask_words=8)
    assert (small["status"] == 1).all() and (small["kept_count"] == 0).all()
