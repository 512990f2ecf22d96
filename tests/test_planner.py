"""CPU tests of the planner (fusion, zero-copy concat, arena liveness) and of the weight
packing: every plan is interpreted op by op in numpy (tests/plan_sim.py) and must give the
oracle's head tensors."""
import numpy as np
import pytest

import plan_sim
from oracle import darknet as od
from yolo_amd import config, planner
from yolo_amd import darknet_weights as dw


def _check(cfg, size, batch, fuse, pad_input=False):
    layers, oi = planner.parse_cfg(cfg, 80)
    ol, ooi = od.parse_layers(cfg, 80)
    params = dw.random_params(layers, seed=11)
    x = np.random.default_rng(1).random((batch, size, size, 3), dtype=np.float32)
    ref = od.forward(ol, ooi, params, x)
    plan = planner.build_plan(layers, oi, batch, size, size, "fp32", fuse=fuse, pad_input=pad_input)
    out = plan_sim.Sim(plan, layers, params, x).run()
    for a, b in zip(out, ref):
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4)
    return plan


@pytest.mark.parametrize("fuse", [True, False])
def test_tiny_plan_matches_oracle(fuse):
    plan = _check(config.yolov3_tiny_cfg_path, 416, 1, fuse)
    kinds = [op["kind"] for op in plan.ops]
    # fused: the first max pool (2x2 stride 2) is taken by the first convolution's kernel
    assert kinds.count("conv") == 13 and kinds.count("maxpool") == (5 if fuse else 6)
    assert bool(plan.ops[0].get("pool")) == fuse and sum(1 for op in plan.ops if op.get("pool")) == (1 if fuse else 0)
    assert kinds.count("upsample") == (0 if fuse else 1) and "copy" not in kinds


@pytest.mark.parametrize("fuse", [True, False])
def test_yolov3_plan_matches_oracle(fuse):
    plan = _check(config.yolov3_cfg_path, 96, 2, fuse)
    kinds = [op["kind"] for op in plan.ops]
    assert kinds.count("conv") == 75 and "copy" not in kinds
    if fuse:      # all 23 shortcuts, both upsamples and both concats are fused away
        assert set(kinds) == {"conv"}
        assert sum(1 for op in plan.ops if op["res"] is not None) == 23
        assert sum(1 for op in plan.ops if op["ups"]) == 2
    else:
        assert kinds.count("add") == 23 and kinds.count("upsample") == 2


def test_plan_shapes_and_alignment():
    layers, oi = planner.parse_cfg(config.yolov3_cfg_path, 80)
    for dtype, epc in (("bf16", 8), ("fp32", 4)):
        plan = planner.build_plan(layers, oi, 64, 608, 608, dtype)
        assert [(h["h"], h["c"], h["ld"]) for h in plan.heads] == [(19, 255, 256), (38, 255, 256), (76, 255, 256)]
        for op in plan.ops:
            for k in ("x", "y", "res"):
                t = op.get(k)
                if t is None or t["buf"] == "in":
                    continue
                assert t["ld"] % epc == 0 and t["off"] % epc == 0
                if t["buf"][0] == "arena":
                    assert t["buf"][1] % 256 == 0
        # route [-1, 61] and [-1, 36] read one buffer: 768 and 384 channels
        assert sorted(op["x"]["ld"] for op in plan.ops if op.get("in_c") in (768, 384)) == [384, 768]
        assert plan.arena_bytes < 8 * 2 ** 30


def test_stem_fusion_plan():
    """bf16 yolov3 plans run layers 0-2 as one 'stem' launch whose 32-channel full-resolution
    tensor has no buffer at all; fp32 plans, tiny, and stem=False keep one launch per conv."""
    layers, oi = planner.parse_cfg(config.yolov3_cfg_path, 80)
    plan = planner.build_plan(layers, oi, 2, 608, 608, "bf16")
    kinds = [op["kind"] for op in plan.ops]
    assert kinds[0] == "stem" and kinds.count("conv") == 72 and len(kinds) == 73
    st = plan.ops[0]
    assert st["layers"] == (0, 1, 2) and st["leaky"] == (True, True, True) and st["out_hw"] == (304, 304)
    assert st["y"]["ld"] == 64 and st["y2"]["ld"] == 32 and st["y"]["buf"] != st["y2"]["buf"]
    assert plan.conv_layers[:3] == [0, 1, 2] and len(plan.conv_layers) == 75
    # the shortcut at layer 4 still finds conv B's output as the residual of layer 3's conv
    assert plan.ops[1]["res"] == st["y"] and plan.ops[1]["x"] == st["y2"]
    flops = sum(planner.op_flops(op, 1) for op in plan.ops)
    assert abs(flops / 1e9 - planner.conv_gflops(layers, 608, 608)) < 1e-6
    # from batch 4 on (>= RESBLOCK_MIN_TILES tiles) the 1x1 moves into the first residual block's launch
    p4 = planner.build_plan(layers, oi, 4, 608, 608, "bf16")
    assert p4.ops[0]["layers"] == (0, 1, None) and p4.ops[0]["y2"] is None and p4.ops[0]["leaky"] == (True, True)
    assert p4.ops[1]["kind"] == "resblock" and p4.ops[1]["c"] == 64 and p4.ops[1]["x"] == p4.ops[0]["y"]
    assert abs(sum(planner.op_flops(op, 1) for op in p4.ops) / 1e9 - planner.conv_gflops(layers, 608, 608)) < 1e-6
    assert len(p4.conv_layers) == 75
    unfused = planner.build_plan(layers, oi, 2, 608, 608, "bf16", stem=False)
    assert [op["kind"] for op in unfused.ops].count("conv") == 75
    assert plan.arena_bytes <= unfused.arena_bytes
    assert planner.build_plan(layers, oi, 4, 608, 608, "fp32").ops[0]["kind"] == "conv"
    tl, toi = planner.parse_cfg(config.yolov3_tiny_cfg_path, 80)
    assert planner.build_plan(tl, toi, 1, 416, 416, "bf16").ops[0]["kind"] == "conv"


def test_yolov3_spp_plan_matches_oracle():
    """cfg stem 'yolov3-spp' (build_model.py:70-71: head convolutions 59, 67, 75): max-pool 5/9/13 stride 1
    SAME and a four-way route.  The three pools and the conv in front of them write straight into their
    slices of the 2048-channel concat buffer; Darknet prints 141.45 BFLOPs for this cfg at 608."""
    layers, oi = planner.parse_cfg(config.yolov3_spp_cfg_path, 80)
    assert [L["conv_idx"] for L in layers if L["type"] == "conv" and L["head"]] == [59, 67, 75]
    assert abs(planner.conv_gflops(layers, 608, 608) - 141.449) < 1e-3
    plan = _check(config.yolov3_spp_cfg_path, 64, 1, True)
    kinds = [op["kind"] for op in plan.ops]
    assert kinds.count("maxpool") == 3 and kinds.count("conv") == 76 and "copy" not in kinds
    pools = [op for op in plan.ops if op["kind"] == "maxpool"]
    assert sorted(op["size"] for op in pools) == [5, 9, 13] and all(op["y"]["ld"] == 2048 for op in pools)


def test_layer_walk_quirks():
    layers, oi = planner.parse_cfg(config.yolov3_tiny_cfg_path, 80)
    assert oi == [15, 22] and len(layers) == 24
    assert layers[17] == dict(type="route", ids=[13]) and layers[20]["ids"] == [19, 8]
    assert layers[15]["filters"] == 255 and layers[15]["head"]
    layers5, _ = planner.parse_cfg(config.yolov3_tiny_cfg_path, 5)
    assert layers5[15]["filters"] == 30 and layers5[22]["filters"] == 30     # build_model.py:156-159
    shapes = planner.infer_shapes(layers, 416, 416)
    assert shapes[11] == (13, 13, 512) and shapes[10] == (13, 13, 512)       # maxpool size 2 stride 1, SAME
    assert shapes[20] == (26, 26, 384)


def test_direct_first_layer_plan_still_available():
    layers, oi = planner.parse_cfg(config.yolov3_tiny_cfg_path, 80)
    plan = planner.build_plan(layers, oi, 1, 416, 416, "bf16")
    assert plan.ops[0]["kind"] == "conv" and plan.ops[0]["in_f32"] and plan.ops[0]["in_c"] == 3
    plan = planner.build_plan(layers, oi, 1, 416, 416, "bf16", pad_input=True)
    assert plan.ops[0]["kind"] == "padin" and plan.ops[1]["in_c"] == 8 and not plan.ops[1]["in_f32"]
    plan = planner.build_plan(layers, oi, 1, 416, 416, "fp32", pad_input=True)
    assert plan.ops[1]["in_c"] == 4


def test_padded_input_plan_matches_oracle():
    plan = _check(config.yolov3_tiny_cfg_path, 416, 1, True, pad_input=True)
    assert plan.ops[0]["kind"] == "padin"


def test_residual_blocks_fuse_into_one_op():
    """1x1 -> 3x3 -> shortcut with 256 channels on a map whose width is a multiple of 19, or with 128 / 64 channels at any
    width, becomes one `resblock` op (csrc/conv_resblock*.hip): the 1x1's output gets no buffer.  Interpreted in numpy, the fused plan gives the
    oracle's heads; automatic mode fuses only the widths that measured faster (RESBLOCK_AUTO_CHANNELS), only launches of
    >= RESBLOCK_MIN_TILES tiles, and only in bf16 plans."""
    cfg = config.yolov3_cfg_path
    layers, oi = planner.parse_cfg(cfg, 80)
    ol, ooi = od.parse_layers(cfg, 80)
    params = dw.random_params(layers, seed=11)
    x = np.random.default_rng(1).random((1, 608, 608, 3), dtype=np.float32)           # 608 = 32 * 19: the only input size with such maps
    ref = od.forward(ol, ooi, params, x)
    plan = planner.build_plan(layers, oi, 1, 608, 608, "fp32", resblock=True)
    rbs = [op for op in plan.ops if op["kind"] == "resblock"]
    assert [(op["c"], op["hw"]) for op in rbs] == [(64, (304, 304))] + [(128, (152, 152))] * 2 + [(256, (76, 76))] * 8
    assert len(plan.ops) == 75 - 11 and all(op["writes_layer"] == op["layers"][2] for op in rbs)
    for a, b in zip(plan_sim.Sim(plan, layers, params, x).run(), ref):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4)
    big = planner.build_plan(layers, oi, 64, 608, 608, "bf16")
    assert [(op["c"], op["tiles"]) for op in big.ops if op["kind"] == "resblock"] == [(64, 23104)] + [(128, 12160)] * 2 and len(big.ops) == 71
    every = planner.build_plan(layers, oi, 64, 608, 608, "bf16", resblock=True)
    assert [op["tiles"] for op in every.ops if op["kind"] == "resblock"] == [23104] + [12160] * 2 + [1536] * 8 and len(every.ops) == 63
    stem = every.ops[0]                            # the 64-channel block computes the 1x1 the stem's launch otherwise would
    assert stem["kind"] == "stem" and stem["layers"] == (0, 1, None) and stem["y2"] is None
    assert not any(op["kind"] == "resblock" for op in planner.build_plan(layers, oi, 1, 608, 608, "bf16").ops)
    assert not any(op["kind"] == "resblock" for op in planner.build_plan(layers, oi, 64, 608, 608, "fp32").ops)
    # the 128-channel kernel takes any width (104 x 104 at 416); the 256-channel one needs a multiple of 19
    assert [op["c"] for op in planner.build_plan(layers, oi, 64, 416, 416, "bf16").ops if op["kind"] == "resblock"] == [64, 128, 128]
    small = planner.build_plan(layers, oi, 64, 608, 608, "bf16", resblock=False)
    # the 64-channel block cannot write over its input (its tiles read their neighbours' halo), so the arena's peak is
    # x + out at 304 x 304 instead of x + x/2: 1.5 GB against 1.1 GB of a 288 GB card
    assert big.arena_bytes <= small.arena_bytes * 4 // 3 + 4096


def test_inplace_shortcuts_alias_the_dead_operand():
    """A convolution with a fused shortcut writes its sum over the shortcut operand when that is the operand's last
    use (18 of YOLOv3's 23 shortcuts: not the stem's, not the two whose operand a later route reads... and the numpy
    interpreter, which executes the aliasing for real, still reproduces the oracle)."""
    cfg = config.yolov3_cfg_path
    layers, oi = planner.parse_cfg(cfg, 80)
    def aliased(plan):
        return [op["writes_layer"] for op in plan.ops
                if op.get("res") is not None and (op["res"]["buf"], op["res"]["off"]) == (op["y"]["buf"], op["y"]["off"])]
    on = planner.build_plan(layers, oi, 2, 160, 160, "fp32", inplace_shortcut=True)
    off = planner.build_plan(layers, oi, 2, 160, 160, "fp32", inplace_shortcut=False)
    assert aliased(off) == [] and len(aliased(on)) >= 18
    # a shortcut operand that a later layer still reads keeps its own storage: layer 36 and 61 feed the head branches
    for op in on.ops:
        if op.get("res") is not None and op["writes_layer"] in aliased(on):
            src = layers[op["writes_layer"]]["src"]
            assert src not in (36, 61)
    ol, ooi = od.parse_layers(cfg, 80)
    params = dw.random_params(layers, seed=12)
    x = np.random.default_rng(2).random((2, 160, 160, 3), dtype=np.float32)
    ref = od.forward(ol, ooi, params, x)
    for a, b in zip(plan_sim.Sim(on, layers, params, x).run(), ref):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4)
    assert on.arena_bytes <= off.arena_bytes


def test_automatic_mode_fuses_the_64_channel_block_only_behind_the_stem():
    """resblock64_kernel sums its 1x1 in the stem's k-slot order, so the automatic mode (same bits at every
    batch size) may fuse the 64-channel block only where the small-batch plan computes that 1x1 in the stem's launch."""
    layers, oi = planner.parse_cfg(config.yolov3_cfg_path, 80)
    for kw, want in ((dict(), True), (dict(stem=False), False), (dict(pad_input=True), False)):
        plan = planner.build_plan(layers, oi, 64, 608, 608, "bf16", **kw)
        cs = [op["c"] for op in plan.ops if op["kind"] == "resblock"]
        assert (64 in cs) == want, (kw, cs)
        assert 128 in cs                                        # the 128-channel blocks do not depend on the stem
    forced = planner.build_plan(layers, oi, 64, 608, 608, "bf16", stem=False, resblock=True)
    assert 64 in [op["c"] for op in forced.ops if op["kind"] == "resblock"]      # resblock=True: the caller's choice


ALIAS_CFG = """[net]
[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky
[route]
layers=-1
[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky
[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky
[shortcut]
from=-4
activation=linear
[route]
layers=1
[convolutional]
filters=8
size=1
stride=1
pad=1
activation=linear
"""


def test_inplace_shortcut_sees_readers_through_route_aliases(tmp_path):
    """a single-input route shares its source's storage; a layer that reads the shortcut operand through
    such an alias AFTER the sum is written must keep the sum out of the operand's storage."""
    path = tmp_path / "yolov3.cfg"
    path.write_text(ALIAS_CFG)
    layers, oi = planner.parse_cfg(str(path), 80)
    ol, ooi = od.parse_layers(str(path), 80)
    params = dw.random_params(layers, seed=5)
    x = np.random.default_rng(5).random((1, 16, 16, 3), dtype=np.float32)
    ref = od.forward(ol, ooi, params, x)
    plan = planner.build_plan(layers, oi, 1, 16, 16, "fp32", stem=False, inplace_shortcut=True)
    sc = [op for op in plan.ops if op.get("res") is not None]
    assert len(sc) == 1 and sc[0]["res"]["buf"] != sc[0]["y"]["buf"]        # not in place: layer 6 still reads layer 0
    out = plan_sim.Sim(plan, layers, params, x).run()
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4)
