"""-m gpu: y3_resblock_fused (one launch: 1x1 -> 3x3 -> shortcut, the 1x1's output never leaving LDS) against the
two y3_conv2d_fused launches it replaces -- BIT-IDENTICAL, every sum running over k in the same order -- and against
the oracle's numpy convolution; then whole networks with and without the fusion."""
import ctypes

import numpy as np
import pytest
import torch

import gpu_util
from oracle import darknet as od
from yolo_amd import _lib, build_model, config
from yolo_amd import darknet_weights as dw
from yolo_amd.build_model import pack_conv

pytestmark = pytest.mark.gpu


def _run_block(x, k1, b1, k2, b2, leaky=(True, True), x_pad=(0, 0), out_pad=(0, 0), reverse=False):
    """x NHWC fp32 numpy (bf16-representable).  Returns the fused launch's output, fp32 numpy."""
    lib = _lib.load()
    B, H, W, C = x.shape
    w1, bb1, f1, k1p, n1p = pack_conv(k1, b1, "bf16", False, "cuda")
    w2, bb2, f2, k2p, n2p = pack_conv(k2, b2, "bf16", False, "cuda")
    x_ld, out_ld = x_pad[0] + C + x_pad[1], out_pad[0] + C + out_pad[1]
    xin = torch.full((B, H, W, x_ld), 7.0, dtype=torch.bfloat16, device="cuda")
    xin[..., x_pad[0]:x_pad[0] + C] = torch.from_numpy(x).cuda().to(torch.bfloat16)
    out = torch.full((B, H, W, out_ld), -9.0, dtype=torch.bfloat16, device="cuda")
    zeros = torch.zeros(256, dtype=torch.uint8, device="cuda")
    d = _lib.ResblockDesc()
    d.dtype, d.batch, d.h, d.w, d.c, d.x_ld, d.out_ld = _lib.Y3_BF16, B, H, W, C, x_ld, out_ld
    d.k1_pad, d.n1_pad, d.k2_pad, d.n2_pad = k1p, n1p, k2p, n2p
    d.flags = (1 if leaky[0] else 0) | (2 if leaky[1] else 0) | (_lib.CONV_REVERSE if reverse else 0)
    _lib.check(lib.y3_resblock_fused(ctypes.byref(d), xin.data_ptr() + x_pad[0] * 2, w1.data_ptr(), bb1.data_ptr(), w2.data_ptr(),
                                     bb2.data_ptr(), out.data_ptr() + out_pad[0] * 2, zeros.data_ptr(), gpu_util.stream()))
    torch.cuda.synchronize()
    full = out.float().cpu().numpy()
    if out_pad[0]:
        assert np.all(full[..., :out_pad[0]] == -9.0)
    if out_pad[1]:
        assert np.all(full[..., out_pad[0] + C:] == -9.0)
    return full[..., out_pad[0]:out_pad[0] + C]


def _case(rng, B, H, W, C):
    x = gpu_util.bf16_round((rng.standard_normal((B, H, W, C))).astype(np.float32))
    k1 = gpu_util.bf16_round((rng.standard_normal((1, 1, C, C // 2)) * np.sqrt(2.0 / C)).astype(np.float32))
    k2 = gpu_util.bf16_round((rng.standard_normal((3, 3, C // 2, C)) * np.sqrt(2.0 / (9 * C // 2)) * 0.3).astype(np.float32))
    b1 = rng.standard_normal(C // 2).astype(np.float32) * 0.2
    b2 = rng.standard_normal(C).astype(np.float32) * 0.2
    return x, k1, b1, k2, b2


@pytest.mark.parametrize("B,H,W,C", [(2, 76, 76, 256), (3, 38, 19, 256), (1, 13, 38, 256), (2, 30, 57, 128), (1, 152, 152, 128),
                                     (1, 5, 19, 256), (2, 40, 50, 64), (1, 96, 64, 64), (1, 304, 304, 64)])
@pytest.mark.parametrize("leaky", [(True, True), (False, True)])
def test_fused_block_equals_two_launches_bit_for_bit(B, H, W, C, leaky):
    rng = np.random.default_rng(B * 1000 + H + C)
    x, k1, b1, k2, b2 = _case(rng, B, H, W, C)
    if C == 64:
        # The 64-channel block is the stem's neighbour: its 1x1 sums the channels in the order of the stem's conv C
        # (conv_stem.hip: MFMA k-slot q holds 8-channel chunk (0, 2, 1, 3)[q] of each 32), because a network must give
        # the same bits whether that 1x1 runs in the stem's launch or in the block's (the whole-network test below).
        # An MFMA's sum depends on the k order in the last bit now and then, so the two-launch chain gets the same order:
        P = np.concatenate([np.arange(8) + 32 * kb + 8 * q for kb in (0, 1) for q in (0, 2, 1, 3)])
        y = gpu_util.run_conv(np.ascontiguousarray(x[..., P]), np.ascontiguousarray(k1[:, :, P, :]), b1, 1, "bf16", leaky=leaky[0])
    else:
        y = gpu_util.run_conv(x, k1, b1, 1, "bf16", leaky=leaky[0])                   # 1x1, rounded to bf16
    want = gpu_util.run_conv(y, k2, b2, 1, "bf16", leaky=leaky[1], residual=x)        # 3x3 + shortcut
    got = _run_block(x, k1, b1, k2, b2, leaky=leaky)
    np.testing.assert_array_equal(got, want)
    got_r = _run_block(x, k1, b1, k2, b2, leaky=leaky, x_pad=(8, 16), out_pad=(16, 8), reverse=True)      # slices, reversed order
    np.testing.assert_array_equal(got_r, want)
    # and the oracle (fp32 numpy on the same bf16 operands, y rounded to bf16 as the two launches do)
    yo = od.conv2d(x, k1, 1, True) + b1
    yo = od.bf16_round((od.leaky_relu(yo) if leaky[0] else yo).astype(np.float32))
    zo = od.conv2d(yo, k2, 1, True) + b2
    zo = (od.leaky_relu(zo) if leaky[1] else zo) + x
    assert np.all(np.abs(got - zo) <= 1e-2 * np.maximum(1.0, np.abs(zo)))


def test_fused_block_rejects_what_it_cannot_run():
    lib = _lib.load()
    d = _lib.ResblockDesc()
    d.dtype, d.batch, d.h, d.w, d.c = _lib.Y3_BF16, 1, 76, 76, 512
    tiles = ctypes.c_int64(0)
    assert lib.y3_resblock_tiles(ctypes.byref(d), ctypes.byref(tiles)) == -1 and b"channels" in lib.y3_last_error()
    d.c, d.w = 256, 52
    assert lib.y3_resblock_tiles(ctypes.byref(d), ctypes.byref(tiles)) == -1 and b"multiple of 19" in lib.y3_last_error()
    d.w, d.dtype = 76, _lib.Y3_F32
    assert lib.y3_resblock_tiles(ctypes.byref(d), ctypes.byref(tiles)) == -1
    d.dtype, d.batch = _lib.Y3_BF16, 64
    assert lib.y3_resblock_tiles(ctypes.byref(d), ctypes.byref(tiles)) == 0 and tiles.value == 1536


@pytest.mark.parametrize("size,batch", [(608, 1), (608, 5)])
def test_network_with_fused_blocks_equals_network_without(size, batch):
    """Whole YOLOv3 plans at 608 (= 32 * 19, the input size whose 76 / 152 maps the tiling fits): with every eligible
    residual block fused (resblock=True: the 8 blocks at 76x76, the 2 at 152x152 and the one at 304x304) the heads are bit-identical to the
    plan without (resblock=False)."""
    layers, oi = od.parse_layers(config.yolov3_cfg_path, 80)
    params = dw.random_params(layers, seed=17)
    x = torch.from_numpy(np.random.default_rng(17).random((batch, size, size, 3), dtype=np.float32)).cuda()
    outs = []
    for rb in (True, False):
        net = build_model.Network(config.yolov3_cfg_path, 80, dtype="bf16", resblock=rb)
        net.set_params(params)
        plan = net.plan_for(batch, size, size).plan
        n_rb = sum(1 for op in plan.ops if op["kind"] == "resblock")
        assert n_rb == (11 if rb else 0)          # (the 64-channel block's 1x1 moves out of the stem's launch: no op less)
        assert len(plan.ops) == 73 - (10 if rb else 0)
        outs.append([h.clone() for h in net(x)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
