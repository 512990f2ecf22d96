"""-m gpu: the fused convolution kernels against the oracle's conv (numpy fp32), through the
C-ABI.  fp32 mode: <= 1e-3 max-abs (north_star's head-tensor tolerance, applied per layer);
bf16 mode: inputs/weights are rounded to bf16 first so that only accumulation order and the
final bf16 rounding differ: |err| <= 1e-2 * max(1, |ref|)."""
import ctypes

import numpy as np
import pytest

import gpu_util
from oracle import darknet as od

pytestmark = pytest.mark.gpu


def _ref(x, kernel, bias, stride, leaky, residual=None, ups=False):
    ks = kernel.shape[0]
    xin = np.pad(x, ((0, 0), (1, 0), (1, 0), (0, 0))) if (stride > 1 and ks == 3) else x
    y = od.conv2d(xin, kernel, stride, stride == 1) + bias
    if leaky:
        y = od.leaky_relu(y)
    if residual is not None:
        y = y + residual
    if ups:
        y = od.upsample2x_nearest(y)
    return y.astype(np.float32)


CASES = [
    # B, H, Cin, Cout, ks, stride, leaky
    (2, 13, 64, 128, 3, 1, True),       # BN=128 tile, M = 338 (not a tile multiple)
    (1, 26, 128, 255, 1, 1, False),     # head: Cout 255 -> padded 256
    (2, 16, 32, 64, 3, 2, True),        # stride 2, Cin=32 (bf16 k-step spans two taps), BN=64
    (1, 20, 64, 32, 1, 1, True),        # BN=32
    (1, 12, 16, 32, 3, 1, True),        # tiny layer 2: Cin=16
    (1, 10, 384, 128, 1, 1, True),      # route input, Cin not a power of two
    (3, 19, 96, 192, 3, 1, True),       # odd sizes: N tiles 2, K = 864 (padded to 896 in bf16)
    (1, 7, 512, 1024, 3, 1, True),      # deep K (4608), 8 n-tiles
]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_igemm(case, dtype):
    B, H, C, O, ks, stride, leaky = case
    rng = np.random.default_rng(abs(hash(case)) % 2 ** 31)
    x = rng.standard_normal((B, H, H, C)).astype(np.float32)
    k = (rng.standard_normal((ks, ks, C, O)) * np.sqrt(2.0 / (ks * ks * C))).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    if dtype == "bf16":
        x, k = gpu_util.bf16_round(x), gpu_util.bf16_round(k)
    got = gpu_util.run_conv(x, k, b, stride, dtype, leaky=leaky)
    want = _ref(x, k, b, stride, leaky)
    assert got.shape == want.shape
    if dtype == "fp32":
        assert np.abs(got - want).max() <= 1e-3
    else:
        assert np.all(np.abs(got - want) <= 1e-2 * np.maximum(1.0, np.abs(want)))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_conv_residual_and_slices(dtype):
    rng = np.random.default_rng(21)
    x = rng.standard_normal((2, 14, 14, 64)).astype(np.float32)
    k = (rng.standard_normal((3, 3, 64, 128)) * 0.06).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    r = rng.standard_normal((2, 14, 14, 128)).astype(np.float32)
    if dtype == "bf16":
        x, k, r = gpu_util.bf16_round(x), gpu_util.bf16_round(k), gpu_util.bf16_round(r)
    got = gpu_util.run_conv(x, k, b, 1, dtype, leaky=True, residual=r, in_pad=(16, 8), out_pad=(8, 24))
    want = _ref(x, k, b, 1, True, residual=r)
    tol = 1e-3 if dtype == "fp32" else 1e-2 * np.maximum(1.0, np.abs(want))
    assert np.all(np.abs(got - want) <= tol)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_conv_upsample_store(dtype):
    rng = np.random.default_rng(22)
    x = rng.standard_normal((2, 13, 13, 256)).astype(np.float32)
    k = (rng.standard_normal((1, 1, 256, 128)) * 0.09).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    if dtype == "bf16":
        x, k = gpu_util.bf16_round(x), gpu_util.bf16_round(k)
    got = gpu_util.run_conv(x, k, b, 1, dtype, leaky=True, ups=True, out_pad=(0, 256))
    want = _ref(x, k, b, 1, True, ups=True)
    assert got.shape == (2, 26, 26, 128)
    tol = 1e-3 if dtype == "fp32" else 1e-2 * np.maximum(1.0, np.abs(want))
    assert np.all(np.abs(got - want) <= tol)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_conv_head_fp32_output(dtype):
    rng = np.random.default_rng(23)
    x = rng.standard_normal((1, 19, 19, 256)).astype(np.float32)
    k = (rng.standard_normal((1, 1, 256, 255)) * 0.05).astype(np.float32)
    b = rng.standard_normal(255).astype(np.float32)
    if dtype == "bf16":
        x, k = gpu_util.bf16_round(x), gpu_util.bf16_round(k)
    got = gpu_util.run_conv(x, k, b, 1, dtype, leaky=False, out_f32=True)
    want = _ref(x, k, b, 1, False)
    assert np.abs(got - want).max() <= 1e-3          # fp32 store: no output rounding in either mode


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cout", [16, 32])
def test_conv_first_layer_direct(dtype, cout):
    rng = np.random.default_rng(24 + cout)
    x = rng.random((2, 33, 33, 3), dtype=np.float32)
    k = (rng.standard_normal((3, 3, 3, cout)) * 0.3).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got = gpu_util.run_conv(x, k, b, 1, dtype, leaky=True, in_f32=True)
    want = _ref(x, k, b, 1, True)
    tol = 1e-4 if dtype == "fp32" else 1e-2 * np.maximum(1.0, np.abs(want))
    assert np.all(np.abs(got - want) <= tol)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 16, 20, 16), (1, 34, 6, 24), (3, 8, 8, 32)], ids=str)
def test_conv_first_layer_with_pool(case, dtype):
    """Y3_CONV_POOL2 (YOLOv3-tiny layers 0-1 in one launch) == the same convolution followed by the 2x2 stride-2
    max pool, bit for bit (and both against the oracle); bad shapes are refused."""
    from oracle import darknet as od
    B, H, W, O = case
    rng = np.random.default_rng(H * 100 + O)
    x = rng.random((B, H, W, 3), dtype=np.float32)
    k = (rng.standard_normal((3, 3, 3, O)) * 0.3).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    full = gpu_util.run_conv(x, k, b, 1, dtype, leaky=True, in_f32=True)
    fused = gpu_util.run_conv(x, k, b, 1, dtype, leaky=True, in_f32=True, pool=True)
    np.testing.assert_array_equal(fused, od.max_pool_same(full, 2, 2))
    want = od.max_pool_same(_ref(x, k, b, 1, True), 2, 2)
    tol = 1e-4 if dtype == "fp32" else 1e-2 * np.maximum(1.0, np.abs(want))
    assert np.all(np.abs(fused - want) <= tol)
    with pytest.raises(ValueError):          # odd output size
        gpu_util.run_conv(x[:, :H - 1], k, b, 1, dtype, leaky=True, in_f32=True, pool=True)


@pytest.mark.parametrize("has_c", [True, False])
@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 48, 80), (1, 37, 45), (3, 16, 32)], ids=str)
def test_conv_stem_fused(shape, has_c):
    """Darknet-53 layers 0-2 in one launch (csrc/conv_stem.hip) against the oracle's conv chain with
    the same bf16 roundings (image, weights, and each layer's stored output)."""
    B, H, W = shape
    rng = np.random.default_rng(100 + H + W)
    x = rng.random((B, H, W, 3), dtype=np.float32)
    ks = [(rng.standard_normal((3, 3, 3, 32)) * 0.4).astype(np.float32),
          (rng.standard_normal((3, 3, 32, 64)) * 0.1).astype(np.float32),
          (rng.standard_normal((1, 1, 64, 32)) * 0.2).astype(np.float32)]
    bs = [rng.standard_normal(32).astype(np.float32) * 0.2, rng.standard_normal(64).astype(np.float32) * 0.2,
          rng.standard_normal(32).astype(np.float32) * 0.2]
    ks = [gpu_util.bf16_round(k) for k in ks]
    n = 3 if has_c else 2
    leaky = (True, True, False)
    got1, got2 = gpu_util.run_stem(x, ks[:n], bs[:n], leaky=leaky, out_pad=(8, 16))
    a = gpu_util.bf16_round(_ref(gpu_util.bf16_round(x), ks[0], bs[0], 1, leaky[0]))
    want1 = _ref(a, ks[1], bs[1], 2, leaky[1])
    assert got1.shape == want1.shape
    assert np.all(np.abs(got1 - want1) <= 1e-2 * np.maximum(1.0, np.abs(want1)))
    if has_c:
        # conv C reads the bf16 tensor the kernel stored
        want2 = _ref(got1, ks[2], bs[2], 1, leaky[2])
        assert np.all(np.abs(got2 - want2) <= 1e-2 * np.maximum(1.0, np.abs(want2)))
    else:
        assert got2 is None


def test_conv_stem_rejects_bad_arguments():
    import ctypes
    import torch
    from yolo_amd import _lib
    lib = _lib.load()
    d = _lib.StemDesc()
    d.batch, d.in_h, d.in_w, d.c0, d.c1, d.c2 = 1, 32, 32, 32, 64, 0
    d.n0_pad, d.k1_pad, d.out1_ld = 32, 320, 64
    t = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    p = t.data_ptr()
    assert lib.y3_conv_stem(ctypes.byref(d), p, p, p, p, p, None, None, p, None, None) == 0
    d.c0 = 16
    assert lib.y3_conv_stem(ctypes.byref(d), p, p, p, p, p, None, None, p, None, None) == -1
    d.c0, d.c2 = 32, 32
    assert lib.y3_conv_stem(ctypes.byref(d), p, p, p, p, p, None, None, p, None, None) == -1    # conv C without weights
    d.c2 = 0
    assert lib.y3_conv_stem(ctypes.byref(d), p, p, p, p, p, None, None, p + 2, None, None) == -2
    torch.cuda.synchronize()


def test_conv_rejects_bad_arguments():
    import ctypes
    import torch
    from yolo_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = 1, 1, 8, 8, 32, 32
    d.out_h, d.out_w, d.out_c, d.out_ld, d.ksize, d.stride, d.k_pad, d.n_pad = 8, 8, 64, 64, 3, 1, 320, 64
    t = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = t.data_ptr()
    assert lib.y3_conv2d_fused(ctypes.byref(d), p, p, p, None, p, p, None) == 0
    d.out_h = 9
    assert lib.y3_conv2d_fused(ctypes.byref(d), p, p, p, None, p, p, None) == -1
    d.out_h = 8
    assert lib.y3_conv2d_fused(ctypes.byref(d), p + 2, p, p, None, p, p, None) == -2      # misaligned input
    d.k_pad = 288
    assert lib.y3_conv2d_fused(ctypes.byref(d), p, p, p, None, p, p, None) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 38, 128, 256, 3, 1, True), (1, 19, 512, 1024, 3, 2, True), (3, 13, 256, 512, 1, 1, False),
                                  (2, 20, 32, 256, 3, 1, True), (1, 9, 64, 128, 1, 1, True)],
                         ids=str)
def test_conv_tile256_equals_tile128(case, dtype):
    """Every tile variant against the oracle, and against each other: variants that share an epilogue
    sum each output in the same order -> identical bits.  Two epilogue families exist in bf16 mode:
    the register epilogue (bias is the accumulator's initial value) and the LDS epilogue (bias added
    last; fp32 mode, and debug bit 64); the ring kernels (257, debug bit 4) have both as well."""
    from yolo_amd import _lib
    lib = _lib.load()
    B, H, C, O, ks, stride, leaky = case
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, H, H, C)).astype(np.float32)
    k = (rng.standard_normal((ks, ks, C, O)) * np.sqrt(2.0 / (ks * ks * C))).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    r = rng.standard_normal((B,) + _ref(x, k, b, stride, leaky).shape[1:]).astype(np.float32)
    if dtype == "bf16":
        x, k, r = gpu_util.bf16_round(x), gpu_util.bf16_round(k), gpu_util.bf16_round(r)
    # low half: tile; high half: experiment bits (4 = ring-pipelined main loop, 2 = no interleave, 64 = LDS epilogue)
    reg_family = (128, 192, 256, (2 << 16) | 256, 257, (4 << 16) | 128, (4 << 16) | 256)
    lds_family = ((64 << 16) | 128, (64 << 16) | 192, (64 << 16) | 256, (64 << 16) | 257, (68 << 16) | 128, (68 << 16) | 256)
    want = _ref(x, k, b, stride, leaky, residual=r)
    tol = 1e-3 if dtype == "fp32" else 1e-2 * np.maximum(1.0, np.abs(want))
    outs = {}
    with _lib.experimental() as lib:          # libyolo3hip_exp.so: the product library has no tile override
        for res in (r, None):
            for tile in reg_family + lds_family:
                _lib.check(lib.y3_set_conv_tile(tile))
                outs[tile] = gpu_util.run_conv(x, k, b, stride, dtype, leaky=leaky, residual=res)
            w = want if res is not None else _ref(x, k, b, stride, leaky)
            t = tol if res is not None else (1e-3 if dtype == "fp32" else 1e-2 * np.maximum(1.0, np.abs(w)))
            for tile in outs:
                assert np.all(np.abs(outs[tile] - w) <= t), hex(tile)
            for fam in (reg_family, lds_family):
                for tile in fam:
                    np.testing.assert_array_equal(outs[fam[0]], outs[tile])
            if dtype == "fp32":      # one epilogue only
                np.testing.assert_array_equal(outs[reg_family[0]], outs[lds_family[0]])
    # ... and the product library's own choice for this shape gives the same bits as its family
    dflt = gpu_util.run_conv(x, k, b, stride, dtype, leaky=leaky, residual=None)
    np.testing.assert_array_equal(dflt, outs[reg_family[0]] if dtype == "bf16" else outs[lds_family[0]])


def test_conv_cout64_tile_variants_agree():
    """33..64 output channels with many pixels: the default 256x64 tile (260), the 128x64 tile (64) and the ring
    form (261) against the oracle and bit for bit against each other (same register epilogue, same k order)."""
    from yolo_amd import _lib
    lib = _lib.load()
    B, H, C, O, ks = 2, 264, 32, 64, 3                 # M = 139 392 >= 256 * 512: the default takes the 256x64 tile
    rng = np.random.default_rng(17)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, H, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((ks, ks, C, O)) * np.sqrt(2.0 / (ks * ks * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    r = gpu_util.bf16_round(rng.standard_normal((B, H, H, O)).astype(np.float32))
    want = _ref(x, k, b, 1, True, residual=r)
    outs = {}
    with _lib.experimental() as lib:
        for tile in (0, 64, 261):
            _lib.check(lib.y3_set_conv_tile(tile))
            outs[tile] = gpu_util.run_conv(x, k, b, 1, "bf16", leaky=True, residual=r)
            assert np.all(np.abs(outs[tile] - want) <= 1e-2 * np.maximum(1.0, np.abs(want))), tile
    np.testing.assert_array_equal(outs[0], outs[64])
    np.testing.assert_array_equal(outs[0], outs[261])
    np.testing.assert_array_equal(outs[0], gpu_util.run_conv(x, k, b, 1, "bf16", leaky=True, residual=r))    # product library


@pytest.mark.parametrize("H,W,res,leaky,pads", [(304, 304, True, True, ((0, 0), (0, 0))), (300, 304, False, True, ((32, 8), (64, 16))),
                                                 (310, 288, True, False, ((0, 0), (0, 0)))])
def test_conv3x3_c32_kernel_equals_the_implicit_gemm(H, W, res, leaky, pads):
    """3x3 32 -> 64 on a large map (Darknet-53 layer 3) runs on conv3x3_c32_kernel (weights in registers, 18 x 18 input
    patch resident in LDS): y3_conv2d_fused picks it by shape.  Against the oracle, and bit for bit against the
    implicit-GEMM tiles it replaces (same MFMA chain per output): heights that are not tile multiples, channel slices
    of wider buffers, with and without shortcut / leaky."""
    from yolo_amd import _lib
    B, C, O = 3, 32, 64
    rng = np.random.default_rng(H + W)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, W, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((3, 3, C, O)) * np.sqrt(2.0 / (9 * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    r = gpu_util.bf16_round(rng.standard_normal((B, H, W, O)).astype(np.float32)) if res else None
    want = _ref(x, k, b, 1, leaky, residual=r)
    kw = dict(leaky=leaky, residual=r, in_pad=pads[0], out_pad=pads[1])
    got = gpu_util.run_conv(x, k, b, 1, "bf16", **kw)
    assert np.all(np.abs(got - want) <= 1e-2 * np.maximum(1.0, np.abs(want)))
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = _lib.Y3_BF16, B, H, W, C, 32
    d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld, d.ksize, d.stride, d.k_pad, d.n_pad = H, W, O, 64, 64, 3, 1, 320, 64
    name = ctypes.create_string_buffer(96)
    _lib.check(lib.y3_conv_kernel_name(ctypes.byref(d), name, 96))
    assert name.value.decode() == "conv3x3_c32_kernel"
    with _lib.experimental() as exp:
        for tile in (260, 64):
            _lib.check(exp.y3_set_conv_tile(tile))
            np.testing.assert_array_equal(gpu_util.run_conv(x, k, b, 1, "bf16", **kw), got)


PP_CASES = [
    # B, H, W, Cin, Cout, ks, stride, residual, leaky, (in_pad, out_pad)
    (16, 76, 76, 128, 256, 3, 1, True, True, ((0, 0), (0, 0))),       # the 76x76 residual-stage layer: 361 tiles on 256 workgroups
    (5, 76, 76, 128, 256, 3, 1, False, True, ((128, 0), (0, 128))),   # channel slices of wider buffers (route)
    (9, 38, 38, 256, 512, 3, 1, True, False, ((0, 0), (0, 0))),       # two channel tiles, linear activation
    (48, 19, 19, 512, 1024, 3, 1, True, True, ((0, 0), (0, 0))),      # four channel tiles, 72 k-steps, 272 tiles, ragged last pixel tile
    (6, 76, 76, 128, 256, 3, 2, False, True, ((0, 0), (0, 0))),       # stride 2 (top/left padding only)
    (7, 52, 52, 1024, 512, 1, 1, False, True, ((0, 0), (0, 0))),      # 1x1, 16 k-steps
    (12, 40, 24, 256, 256, 1, 1, True, True, ((0, 0), (8, 8))),       # 1x1 with the minimum of 4 k-steps
    (3, 33, 47, 96, 256, 3, 1, True, True, ((0, 0), (0, 0))),         # Cin = 96: k-steps straddle taps, K padded 864 -> 896
    (1, 16, 16, 128, 256, 3, 1, True, True, ((0, 0), (0, 0))),        # a single tile
    (4, 40, 40, 64, 256, 3, 1, True, True, ((0, 0), (0, 0))),         # 9 k-steps (odd): stays on conv_igemm_kernel
    (32, 38, 38, 256, 512, 3, 1, True, True, ((0, 0), (0, 0))),       # 362 tiles: the balanced split cuts tiles of two channel tiles
    (64, 19, 19, 512, 1024, 3, 1, False, True, ((0, 0), (0, 0))),     # 364 tiles, 1.4 per workgroup: head + tail and nothing between
]


@pytest.mark.parametrize("case", PP_CASES, ids=str)
def test_conv_pingpong_kernel_equals_the_implicit_gemm(case):
    """conv_pp_kernel (csrc/conv_pp.hip: the 256 x 256 bf16 tile with a four-phase, two-wave-group main loop and a
    loader that runs ahead across k-steps and tiles): against the oracle, and bit for bit against conv_igemm_kernel's
    256 x 256 and 128 x 128 tiles (same sums in the same order) -- residual, leaky / linear, stride 2, 1x1, channel
    slices, ragged tiles, one to four channel tiles, workgroups that walk several tiles, both tile orders."""
    from yolo_amd import _lib
    B, H, W, C, O, ks, stride, res, leaky, pads = case
    rng = np.random.default_rng(B * 1000 + H + C)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, W, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((ks, ks, C, O)) * np.sqrt(2.0 / (ks * ks * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    want = _ref(x, k, b, stride, leaky)
    r = gpu_util.bf16_round(rng.standard_normal(want.shape).astype(np.float32)) if res else None
    if res:
        want = want + r
    kw = dict(leaky=leaky, residual=r, in_pad=pads[0], out_pad=pads[1])
    outs = {}
    with _lib.experimental() as exp:
        for tile in (512, 256, 128):
            _lib.check(exp.y3_set_conv_tile(tile))
            outs[tile] = gpu_util.run_conv(x, k, b, stride, "bf16", **kw)
        _lib.check(exp.y3_set_conv_tile(512))
        outs["rev"] = gpu_util.run_conv(x, k, b, stride, "bf16", reverse=True, **kw)
        # y3_conv2d_fused_ws: with a workspace the k-steps are dealt evenly and a tile cut between two workgroups is summed
        # in order (head by one, tail by the other from the first one's fp32 accumulators): same bits, both tile orders,
        # and also when every tail gives up on its predecessor and recomputes the head itself (test hook, flag bit 30)
        info = {}
        outs["ws"] = gpu_util.run_conv(x, k, b, stride, "bf16", ws=True, info=info, **kw)
        outs["ws_rev"] = gpu_util.run_conv(x, k, b, stride, "bf16", ws=True, reverse=True, **kw)
        outs["ws_recompute"] = gpu_util.run_conv(x, k, b, stride, "bf16", ws=True, extra_flags=1 << 30, **kw)
    tiles = -(-(B * want.shape[1] * want.shape[2]) // 256) * (-(-O // 256))
    cut = tiles > 256 and tiles % 256 != 0 and C * ks * ks % 128 == 0
    assert (info["slab_bytes"] > 0) == cut and (info["flags_raised"] > 100) == cut, (tiles, info)
    assert np.all(np.abs(outs[512] - want) <= 1e-2 * np.maximum(1.0, np.abs(want)))
    for key in (256, 128, "rev", "ws", "ws_rev", "ws_recompute"):
        np.testing.assert_array_equal(outs[512], outs[key], err_msg=str(key))


@pytest.mark.parametrize("B,H,C,O,stride,res,split", [(64, 19, 512, 1024, 1, True, False), (64, 38, 256, 512, 1, True, True)])
def test_conv_workspace_with_the_library_own_tile_choice(B, H, C, O, stride, res, split):
    """The product library's own choice (no tile forced): y3_conv2d_fused_ws deals the k-steps evenly on the launches that
    run on the 256 x 256 tile anyway (38 x 38: 2.8 rounds) and ignores the workspace on the others (19 x 19 at batch 64
    keeps its 192-pixel tiles: y3_conv_workspace_bytes says 0) -- the same bits either way."""
    rng = np.random.default_rng(B + H + C)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, H, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((3, 3, C, O)) * np.sqrt(2.0 / (9 * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    Ho = H // stride
    r = gpu_util.bf16_round(rng.standard_normal((B, Ho, Ho, O)).astype(np.float32)) if res else None
    plain = gpu_util.run_conv(x, k, b, stride, "bf16", leaky=True, residual=r)
    info = {}
    with_ws = gpu_util.run_conv(x, k, b, stride, "bf16", leaky=True, residual=r, ws=True, info=info)
    assert (info["slab_bytes"] > 0) == split and (info["flags_raised"] > 100) == split, info
    np.testing.assert_array_equal(plain, with_ws)


@pytest.mark.parametrize("B,H,C,O", [(64, 76, 128, 256), (40, 76, 128, 256), (29, 76, 128, 256), (64, 38, 256, 512), (37, 38, 256, 512)])
def test_conv_balanced_split_after_whole_rounds(B, H, C, O):
    """Launches of several rounds: whole rounds first, then the remaining tiles' k-steps dealt evenly -- half a tile to a
    tile per workgroup, so tiles are cut into up to three pieces (first k-steps, middle, last k-steps; tests/
    test_pp_partition.py restates the work list).  Bit for bit the unsplit kernel's output, both tile orders and with
    every piece that could continue a predecessor's sums recomputing them instead; sanity against torch's fp32 conv."""
    import torch
    from yolo_amd import _lib
    rng = np.random.default_rng(B + H)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, H, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((3, 3, C, O)) * np.sqrt(2.0 / (9 * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    r = gpu_util.bf16_round(rng.standard_normal((B, H, H, O)).astype(np.float32))
    outs, info = {}, {}
    with _lib.experimental() as exp:
        _lib.check(exp.y3_set_conv_tile(512))
        outs["plain"] = gpu_util.run_conv(x, k, b, 1, "bf16", residual=r)
        outs["ws"] = gpu_util.run_conv(x, k, b, 1, "bf16", residual=r, ws=True, info=info)
        outs["ws_rev"] = gpu_util.run_conv(x, k, b, 1, "bf16", residual=r, ws=True, reverse=True)
        outs["ws_recompute"] = gpu_util.run_conv(x, k, b, 1, "bf16", residual=r, ws=True, extra_flags=1 << 30)
    assert info["slab_bytes"] > 0 and info["flags_raised"] > 100, info
    for key in ("ws", "ws_rev", "ws_recompute"):
        np.testing.assert_array_equal(outs["plain"], outs[key], err_msg=key)
    y = torch.nn.functional.conv2d(torch.from_numpy(x).cuda().permute(0, 3, 1, 2), torch.from_numpy(k).cuda().permute(3, 2, 0, 1), torch.from_numpy(b).cuda(), padding=1)
    want = (torch.nn.functional.leaky_relu(y, 0.1).permute(0, 2, 3, 1) + torch.from_numpy(r).cuda()).cpu().numpy()
    assert np.all(np.abs(outs["plain"] - want) <= 1e-2 * np.maximum(1.0, np.abs(want)))


PP192_CASES = [
    # B, H, W, Cin, Cout, ks, stride, residual, leaky, (in_pad, out_pad)
    (48, 19, 19, 512, 1024, 3, 1, True, True, ((0, 0), (0, 0))),      # the 19x19 residual-stage layer: 364 tiles of 192 pixels, ragged last one
    (64, 19, 19, 1024, 512, 1, 1, False, True, ((0, 0), (0, 0))),     # its 1x1 bottleneck: 16 k-steps, two channel tiles
    (10, 38, 38, 512, 1024, 3, 2, False, False, ((0, 0), (0, 0))),    # stride 2, linear
    (40, 38, 38, 256, 512, 3, 1, True, True, ((256, 0), (0, 512))),   # 602 tiles: several per workgroup; channel slices of wider buffers
    (1, 13, 13, 128, 256, 3, 1, True, True, ((0, 0), (0, 0))),        # a single ragged tile (169 pixels)
    (3, 26, 26, 64, 256, 1, 1, True, True, ((0, 0), (8, 8))),         # K = 64: below four k-steps, stays on conv_igemm_kernel
    (2, 24, 24, 96, 256, 3, 1, False, True, ((0, 0), (0, 0))),        # Cin = 96 (k-steps straddle taps): stays on conv_igemm_kernel
]


@pytest.mark.parametrize("case", PP192_CASES, ids=str)
def test_conv_pingpong_192_pixel_tile_equals_the_implicit_gemm(case):
    """conv_pp_kernel<3>: the two-wave-group main loop on 192-pixel tiles (LDS pixel rows as [half][group][48], a pass and a
    half per half-tile, 7 LDS-DMA instructions per wave and k-step) -- against the oracle, and bit for bit against
    conv_igemm_kernel's 192 x 256 and 128 x 128 tiles (same sums in the same order), both tile orders."""
    from yolo_amd import _lib
    B, H, W, C, O, ks, stride, res, leaky, pads = case
    rng = np.random.default_rng(B * 1000 + H + C + 1)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, W, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((ks, ks, C, O)) * np.sqrt(2.0 / (ks * ks * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    want = _ref(x, k, b, stride, leaky)
    r = gpu_util.bf16_round(rng.standard_normal(want.shape).astype(np.float32)) if res else None
    if res:
        want = want + r
    kw = dict(leaky=leaky, residual=r, in_pad=pads[0], out_pad=pads[1])
    outs = {}
    with _lib.experimental() as exp:
        for tile in (704, 192, 128):
            _lib.check(exp.y3_set_conv_tile(tile))
            outs[tile] = gpu_util.run_conv(x, k, b, stride, "bf16", **kw)
        _lib.check(exp.y3_set_conv_tile(704))
        outs["rev"] = gpu_util.run_conv(x, k, b, stride, "bf16", reverse=True, **kw)
    assert np.all(np.abs(outs[704] - want) <= 1e-2 * np.maximum(1.0, np.abs(want)))
    for key in (192, 128, "rev"):
        np.testing.assert_array_equal(outs[704], outs[key], err_msg=str(key))


def test_conv_pingpong_is_what_the_big_layers_run_on():
    """y3_conv_kernel_name: the product library sends the 256 x 256-tile launches with a bf16 output to conv_pp_kernel
    (the fp32 heads keep conv_igemm_kernel's LDS epilogue with the fused decode)."""
    from yolo_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = _lib.Y3_BF16, 64, 76, 76, 128, 128
    d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld, d.ksize, d.stride, d.k_pad, d.n_pad = 76, 76, 256, 256, 256, 3, 1, 1152, 256
    d.flags = _lib.CONV_LEAKY | _lib.CONV_RESIDUAL
    name = ctypes.create_string_buffer(96)
    _lib.check(lib.y3_conv_kernel_name(ctypes.byref(d), name, 96))
    assert name.value.decode() == "conv_pp_kernel<4>"
    d.flags = _lib.CONV_OUT_F32
    _lib.check(lib.y3_conv_kernel_name(ctypes.byref(d), name, 96))
    assert name.value.decode().startswith("conv_igemm_kernel<true, 256, 256")
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld, d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld, d.k_pad, d.n_pad = 64, 19, 19, 512, 512, 19, 19, 1024, 1024, 1024, 4608, 1024
    d.flags = _lib.CONV_LEAKY | _lib.CONV_RESIDUAL
    _lib.check(lib.y3_conv_kernel_name(ctypes.byref(d), name, 96))
    assert name.value.decode() == "conv_pp_kernel<3>"           # 19 x 19 at batch 64: 192-pixel tiles, the same main loop


@pytest.mark.parametrize("B,H,W,leaky,pads", [(3, 304, 304, True, ((0, 0), (0, 0))), (4, 296, 280, True, ((64, 0), (128, 16))),
                                               (3, 306, 304, False, ((0, 0), (0, 0)))])
def test_conv3x3s2_c64_kernel_equals_the_implicit_gemm(B, H, W, leaky, pads):
    """3x3 stride 2, 64 -> 128 on a large map (Darknet-53 layer 5) runs on conv3x3s2_c64_kernel (weights in registers,
    9 x 33 input patch resident in LDS): y3_conv2d_fused picks it by shape.  Against the oracle, and bit for bit against
    the implicit-GEMM tiles it replaces (same MFMA chain per output): output widths / heights that are not tile
    multiples, channel slices of wider buffers, with and without leaky."""
    from yolo_amd import _lib
    C, O = 64, 128
    rng = np.random.default_rng(H + W)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, W, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((3, 3, C, O)) * np.sqrt(2.0 / (9 * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    want = _ref(x, k, b, 2, leaky)
    kw = dict(leaky=leaky, in_pad=pads[0], out_pad=pads[1])
    got = gpu_util.run_conv(x, k, b, 2, "bf16", **kw)
    assert got.shape == want.shape == (B, H // 2, W // 2, O)
    assert np.all(np.abs(got - want) <= 1e-2 * np.maximum(1.0, np.abs(want)))
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = _lib.Y3_BF16, B, H, W, C, 64
    d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld, d.ksize, d.stride, d.k_pad, d.n_pad = H // 2, W // 2, O, 128, 128, 3, 2, 576, 128
    name = ctypes.create_string_buffer(96)
    _lib.check(lib.y3_conv_kernel_name(ctypes.byref(d), name, 96))
    assert name.value.decode() == "conv3x3s2_c64_kernel"
    with _lib.experimental() as exp:
        for tile in (257, 128):
            _lib.check(exp.y3_set_conv_tile(tile))
            np.testing.assert_array_equal(gpu_util.run_conv(x, k, b, 2, "bf16", **kw), got)


@pytest.mark.parametrize("B,H,W,C,O,leaky,pads", [(48, 38, 38, 512, 256, True, ((0, 0), (0, 0))), (12, 76, 76, 256, 128, True, ((0, 0), (0, 0))),
                                                   (3, 150, 153, 256, 128, False, ((128, 0), (64, 192))),
                                                   (5, 117, 119, 512, 256, True, ((0, 256), (256, 0))), (12, 76, 76, 384, 128, True, ((0, 0), (0, 0)))])
def test_conv1x1_regw_kernel_equals_the_implicit_gemm(B, H, W, C, O, leaky, pads):
    """The residual stages' 1x1 bottlenecks (512 -> 256, 256 -> 128) on large launches run on conv1x1_regw_kernel (the
    whole weight matrix in registers, 64-pixel tiles of activations through LDS): y3_conv2d_fused picks it by shape.
    Against the oracle, and bit for bit against the implicit-GEMM tiles it replaces (same MFMA chain per output): pixel
    counts that are not tile multiples, channel slices of wider buffers (route inputs / concat outputs)."""
    from yolo_amd import _lib
    rng = np.random.default_rng(H + W + C)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, W, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((1, 1, C, O)) * np.sqrt(2.0 / C)).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    want = _ref(x, k, b, 1, leaky)
    kw = dict(leaky=leaky, in_pad=pads[0], out_pad=pads[1])
    got = gpu_util.run_conv(x, k, b, 1, "bf16", **kw)
    assert np.all(np.abs(got - want) <= 1e-2 * np.maximum(1.0, np.abs(want)))
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = _lib.Y3_BF16, B, H, W, C, C
    d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld, d.ksize, d.stride, d.k_pad, d.n_pad = H, W, O, O, O, 1, 1, C, O
    name = ctypes.create_string_buffer(96)
    _lib.check(lib.y3_conv_kernel_name(ctypes.byref(d), name, 96))
    assert name.value.decode() == "conv1x1_regw_kernel<%d, %d>" % (C, O)
    with _lib.experimental() as exp:
        for tile in ((256, 192, 128) if O == 256 else (257, 128)):
            _lib.check(exp.y3_set_conv_tile(tile))
            np.testing.assert_array_equal(gpu_util.run_conv(x, k, b, 1, "bf16", **kw), got)


@pytest.mark.parametrize("case", ["ring_ups", "ring_1x1_res", "head_f32_256tile"])
def test_conv_bench_scale_defaults(case):
    """Kernel choices that only engage at bench-scale pixel counts (M >= 65 536), against the oracle: the
    256x128 ring tile with the 2x2 upsampling store (LDS epilogue inside the ring kernel), the same tile with
    a shortcut (register epilogue, early residual loads), and an fp32-output head on the 256-row tiles."""
    rng = np.random.default_rng(23)
    if case == "ring_ups":
        B, H, C, O, ks, kw = 12, 76, 256, 128, 1, dict(ups=True, out_pad=(0, 256))
    elif case == "ring_1x1_res":
        B, H, C, O, ks, kw = 12, 76, 256, 128, 1, dict()
    else:
        B, H, C, O, ks, kw = 12, 76, 256, 255, 1, dict(out_f32=True, leaky=False)
    x = gpu_util.bf16_round(rng.standard_normal((B, H, H, C)).astype(np.float32))
    k = gpu_util.bf16_round((rng.standard_normal((ks, ks, C, O)) * np.sqrt(2.0 / (ks * ks * C))).astype(np.float32))
    b = rng.standard_normal(O).astype(np.float32)
    r = gpu_util.bf16_round(rng.standard_normal((B, H, H, O)).astype(np.float32)) if case == "ring_1x1_res" else None
    leaky = kw.pop("leaky", True)
    got = gpu_util.run_conv(x, k, b, 1, "bf16", leaky=leaky, residual=r, **kw)
    want = _ref(x, k, b, 1, leaky, residual=r, ups=kw.get("ups", False))
    assert got.shape == want.shape
    tol = (2e-3 if case == "head_f32_256tile" else 1e-2) * np.maximum(1.0, np.abs(want))
    assert np.all(np.abs(got - want) <= tol)


def test_conv_fuzz_shapes():
    """Seeded sweep over odd shapes against the oracle: sizes that are not tile multiples, channel counts that are
    not multiples of 16 / 32, both strides, both kernel sizes, shortcut / upsample / fp32-out variants, both dtypes,
    with and without split-K."""
    from yolo_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(4242)
    n_cases = n_split = 0
    for it in range(36):
        dtype = "bf16" if it % 3 else "fp32"
        ks = int(rng.choice([1, 3]))
        stride = int(rng.choice([1, 2])) if ks == 3 else 1
        H = int(rng.integers(5, 41))
        if stride == 2 and H % 2:
            H += 1
        B = int(rng.integers(1, 5))
        C = int(rng.choice([8, 16, 24, 32, 40, 64, 96, 128, 136]))
        O = int(rng.choice([8, 18, 32, 33, 64, 75, 128, 130, 255, 256, 264]))
        if it >= 28:                                # long k, few pixels: the shapes split-K is for
            ks, C, H, B = 3, int(rng.choice([256, 512, 1024])), int(rng.integers(9, 20)), 1
            if stride == 2 and H % 2:
                H += 1
        mode = int(rng.integers(0, 4))              # 0 plain, 1 shortcut, 2 upsample, 3 fp32 out (linear)
        x = rng.standard_normal((B, H, H, C)).astype(np.float32)
        k = (rng.standard_normal((ks, ks, C, O)) * np.sqrt(2.0 / (ks * ks * C))).astype(np.float32)
        b = rng.standard_normal(O).astype(np.float32)
        leaky = mode != 3
        want_shape = _ref(x[:1], k, b, stride, leaky).shape[1:]
        r = rng.standard_normal((B,) + want_shape).astype(np.float32) if mode == 1 else None
        if dtype == "bf16":
            x, k = gpu_util.bf16_round(x), gpu_util.bf16_round(k)
            r = gpu_util.bf16_round(r) if r is not None else None
        kw = dict(ups=True) if mode == 2 else (dict(out_f32=True) if mode == 3 else {})
        want = _ref(x, k, b, stride, leaky, residual=r, ups=(mode == 2))
        tol = 1e-3 if dtype == "fp32" else (2e-3 if mode == 3 else 1e-2) * np.maximum(1.0, np.abs(want))
        for sk in (False, True):
            info = {}
            got = gpu_util.run_conv(x, k, b, stride, dtype, leaky=leaky, residual=r, splitk=sk, info=info, **kw)
            n_split += info.get("splits", 1) > 1
            assert got.shape == want.shape, (it, dtype, ks, stride, H, B, C, O, mode, sk)
            assert np.all(np.abs(got - want) <= tol), (it, dtype, ks, stride, H, B, C, O, mode, sk, float(np.abs(got - want).max()))
        n_cases += 1
    assert n_cases == 36 and n_split >= 4
