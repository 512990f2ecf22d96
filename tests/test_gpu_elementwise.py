"""-m gpu: max-pool and the un-fused fallback kernels against the oracle ops (bit-exact: they
only move or compare values; the add rounds once to the storage type)."""
import numpy as np
import pytest
import torch

import gpu_util
from oracle import darknet as od
from yolo_amd import _lib

pytestmark = pytest.mark.gpu


def _code(dtype):
    return _lib.Y3_BF16 if dtype == "bf16" else _lib.Y3_F32


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("size,stride,h", [(2, 2, 26), (2, 1, 13), (2, 2, 13), (3, 2, 17)])
def test_maxpool(dtype, size, stride, h):
    lib = _lib.load()
    x = gpu_util.bf16_round(np.random.default_rng(h).standard_normal((2, h, h, 32)).astype(np.float32))
    xt = gpu_util.to_dev(x, dtype)
    want = od.max_pool_same(x, size, stride)
    out = torch.zeros(want.shape[:3] + (48,), dtype=xt.dtype, device="cuda")
    _lib.check(lib.y3_maxpool2d(_code(dtype), xt.data_ptr(), out.data_ptr(), 2, h, h, 32, 32, 48, size, stride, gpu_util.stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out[..., :32].float().cpu().numpy(), want)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_upsample_copy_add(dtype):
    lib = _lib.load()
    rng = np.random.default_rng(5)
    x = gpu_util.bf16_round(rng.standard_normal((2, 7, 7, 64)).astype(np.float32))
    y = gpu_util.bf16_round(rng.standard_normal((2, 7, 7, 64)).astype(np.float32))
    xt, yt = gpu_util.to_dev(x, dtype), gpu_util.to_dev(y, dtype)
    st = gpu_util.stream()
    up = torch.zeros((2, 14, 14, 96), dtype=xt.dtype, device="cuda")
    _lib.check(lib.y3_upsample2x(_code(dtype), xt.data_ptr(), up.data_ptr(), 2, 7, 7, 64, 64, 96, st))
    cat = torch.zeros((2, 7, 7, 128), dtype=xt.dtype, device="cuda")
    es = xt.element_size()
    _lib.check(lib.y3_copy_channels(_code(dtype), xt.data_ptr(), cat.data_ptr(), 2 * 49, 64, 64, 128, st))
    _lib.check(lib.y3_copy_channels(_code(dtype), yt.data_ptr(), cat.data_ptr() + 64 * es, 2 * 49, 64, 64, 128, st))
    s = torch.zeros((2, 7, 7, 64), dtype=xt.dtype, device="cuda")
    _lib.check(lib.y3_shortcut_add(_code(dtype), xt.data_ptr(), yt.data_ptr(), s.data_ptr(), 2 * 49, 64, 64, 64, 64, st))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(up[..., :64].float().cpu().numpy(), od.upsample2x_nearest(x))
    np.testing.assert_array_equal(cat.float().cpu().numpy(), np.concatenate([x, y], axis=-1))
    want = x + y if dtype == "fp32" else gpu_util.bf16_round(x + y)
    np.testing.assert_array_equal(s.float().cpu().numpy(), want)


def test_casts():
    lib = _lib.load()
    x = np.random.default_rng(6).standard_normal(10001).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    h = torch.zeros(10001, dtype=torch.bfloat16, device="cuda")
    back = torch.zeros(10001, dtype=torch.float32, device="cuda")
    _lib.check(lib.y3_cast_f32_to_bf16(xt.data_ptr(), h.data_ptr(), 10001, gpu_util.stream()))
    _lib.check(lib.y3_cast_bf16_to_f32(h.data_ptr(), back.data_ptr(), 10001, gpu_util.stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(back.cpu().numpy(), gpu_util.bf16_round(x))


def test_rejects_bad_channels():
    lib = _lib.load()
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    assert lib.y3_maxpool2d(1, t.data_ptr(), t.data_ptr(), 1, 4, 4, 12, 16, 16, 2, 2, None) == -1     # 12 % 8 != 0
    assert lib.y3_upsample2x(0, t.data_ptr(), t.data_ptr() + 4, 1, 4, 4, 8, 8, 8, None) == -2         # misaligned
