"""Letterbox pre-process (reference utils/utils.py:7-40 + inference.py:242).
CPU: the oracle's properties and its agreement with the package's host resize_image.
-m gpu: y3_letterbox_u8 against the oracle, bit for bit (uint8 pixels and the fp32 /255 values)."""
import numpy as np
import pytest

from oracle import letterbox as olb


def test_oracle_letterbox_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    # no resize when the scaled size equals the image: plain centre pad with 128 (utils.py:34-38)
    out = olb.resize_image(img, (37, 60))
    assert out.shape == (37, 60, 3) and np.all(out[:, :3] == 128) and np.all(out[:, 56:] == 128)
    np.testing.assert_array_equal(out[:, 3:56], img)
    # constant images stay constant under the cubic (weights sum to 1); pad is 128
    flat = np.full((20, 30, 3), 77, dtype=np.uint8)
    out = olb.resize_image(flat, (64, 64))
    new_h = int(20 * min(64 / 30, 64 / 20))
    top = (64 - new_h) // 2
    assert np.all(out[top:top + new_h] == 77) and np.all(out[:top] == 128) and np.all(out[top + new_h:] == 128)
    # truncation of the new size (utils.py:22-23): 100x75 -> 416: scale 4.16 -> 416 x 312
    out = olb.resize_image(np.zeros((75, 100, 3), dtype=np.uint8), (416, 416))
    assert np.all(out[:52] == 128) and np.all(out[52:364] == 0) and np.all(out[364:] == 128)
    # hand-computed 1-D known answer: 2 pixels [0, 100] upsampled to 4 (a = -0.75, replicate border)
    row = np.array([[[0, 0, 0], [100, 100, 100]]], dtype=np.uint8)
    up = olb.resize_image(np.repeat(row, 2, axis=0), (4, 4))
    # src x = -0.25, 0.25, 0.75, 1.25 -> t = .75, .25, .75, .25 around bases -1, 0, 0, 1
    w = lambda t: np.array(olb.cubic_weights(t))
    exp = [np.dot(w(0.75), [0, 0, 0, 100]), np.dot(w(0.25), [0, 0, 100, 100]), np.dot(w(0.75), [0, 0, 100, 100]),
           np.dot(w(0.25), [0, 100, 100, 100])]
    np.testing.assert_array_equal(up[0, :, 0], np.clip(np.floor(np.array(exp) + 0.5), 0, 255).astype(np.uint8))
    x = olb.network_input(img, (64, 64))
    assert x.dtype == np.float32 and x.shape == (1, 64, 64, 3) and x.max() <= 1.0


def test_host_resize_image_equals_oracle():
    from yolo_amd.utils.utils import resize_image
    rng = np.random.default_rng(1)
    for (h, w), s in (((48, 64), 96), ((200, 120), 64), ((33, 33), 33), ((90, 160), 128)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a, b = resize_image(img, (s, s)), olb.resize_image(img, (s, s))
        assert a.shape == b.shape
        assert np.abs(a.astype(int) - b.astype(int)).max() <= 1          # einsum vs ordered sums: at most a rounding tie
        assert (a != b).mean() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("hw,size", [((480, 640), (608, 608)), ((375, 500), (416, 416)), ((1080, 1920), (608, 608)),
                                     ((100, 37), (320, 320)), ((64, 64), (64, 64)), ((300, 200), (416, 608))], ids=str)
def test_letterbox_gpu_bit_exact(hw, size):
    import torch
    from yolo_amd.utils.utils import letterbox_gpu
    rng = np.random.default_rng(hw[0] + size[0])
    img = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    got = letterbox_gpu(img, size)
    torch.cuda.synchronize()
    want = olb.network_input(img, size)[0]
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # into a slot of a batch tensor, from a strided device image
    batch = torch.zeros((2,) + size + (3,), dtype=torch.float32, device="cuda")
    wide = torch.zeros(hw + (4,), dtype=torch.uint8, device="cuda")
    wide[..., :3] = torch.from_numpy(img).cuda()
    letterbox_gpu(wide[..., :3], size, out=batch[1])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(batch[1].cpu().numpy(), want)
    assert float(batch[0].abs().max()) == 0.0


@pytest.mark.gpu
def test_letterbox_gpu_rejects_bad_arguments():
    import torch
    from yolo_amd.utils.utils import letterbox_gpu
    with pytest.raises(ValueError):
        letterbox_gpu(np.zeros((8, 8), dtype=np.uint8), (16, 16))
    with pytest.raises(ValueError):
        letterbox_gpu(np.zeros((8, 8, 3), dtype=np.uint8), (16, 16), out=torch.zeros((16, 16, 3), dtype=torch.float16, device="cuda"))
