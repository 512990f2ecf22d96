"""The C-ABI library loads and exports every symbol include/yolo3hip.h declares; host-only
entry points behave (no GPU compute is attempted here)."""
import ctypes
import os
import re

import pytest

from yolo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="yolo3hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(y3_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert "y3_conv2d_fused" in names and "y3_nms" in names and len(names) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libyolo3hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_product_library_has_no_knobs():
    """The tile / phase / CU / decode-variant setters (and the timing paths behind them) exist only in the
    -DY3_EXPERIMENTS flavour: libyolo3hip.so exports none of them, libyolo3hip_exp.so exports the product ABI plus
    exactly what include/yolo3hip_experimental.h declares."""
    knobs = _declared("yolo3hip_experimental.h")
    assert knobs == sorted(_lib.EXPERIMENTAL_SIGNATURES) and "y3_set_conv_tile" in knobs
    lib = ctypes.CDLL(_lib.LIB_PATH)
    exp = ctypes.CDLL(_lib.EXP_LIB_PATH)
    for n in knobs:
        assert not hasattr(lib, n), "libyolo3hip.so exports the experiment knob %s" % n
        assert hasattr(exp, n)
    for n in _declared():
        assert hasattr(exp, n)
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "y3_conv2d_fused" in syms and "y3_set_conv" not in syms and "y3_decode_variant" not in syms


def test_version_and_layout(lib):
    assert lib.y3_version() == 103
    assert _lib.conv_weight_layout(_lib.Y3_BF16, 3, 32, 3, in_f32=True) == (1, 27, 32)
    assert _lib.conv_weight_layout(_lib.Y3_BF16, 32, 64, 3) == (0, 320, 64)
    assert _lib.conv_weight_layout(_lib.Y3_BF16, 1024, 255, 1) == (0, 1024, 256)
    assert _lib.conv_weight_layout(_lib.Y3_F32, 512, 1024, 3) == (0, 4608, 1024)
    assert _lib.conv_weight_layout(_lib.Y3_F32, 3, 16, 3) == (1, 27, 32)
    assert lib.y3_nms_lds_bytes(22743) == 22743 * 5 + 16


def test_errors_surface_as_exceptions(lib):
    with pytest.raises(ValueError, match="ksize"):
        _lib.conv_weight_layout(_lib.Y3_BF16, 32, 64, 5)
    d = _lib.ConvDesc()
    rc = lib.y3_conv2d_fused(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.y3_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libyolo3hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_committed_traffic_file_names_the_bench_kernels():
    """bench.py takes roofline.traffic from profiles/traffic.json by kernel-name prefix: the file must exist, point at a
    committed summary, and carry the kernels of the default plan (so that `traffic` is never silently null)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tj = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    src = tj["source"].split(" ")[0]
    assert os.path.exists(os.path.join(root, src)), src
    names = tj["hbm_bytes_per_launch"]
    for prefix in ("conv_igemm_kernel<true, 256, 256, 2, 4", "conv_igemm_kernel<true, 192, 256, 2, 4", "conv_stem_kernel",
                   "conv_igemm_ring_kernel<true, 256, 128"):
        hits = [v for k, v in names.items() if k.startswith(prefix)]      # (with and without the fused head decode)
        assert 1 <= len(hits) <= 2 and hits[0] > 0, prefix
