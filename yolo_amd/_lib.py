"""ctypes face of libyolo3hip.so (include/yolo3hip.h).  There is no fallback: if the
HIP library is missing or a call fails, the caller gets an exception."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YOLO3HIP_LIB") or os.path.join(_HERE, "libyolo3hip.so")     # override: A/B builds
EXP_LIB_PATH = os.path.join(_HERE, "libyolo3hip_exp.so")       # -DY3_EXPERIMENTS flavour: tests and A/B tools only

Y3_F32, Y3_BF16 = 0, 1
CONV_LEAKY, CONV_RESIDUAL, CONV_UPS2X, CONV_OUT_F32, CONV_IN_F32, CONV_REVERSE, CONV_POOL2 = 1, 2, 4, 8, 16, 32, 64

c_int, c_i64, c_f32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "dtype", "batch", "in_h", "in_w", "in_c", "in_ld", "out_h", "out_w", "out_c", "out_ld", "res_ld",
        "ksize", "stride", "k_pad", "n_pad", "flags")]


class StemDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "batch", "in_h", "in_w", "c0", "c1", "c2", "n0_pad", "k1_pad", "k2_pad", "out1_ld", "out2_ld", "flags")]


class ResblockDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "dtype", "batch", "h", "w", "c", "x_ld", "out_ld", "k1_pad", "n1_pad", "k2_pad", "n2_pad", "flags")]


class DecodeDesc(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("num_scales", ctypes.c_int32), ("num_classes", ctypes.c_int32),
                ("num_anchors", ctypes.c_int32), ("grid_h", ctypes.c_int32 * 3), ("grid_w", ctypes.c_int32 * 3),
                ("head_ld", ctypes.c_int32 * 3), ("anchors", ((c_f32 * 2) * 3) * 3),
                ("stride_x", c_f32 * 3), ("stride_y", c_f32 * 3), ("score_threshold", c_f32),
                ("boxes_per_image", ctypes.c_int32)]


# every symbol include/yolo3hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "y3_conv_weight_layout": (c_int, [ctypes.POINTER(ConvDesc)]),
    "y3_conv2d_fused": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_conv_workspace_bytes": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_int64)]),
    "y3_conv2d_fused_ws": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp]),
    "y3_conv_splitk_plan": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]),
    "y3_conv2d_splitk": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, c_vp]),
    "y3_conv_stem": (c_int, [ctypes.POINTER(StemDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_resblock_tiles": (c_int, [ctypes.POINTER(ResblockDesc), ctypes.POINTER(ctypes.c_int64)]),
    "y3_resblock_fused": (c_int, [ctypes.POINTER(ResblockDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_conv_kernel_name": (c_int, [ctypes.POINTER(ConvDesc), ctypes.c_char_p, c_int]),
    "y3_maxpool2d": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "y3_upsample2x": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "y3_copy_channels": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp]),
    "y3_shortcut_add": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp]),
    "y3_decode": (c_int, [ctypes.POINTER(DecodeDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_conv_head_decode_ok": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(DecodeDesc), c_int]),
    "y3_conv2d_head_decode": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(DecodeDesc), c_int,
                                      c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_get_head": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_f32, c_f32,
                            c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_correct_boxes": (c_int, [c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_vp, c_vp]),
    "y3_expand_scores": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "y3_nms": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "y3_nms_bucketed": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_nms_bucketed_static": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "y3_nms_workspace_ints": (c_i64, [c_int, c_int]),
    "y3_nms_lds_bytes": (c_int, [c_int]),
    "y3_pack_outputs": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "y3_letterbox_u8": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "y3_pad_input": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_vp]),
    "y3_cast_f32_to_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "y3_cast_bf16_to_f32": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "y3_version": (c_int, []),
    "y3_last_error": (ctypes.c_char_p, []),
}

# include/yolo3hip_experimental.h: only libyolo3hip_exp.so exports these
EXPERIMENTAL_SIGNATURES = {
    "y3_set_conv_tile": (c_int, [c_int]),
    "y3_set_conv_phase": (c_int, [c_int]),
    "y3_set_conv_cus": (c_int, [c_int]),
    "y3_decode_variant": (c_int, [c_int]),
    "y3_nms_bitmask": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32,
                               c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
}

_lib = None
_exp = None


def _open(path, signatures):
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C yolo_amd/csrc` (there is no CPU fallback)" % path)
    # torch first: it brings its own HIP runtime (libamdhip64), and the library must bind to that one.
    # Loaded before torch, libyolo3hip.so pulls the system runtime in and every launch then fails with
    # "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def load():
    """Load the HIP library; raise (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH, SIGNATURES)
    return _lib


def load_experimental():
    """libyolo3hip_exp.so: the same kernels built with -DY3_EXPERIMENTS, plus the tile / variant setters.
    For the kernel-variant tests and the A/B tools; the product path never loads it."""
    global _exp
    if _exp is None:
        sig = dict(SIGNATURES)
        sig.update(EXPERIMENTAL_SIGNATURES)
        _exp = _open(EXP_LIB_PATH, sig)
    return _exp


class experimental(object):
    """`with _lib.experimental() as lib:` -- inside the block every `_lib.load()` (hence the whole Python
    surface: Network, PostProcessor, ...) runs on the experimental flavour; knobs are reset on exit."""

    def __enter__(self):
        global _lib
        self._saved = _lib
        _lib = load_experimental()
        return _lib

    def __exit__(self, *exc):
        global _lib
        lib = _lib
        lib.y3_set_conv_tile(0)
        lib.y3_set_conv_phase(50)
        lib.y3_set_conv_cus(0)
        lib.y3_decode_variant(0)
        _lib = self._saved
        return False


class Y3Error(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().y3_last_error().decode("utf-8", "replace")
        if rc in (-1, -2):
            raise ValueError("yolo3hip: %s" % msg)         # bad argument / alignment
        raise Y3Error("yolo3hip (code %d): %s" % (rc, msg))


def conv_weight_layout(dtype, in_c, out_c, ksize, in_f32=False):
    """(family, k_pad, n_pad) for a conv; host-only, needs no GPU."""
    d = ConvDesc()
    d.dtype, d.in_c, d.out_c, d.ksize = dtype, in_c, out_c, ksize
    d.flags = CONV_IN_F32 if in_f32 else 0
    fam = load().y3_conv_weight_layout(ctypes.byref(d))
    if fam < 0:
        check(fam)
    return fam, d.k_pad, d.n_pad
