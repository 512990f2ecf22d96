"""Darknet `.weights` files: reader (the layout build_model.load_weights consumes,
reference build_model.py:286-335), writer, and a seeded random initialiser.

File layout: 5 x int32 header (major, minor, revision, seen lo, seen hi), then fp32,
per convolution in cfg order:  BN -> beta[O], gamma[O], mean[O], var[O]; otherwise
bias[O]; then the kernel as (O, I, kh, kw).

The writer and the initialiser exist because no pretrained weights are reachable
offline (the reference fetches them with wget, build_model.py:283-285): benchmarks and
parity tests run on seeded random weights written in this format.
"""
import os

import numpy as np


def conv_param_shapes(layers):
    """[(layer_index, out_c, in_c, ksize, has_bn)] in file order."""
    return [(i, L["filters"], L["cin"], L["size"], L["bn"]) for i, L in enumerate(layers) if L["type"] == "conv"]


def read_weights(path, layers):
    """-> {layer_index: {'kernel': HWIO f32, 'beta','gamma','mean','var' | 'bias'}}.
    Raises FileNotFoundError instead of fetching (no network; cf. build_model.py:281-285)."""
    if not os.path.exists(path):
        raise FileNotFoundError("darknet weights not found: %s (nothing is downloaded)" % path)
    with open(path, "rb") as fp:
        header = np.fromfile(fp, dtype=np.int32, count=5)
        flat = np.fromfile(fp, dtype=np.float32)
    if header.size < 5:
        raise ValueError("%s: truncated header" % path)
    params, ptr = {}, 0

    def take(count):
        nonlocal ptr
        if ptr + count > flat.size:
            raise ValueError("%s: file ends after %d floats, layer needs %d more" % (path, flat.size, ptr + count - flat.size))
        out = flat[ptr:ptr + count]
        ptr += count
        return out

    for li, o, i, k, bn in conv_param_shapes(layers):
        p = {}
        if bn:
            p["beta"], p["gamma"], p["mean"], p["var"] = take(o), take(o), take(o), take(o)   # build_model.py:306
        else:
            p["bias"] = take(o)
        w = take(o * i * k * k).reshape(o, i, k, k)                                          # build_model.py:331
        p["kernel"] = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))                    # build_model.py:333
        params[li] = p
    if ptr != flat.size:
        # the reference does not check (build_model.py:295-336); a cfg / weights mismatch must not pass silently
        import warnings
        warnings.warn("%s: %d of %d floats consumed by this cfg -- weights written for a different network?"
                      % (path, ptr, flat.size), RuntimeWarning)
    return params


def write_weights(path, layers, params, seen=0):
    with open(path, "wb") as fp:
        np.array([0, 2, 0, seen & 0xffffffff, seen >> 32], dtype=np.int32).tofile(fp)
        for li, o, i, k, bn in conv_param_shapes(layers):
            p = params[li]
            if bn:
                for name in ("beta", "gamma", "mean", "var"):
                    np.asarray(p[name], dtype=np.float32).reshape(o).tofile(fp)
            else:
                np.asarray(p["bias"], dtype=np.float32).reshape(o).tofile(fp)
            w = np.asarray(p["kernel"], dtype=np.float32).reshape(k, k, i, o)
            np.ascontiguousarray(np.transpose(w, (3, 2, 0, 1))).tofile(fp)                   # HWIO -> (O, I, kh, kw)


def random_params(layers, seed=0, residual_gain=0.2, weight_gain=1.7, head_gain=0.3):
    """Seeded init that keeps activations O(1) through 75 layers:
    W ~ N(0, weight_gain/(k^2 Cin)), gamma ~ U(.8,1.2), beta ~ N(0,.1), mean ~ N(0,.1),
    var ~ U(.5,1.5), head bias ~ N(0,1).  The gamma of a conv that feeds a [shortcut] is
    scaled by residual_gain (otherwise the 23 residual adds of Darknet-53 double the
    variance each time) and the un-normalised head kernels by head_gain, so that the head
    logits stay a few units wide and the sigmoids do not saturate into ties."""
    rng = np.random.default_rng(seed)
    params = {}
    for li, o, i, k, bn in conv_param_shapes(layers):
        std = np.sqrt(weight_gain / (k * k * i)) * (1.0 if bn else head_gain)
        p = {"kernel": (rng.standard_normal((k, k, i, o)) * std).astype(np.float32)}
        if bn:
            feeds_shortcut = li + 1 < len(layers) and layers[li + 1]["type"] == "shortcut"
            p["gamma"] = (rng.uniform(0.8, 1.2, o) * (residual_gain if feeds_shortcut else 1.0)).astype(np.float32)
            p["beta"] = (rng.standard_normal(o) * 0.1).astype(np.float32)
            p["mean"] = (rng.standard_normal(o) * 0.1).astype(np.float32)
            p["var"] = rng.uniform(0.5, 1.5, o).astype(np.float32)
        else:
            p["bias"] = rng.standard_normal(o).astype(np.float32)
        params[li] = p
    return params


def fold_batch_norm(p, eps=1e-5):
    """conv + BN(inference) == conv with kernel*s and bias beta - mean*s, s = gamma/sqrt(var+eps)
    (build_model.py:164-167).  Folded in float64, returned as fp32 (kernel HWIO, bias [O])."""
    k = p["kernel"].astype(np.float64)
    if "gamma" in p:
        s = p["gamma"].astype(np.float64) / np.sqrt(p["var"].astype(np.float64) + eps)
        return (k * s).astype(np.float32), (p["beta"].astype(np.float64) - p["mean"].astype(np.float64) * s).astype(np.float32)
    return k.astype(np.float32), p["bias"].astype(np.float32)
