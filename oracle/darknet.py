"""Oracle (test infrastructure): Darknet cfg -> layer list -> numpy forward pass.

Restates, in numpy fp32, what the reference's TF-1.12 graph computes:
  * cfg section renaming            build_model.py:12-32
  * layer semantics                 build_model.py:97-270
  * Darknet .weights layout         build_model.py:273-339
PARITY UNPINNED (see oracle/__init__.py): TF op semantics (SAME padding, nearest
resize, batch-norm inference) are restated from their documented behaviour.
"""
import configparser
import io
import os
from collections import defaultdict

import numpy as np

BN_EPS = np.float32(1e-5)          # build_model.py:166
LEAKY_ALPHA = np.float32(0.1)      # build_model.py:176

# head-conv indices by cfg file stem, build_model.py:68-73
_OUT_LAYERS = {"yolov3": [58, 66, 74], "yolov3-spp": [59, 67, 75], "yolov3-tiny": [9, 12, 12]}


def unique_config_sections(cfg_file):
    """build_model.py:12-32 -- append _<n> to every [section] header so that
    configparser accepts the repeated Darknet section names."""
    seen = defaultdict(int)
    out = io.StringIO()
    with open(cfg_file) as fh:
        for line in fh:
            if line.startswith("["):
                name = line.strip().strip("[]")
                line = line.replace(name, "%s_%d" % (name, seen[name]))
                seen[name] += 1
            out.write(line)
    out.seek(0)
    return out


def parse_layers(config_path, num_classes, num_anchors_per_scale=3, in_channels=3):
    """Walk the cfg exactly as build_model.py:97-255 does and return one dict per
    non-[net] section (so list index == Darknet layer number) plus
    the indices of the tensors feeding each [yolo] section (build_model.py:243-247,
    265-267)."""
    assert config_path.endswith(".cfg"), "%s is not a .cfg file" % config_path   # :51
    parser = configparser.ConfigParser()
    parser.read_file(unique_config_sections(config_path))
    stem = os.path.basename(config_path).split(".")[0]                            # :64-65
    if stem not in _OUT_LAYERS:
        # reference: out_layers undefined -> NameError at first conv (:132)
        raise NameError("out_layers undefined for cfg stem %r (build_model.py:68-73)" % stem)
    out_layers = _OUT_LAYERS[stem]

    layers, channels, out_index = [], [], []
    prev_c = in_channels
    conv_idx = -1
    count = -1
    for section in parser.sections():
        sec = parser[section]
        if section.startswith("convolutional"):
            conv_idx += 1
            filters = int(sec["filters"])
            if conv_idx in out_layers:                                            # :156-159
                filters = num_anchors_per_scale * (num_classes + 5)
            size, stride, pad = int(sec["size"]), int(sec["stride"]), int(sec["pad"])
            act = sec["activation"]
            layers.append(dict(type="conv", conv_idx=conv_idx, cin=prev_c, filters=filters, size=size,
                               stride=stride, pad=pad, bn=("batch_normalize" in sec), activation=act,
                               same=(pad == 1 and stride == 1)))                   # :124
            prev_c = filters
            channels.append(prev_c)
            count += 1
        elif section.startswith("route"):
            ids = [int(i) for i in sec["layers"].split(",")]
            # python list indexing on all_layers as it stands now (:186)
            n = len(layers)
            abs_ids = [i if i >= 0 else n + i for i in ids]
            prev_c = sum(channels[i] for i in abs_ids)
            layers.append(dict(type="route", ids=abs_ids))
            channels.append(prev_c)
            count += 1
        elif section.startswith("maxpool"):
            layers.append(dict(type="maxpool", size=int(sec["size"]), stride=int(sec["stride"])))
            channels.append(prev_c)
            count += 1
        elif section.startswith("shortcut"):
            assert sec["activation"] == "linear", "Only linear activation is supported"   # :221
            idx = int(sec["from"])
            n = len(layers)
            layers.append(dict(type="shortcut", src=idx if idx >= 0 else n + idx))
            channels.append(prev_c)
            count += 1
        elif section.startswith("upsample"):
            assert int(sec["stride"]) == 2, "Only stride=2 is supported"                   # :233
            layers.append(dict(type="upsample"))
            channels.append(prev_c)
            count += 1
        elif section.startswith("yolo"):
            out_index.append(count)                                                        # :244
            layers.append(dict(type="yolo"))
            channels.append(0)
            count += 1
        elif section.startswith("net"):
            pass
        else:
            raise ValueError("Unsupported section header type: %s" % section)              # :255
    if not out_index:
        out_index.append(len(layers) - 1)                                                  # :265-266
    return layers, out_index


# ----------------------------------------------------------------------------- weights
def read_darknet_weights(path, layers):
    """build_model.py:286-335: 5 x int32 header, then per conv (cfg order):
    BN -> beta, gamma, mean, var [O] each; else bias [O]; then kernel (O,I,kh,kw)
    transposed to HWIO.  Returns {layer_index: dict(...)} of fp32 arrays."""
    with open(path, "rb") as fp:
        np.fromfile(fp, dtype=np.int32, count=5)
        flat = np.fromfile(fp, dtype=np.float32)
    ptr = 0
    params = {}
    for li, L in enumerate(layers):
        if L["type"] != "conv":
            continue
        o, i, k = L["filters"], L["cin"], L["size"]
        p = {}
        if L["bn"]:
            for name in ("beta", "gamma", "mean", "var"):                                  # :306
                p[name] = flat[ptr:ptr + o].copy(); ptr += o
        else:
            p["bias"] = flat[ptr:ptr + o].copy(); ptr += o
        n = o * i * k * k
        w = flat[ptr:ptr + n].reshape(o, i, k, k); ptr += n                               # :331
        p["kernel"] = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))                # :333 HWIO
        params[li] = p
    return params, ptr, flat.size


# ----------------------------------------------------------------------------- ops
def _same_pads(in_size, k, s):
    """TF 'SAME': out = ceil(in/s); total = max((out-1)*s + k - in, 0); before = total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def conv2d(x, kernel, stride, same):
    """tf.layers.conv2d NHWC x HWIO (build_model.py:158-162).  VALID or SAME."""
    B, H, W, C = x.shape
    kh, kw, ci, co = kernel.shape
    assert ci == C
    if same:
        Ho, pt, pb = _same_pads(H, kh, stride)
        Wo, pl, pr = _same_pads(W, kw, stride)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    else:
        Ho = (H - kh) // stride + 1
        Wo = (W - kw) // stride + 1
    out = np.zeros((B, Ho, Wo, co), dtype=np.float32)
    for dy in range(kh):
        for dx in range(kw):
            patch = x[:, dy:dy + (Ho - 1) * stride + 1:stride, dx:dx + (Wo - 1) * stride + 1:stride, :]
            out += np.matmul(np.ascontiguousarray(patch).reshape(-1, C), kernel[dy, dx]).reshape(B, Ho, Wo, co)
    return out


def batch_norm_inference(x, p):
    """tf.layers.batch_normalization(training=False, epsilon=1e-5), build_model.py:166."""
    inv = p["gamma"] / np.sqrt(p["var"] + BN_EPS)
    return (x - p["mean"]) * inv + p["beta"]


def leaky_relu(x):
    return np.where(x > 0, x, LEAKY_ALPHA * x).astype(np.float32)


def max_pool_same(x, k, s):
    """tf.nn.max_pool(padding='SAME'), build_model.py:207-210: padded cells never win."""
    B, H, W, C = x.shape
    Ho, pt, pb = _same_pads(H, k, s)
    Wo, pl, pr = _same_pads(W, k, s)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    out = np.full((B, Ho, Wo, C), -np.inf, dtype=np.float32)
    for dy in range(k):
        for dx in range(k):
            out = np.maximum(out, xp[:, dy:dy + (Ho - 1) * s + 1:s, dx:dx + (Wo - 1) * s + 1:s, :])
    return out


def upsample2x_nearest(x):
    """tf.image.resize_nearest_neighbor to [2H,2H], align_corners=False
    (build_model.py:236): out[y,x] = in[y//2, x//2].  Square inputs only."""
    assert x.shape[1] == x.shape[2], "reference uses shape[1] for both dims"
    return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


def forward(layers, out_index, params, x, round_fn=None, collect=False):
    """Run the graph of build_model.py:97-270 on NHWC fp32 input; returns the head
    tensors in cfg order (coarse -> fine).  round_fn, if given, is applied to every
    layer output and to the folded weights' inputs (used to model bf16 storage)."""
    x = np.asarray(x, dtype=np.float32)
    rf = (lambda t: t) if round_fn is None else round_fn
    all_layers = []
    prev = rf(x)
    for li, L in enumerate(layers):
        t = L["type"]
        if t == "conv":
            p = params[li]
            if L["stride"] > 1:                                         # :151-154
                prev = np.pad(prev, ((0, 0), (1, 0), (1, 0), (0, 0)))
            y = conv2d(prev, p["kernel"], L["stride"], L["same"])
            if L["bn"]:
                y = batch_norm_inference(y, p)
            else:
                y = y + p["bias"]
            if L["activation"] == "leaky":
                y = leaky_relu(y)
            prev = rf(y.astype(np.float32))
            all_layers.append(prev)
        elif t == "route":
            srcs = [all_layers[i] for i in L["ids"]]
            prev = srcs[0] if len(srcs) == 1 else np.concatenate(srcs, axis=-1)
            all_layers.append(prev)
        elif t == "maxpool":
            prev = max_pool_same(prev, L["size"], L["stride"])
            all_layers.append(prev)
        elif t == "shortcut":
            prev = rf(all_layers[L["src"]] + prev)
            all_layers.append(prev)
        elif t == "upsample":
            prev = upsample2x_nearest(prev)
            all_layers.append(prev)
        elif t == "yolo":
            all_layers.append(None)
            prev = None
    heads = [all_layers[i] for i in out_index]
    return (heads, all_layers) if collect else heads


def bf16_round(a):
    """fp32 -> nearest bfloat16 (ties to even) -> fp32, on the bit pattern (finite inputs)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return r.view(np.float32)


def forward_bf16_model(layers, out_index, params, x, collect=False, first_conv_fp32=False):
    """The same graph with bf16 STORAGE modelled: what the reference's fp32 graph becomes when weights and
    activations are kept in bfloat16 and every sum runs in fp32 (the build's bf16 mode, BASELINE north_star
    "bf16 MFMA").  Batch norm is folded into the kernel (float64, then fp32, then bf16); the network input and
    every layer output are rounded to bf16 once -- a shortcut's sum (build_model.py:223-225) is rounded once, after
    the fp32 add of the un-rounded convolution output, and the head tensors stay fp32.  first_conv_fp32: the first
    convolution reads the fp32 image with fp32 weights (how the build runs a 3-channel first layer outside
    Darknet-53's fused stem, e.g. YOLOv3-tiny's).  Separates the two error
    sources of a bf16 run: bf16 storage itself (this model vs `forward`) and the kernels' arithmetic (GPU vs
    this model, where only fp32 summation order and the bf16 roundings it flips remain)."""
    x = np.asarray(x, dtype=np.float32)
    x = x if first_conv_fp32 else bf16_round(x)
    all_layers, pre = [], {}
    prev = x
    heads = set(out_index)
    first = next(i for i, L in enumerate(layers) if L["type"] == "conv")
    for li, L in enumerate(layers):
        t = L["type"]
        if t == "conv":
            p = params[li]
            k = p["kernel"].astype(np.float64)
            if L["bn"]:
                s = p["gamma"].astype(np.float64) / np.sqrt(p["var"].astype(np.float64) + 1e-5)
                k, b = k * s, p["beta"].astype(np.float64) - p["mean"].astype(np.float64) * s
            else:
                b = p["bias"].astype(np.float64)
            k, b = k.astype(np.float32), b.astype(np.float32)
            if not (first_conv_fp32 and li == first):
                k = bf16_round(k)
            if L["stride"] > 1:
                prev = np.pad(prev, ((0, 0), (1, 0), (1, 0), (0, 0)))
            y = conv2d(prev, k, L["stride"], L["same"]) + b
            if L["activation"] == "leaky":
                y = leaky_relu(y)
            y = y.astype(np.float32)
            pre[li] = y
            prev = y if li in heads else bf16_round(y)
            all_layers.append(prev)
        elif t == "route":
            srcs = [all_layers[i] for i in L["ids"]]
            prev = srcs[0] if len(srcs) == 1 else np.concatenate(srcs, axis=-1)
            all_layers.append(prev)
        elif t == "maxpool":
            prev = max_pool_same(prev, L["size"], L["stride"])
            all_layers.append(prev)
        elif t == "shortcut":
            a = pre.get(li - 1, prev)                       # the producing convolution's fp32 value
            prev = bf16_round((all_layers[L["src"]] + a).astype(np.float32))
            all_layers.append(prev)
        elif t == "upsample":
            prev = upsample2x_nearest(prev)
            all_layers.append(prev)
        elif t == "yolo":
            all_layers.append(None)
            prev = None
    heads_out = [all_layers[i] for i in out_index]
    return (heads_out, all_layers) if collect else heads_out


def conv_gflops(layers, height, width):
    """2*MAC count of the convs, for the topology check against Darknet's BFLOPs."""
    shapes = []
    h, w = height, width
    total = 0
    for L in layers:
        t = L["type"]
        if t == "conv":
            if L["stride"] > 1:
                h, w = (h + 1 - L["size"]) // L["stride"] + 1, (w + 1 - L["size"]) // L["stride"] + 1
            total += 2 * h * w * L["filters"] * L["size"] ** 2 * L["cin"]
        elif t == "maxpool":
            h, w = -(-h // L["stride"]), -(-w // L["stride"])
        elif t == "upsample":
            h, w = 2 * h, 2 * w
        elif t == "route":
            h, w = shapes[L["ids"][0]]
        elif t == "shortcut":
            pass
        shapes.append((h, w))
    return total / 1e9
