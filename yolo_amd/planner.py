"""Darknet cfg -> fused launch plan for the MI355X kernels.

The reference walks the cfg and emits one TF op per section
(build_model.py:97-255).  Here the same walk produces a *plan*: a short list of
kernel launches over NHWC buffers in one arena, with
  * BatchNorm folded into the conv weights, leaky-ReLU in the conv epilogue
    (build_model.py:164-178),
  * shortcut (build_model.py:218-227) fused into the producing conv as a residual
    operand,
  * upsample (build_model.py:231-239) fused into the producing conv as a 2x2
    replicated store,
  * route/concat (build_model.py:184-198) made zero-copy: producers write straight
    into their channel slice of the concat buffer, single-input routes are aliases,
and un-fused fallback kernels whenever a cfg does not fit those patterns.
Pure Python, no torch / GPU: the plan is data, tests can interpret it on the CPU.
"""
import configparser
import io
import os
from collections import defaultdict

# head-conv indices chosen from the cfg file stem, build_model.py:64-73
OUT_LAYERS = {"yolov3": [58, 66, 74], "yolov3-spp": [59, 67, 75], "yolov3-tiny": [9, 12, 12]}


def unique_config_sections(cfg_file):
    """Same contract as the reference's build_model.unique_config_sections
    (build_model.py:12-32): returns a StringIO of the cfg in which every
    `[section]` header carries a `_<count>` suffix, so configparser accepts it."""
    counts = defaultdict(int)
    stream = io.StringIO()
    with open(cfg_file) as fh:
        for line in fh:
            if line.startswith("["):
                name = line.strip().strip("[]")
                line = line.replace(name, name + "_" + str(counts[name]))
                counts[name] += 1
            stream.write(line)
    stream.seek(0)
    return stream


def parse_cfg(config_path, num_classes, num_anchors_per_scale=3, in_channels=3):
    """cfg -> (layers, out_index).  One entry per non-[net] section, so list index ==
    Darknet layer number; out_index = layers feeding each [yolo] (build_model.py:243-247)."""
    config_path = os.path.expanduser(config_path)
    assert config_path.endswith(".cfg"), "{} is not a .cfg file".format(config_path)   # build_model.py:51
    if not os.path.exists(config_path):
        raise FileNotFoundError("cfg file not found: %s" % config_path)
    cfg = configparser.ConfigParser()
    cfg.read_file(unique_config_sections(config_path))
    stem = config_path.split("/")[-1].split(".")[0]
    if stem not in OUT_LAYERS:
        # the reference leaves out_layers undefined here and dies with NameError (build_model.py:68-73,132)
        raise NameError("name 'out_layers' is not defined (cfg stem %r is not yolov3 / yolov3-spp / yolov3-tiny)" % stem)
    heads = OUT_LAYERS[stem]

    layers, chans, out_index = [], [], []
    prev_c, conv_idx = in_channels, -1
    for section in cfg.sections():
        sec = cfg[section]
        n = len(layers)
        if section.startswith("convolutional"):
            conv_idx += 1
            is_head = conv_idx in heads
            filters = num_anchors_per_scale * (num_classes + 5) if is_head else int(sec["filters"])
            act = sec["activation"]
            if act not in ("linear", "leaky"):
                raise ValueError("Unsupported activation: %s" % act)
            layers.append(dict(type="conv", conv_idx=conv_idx, cin=prev_c, filters=filters, size=int(sec["size"]),
                               stride=int(sec["stride"]), pad=int(sec["pad"]), bn="batch_normalize" in sec,
                               activation=act, head=is_head))
            prev_c = filters
        elif section.startswith("route"):
            ids = [int(i) for i in sec["layers"].split(",")]
            ids = [i if i >= 0 else n + i for i in ids]
            layers.append(dict(type="route", ids=ids))
            prev_c = sum(chans[i] for i in ids)
        elif section.startswith("maxpool"):
            layers.append(dict(type="maxpool", size=int(sec["size"]), stride=int(sec["stride"])))
        elif section.startswith("shortcut"):
            assert sec["activation"] == "linear", "Only linear activation is supported"
            src = int(sec["from"])
            layers.append(dict(type="shortcut", src=src if src >= 0 else n + src))
        elif section.startswith("upsample"):
            assert int(sec["stride"]) == 2, "Only stride=2 is supported"
            layers.append(dict(type="upsample"))
        elif section.startswith("yolo"):
            out_index.append(n - 1)
            layers.append(dict(type="yolo"))
        elif section.startswith("net"):
            continue
        else:
            raise ValueError("Unsupported section header type: {}".format(section))
        chans.append(prev_c if layers[-1]["type"] != "yolo" else 0)
    if not out_index:
        out_index.append(len(layers) - 1)
    return layers, out_index


def _ceil_to(x, m):
    return (x + m - 1) // m * m


def conv_out_hw(h, w, size, stride):
    """Padding rule of build_model.py:124,151-154 (see y3_conv_desc)."""
    pad = 1 if size == 3 else 0
    if stride == 1:
        return h, w
    return (h + pad - size) // stride + 1, (w + pad - size) // stride + 1


def infer_shapes(layers, in_h, in_w, in_c=3):
    shapes = []
    cur = (in_h, in_w, in_c)
    for L in layers:
        t = L["type"]
        if t == "conv":
            if L["size"] not in (1, 3) or L["stride"] not in (1, 2):
                raise ValueError("conv size %d stride %d unsupported" % (L["size"], L["stride"]))
            if L["size"] == 3 and L["pad"] != 1:
                raise ValueError("3x3 conv with pad=0 unsupported")
            if L["size"] == 1 and L["stride"] > 1:
                # the reference pads one row/column top/left for ANY stride > 1 (build_model.py:151-154), so a 1x1
                # stride-2 conv samples rows -1, 1, 3, ... and yields h//2 + 1 outputs; no shipped cfg has one and
                # the kernels do not implement that sampling: refuse rather than compute something else
                raise ValueError("1x1 conv with stride %d unsupported (build_model.py:151-154 pads top/left)" % L["stride"])
            h, w = conv_out_hw(cur[0], cur[1], L["size"], L["stride"])
            cur = (h, w, L["filters"])
        elif t == "maxpool":
            s = L["stride"]
            cur = (-(-cur[0] // s), -(-cur[1] // s), cur[2])
        elif t == "upsample":
            assert cur[0] == cur[1], "reference upsample is square-only (build_model.py:236)"
            cur = (2 * cur[0], 2 * cur[1], cur[2])
        elif t == "route":
            srcs = [shapes[i] for i in L["ids"]]
            assert all(s[:2] == srcs[0][:2] for s in srcs), "route: spatial sizes differ"
            cur = (srcs[0][0], srcs[0][1], sum(s[2] for s in srcs))
        elif t == "shortcut":
            assert shapes[L["src"]] == cur, "shortcut: shapes differ"
        elif t == "yolo":
            shapes.append(None)
            cur = None
            continue
        shapes.append(cur)
    return shapes


class Plan(object):
    """ops: list of dicts; every tensor reference is (buf, elem_offset, ld) where buf is
    'in' (network input), ('arena', byte_offset) or ('head', k)."""

    def __init__(self):
        self.ops = []
        self.arena_bytes = 0
        self.heads = []          # per head: dict(h, w, c, ld, layer)
        self.shapes = None
        self.batch = 0
        self.dtype = "bf16"
        self.conv_layers = []    # layer indices of convs, cfg order


def find_stem(layers, inputs, uses, head_of, in_c=3):
    """Darknet-53's first layers as one launch (csrc/conv_stem.hip): conv 3x3 s1 3->32, conv 3x3 s2
    32->64 and, if it follows, conv 1x1 64->32.  Returns (i0, i1, i2 or None) or None."""
    convs = [i for i, L in enumerate(layers) if L["type"] == "conv"]
    if len(convs) < 2:
        return None
    i0, i1 = convs[0], convs[1]
    A, B = layers[i0], layers[i1]
    if not (inputs[i0] == [-1] and in_c == 3 and (A["size"], A["stride"], A["filters"]) == (3, 1, 32)):
        return None
    if not (inputs[i1] == [i0] and uses[i0] == 1 and (B["size"], B["stride"], B["filters"]) == (3, 2, 64)):
        return None
    if i0 in head_of or i1 in head_of:
        return None
    i2 = None
    if len(convs) > 2:
        c = convs[2]
        C = layers[c]
        if inputs[c] == [i1] and (C["size"], C["stride"], C["filters"]) == (1, 1, 32) and c not in head_of:
            i2 = c
    return (i0, i1, i2)


RESBLOCK_TILE_W, RESBLOCK_TILE_H = 19, 13      # csrc/conv_resblock.hip: RB_TW, RB_TH
RESBLOCK_MIN_TILES = 1024                      # fuse only where the launch fills the chip several times over
# Block widths the automatic mode fuses.  Measured at batch 64 / 608: C = 128 @152 0.42 ms fused against
# 0.47 ms in two launches; C = 256 @76 0.317 ms against 0.29 ms (the tile boundary of the fused kernel is bound by the
# CU's ~15 B/clk memory ingest, which the 1x1 launch, HBM-bound chip-wide, does not pay), so 256 stays un-fused.
# C = 64 @304 (csrc/conv_resblock64.hip): stem + block 0.764 ms against stem-with-1x1 + 3x3 0.814 ms.
RESBLOCK_AUTO_CHANNELS = (128, 64)


def resblock_tiles(batch, h, w, c=256):
    """Workgroup tiles of y3_resblock_fused on a [batch, h, w, c] tensor, or 0 when the shape is not supported:
    19 x 13-pixel tiles for c = 256 (csrc/conv_resblock.hip), 16 x 8 for c = 128 (csrc/conv_resblock128.hip, any width),
    16 x 16 for c = 64 (csrc/conv_resblock64.hip, any width)."""
    if c == 128:
        return batch * (-(-h // 8)) * (-(-w // 16))
    if c == 64:
        return batch * (-(-h // 16)) * (-(-w // 16))
    if w % RESBLOCK_TILE_W:
        return 0
    return batch * (-(-h // RESBLOCK_TILE_H)) * (w // RESBLOCK_TILE_W)


def find_resblocks(layers, inputs, uses, head_of, shapes, batch, dtype, mode, channels=None):
    """Darknet residual blocks that run as ONE launch (csrc/conv_resblock.hip): [convolutional 1x1 C -> C/2],
    [convolutional 3x3 C/2 -> C], [shortcut from=-3] with C = 256 or 128 on a feature map whose width is a multiple of
    19, bf16.  mode: None = automatic (bf16 plans, C in RESBLOCK_AUTO_CHANNELS, launches of at least
    RESBLOCK_MIN_TILES tiles: at small batch the un-fused 128-pixel tiles fill the chip better, and both forms give the
    same bits), True = wherever the pattern fits (tests), False = never.  channels overrides the widths considered.
    Returns {index of the 3x3: (i1, i3, i_shortcut)}."""
    found = {}
    if channels is None:
        channels = RESBLOCK_AUTO_CHANNELS if mode is None else (256, 128, 64)
    if mode is False or (mode is None and dtype != "bf16"):
        return found
    for i3, L3 in enumerate(layers):
        if i3 < 1 or i3 + 1 >= len(layers) or L3["type"] != "conv":
            continue
        i1, isc = inputs[i3][0], i3 + 1
        if i1 < 0 or layers[i1]["type"] != "conv" or layers[isc]["type"] != "shortcut":
            continue
        L1, S = layers[i1], layers[isc]
        src = inputs[i1][0]
        if src < 0 or inputs[isc] != [src, i3] or S["src"] != src:
            continue
        c = shapes[src][2]
        if c not in channels or (L1["size"], L1["stride"], L1["filters"]) != (1, 1, c // 2):
            continue
        if (L3["size"], L3["stride"], L3["filters"], L3["pad"]) != (3, 1, c, 1):
            continue
        if uses[i1] != 1 or uses[i3] != 1 or i1 in head_of or i3 in head_of:
            continue
        tiles = resblock_tiles(batch, shapes[src][0], shapes[src][1], c)
        if tiles == 0 or (mode is None and tiles < RESBLOCK_MIN_TILES):
            continue
        found[i3] = (i1, i3, isc)
    return found


def build_plan(layers, out_index, batch, in_h, in_w, dtype="bf16", in_c=3, fuse=True, pad_input=False, stem=True,
               snake=True, resblock=None, inplace_shortcut=None):
    assert dtype in ("bf16", "fp32")
    if inplace_shortcut is None:
        inplace_shortcut = os.environ.get("YOLO_AMD_INPLACE", "1") != "0"
    es = 2 if dtype == "bf16" else 4
    epc = 16 // es
    shapes = infer_shapes(layers, in_h, in_w, in_c)
    n = len(layers)

    # ---- data-flow ------------------------------------------------------------------------
    inputs = []
    prev = -1
    for i, L in enumerate(layers):
        t = L["type"]
        if t in ("conv", "maxpool", "upsample"):
            inputs.append([prev])
        elif t == "shortcut":
            inputs.append([L["src"], prev])
        elif t == "route":
            inputs.append(list(L["ids"]))
        else:
            inputs.append([])
        prev = None if t == "yolo" else i
    for i, ins in enumerate(inputs):
        if any(j is None for j in ins):
            raise ValueError("layer %d consumes the output of a [yolo] section" % i)
    uses = [0] * n
    for ins in inputs:
        for j in ins:
            if j >= 0:
                uses[j] += 1
    head_of = {li: k for k, li in enumerate(out_index)}
    for li in out_index:
        if layers[li]["type"] != "conv":
            raise ValueError("output layer %d is not a convolution" % li)

    # ---- fusion decisions -------------------------------------------------------------------
    fused_into = {}      # consumer layer (shortcut/upsample) -> producing conv
    absorbed = {}        # conv -> consumer layer whose tensor it writes
    stem_layers = find_stem(layers, inputs, uses, head_of, in_c) if (fuse and stem and dtype == "bf16" and not pad_input) else None
    if fuse:
        for i, L in enumerate(layers):
            if L["type"] not in ("shortcut", "upsample") or i == 0:
                continue
            p = inputs[i][-1]
            if p < 0 or layers[p]["type"] != "conv" or uses[p] != 1 or p in head_of or p in absorbed:
                continue
            if stem_layers is not None and p in stem_layers:
                continue
            if layers[p]["cin"] % epc != 0:        # direct-conv family has no fused epilogue
                continue
            if L["type"] == "shortcut" and shapes[L["src"]] != shapes[p]:
                continue
            fused_into[i] = p
            absorbed[p] = i

        # the first convolution (3 input channels: the direct kernel) absorbs a 2x2 stride-2 max pool that follows it
        # (YOLOv3-tiny layers 0-1): the full-resolution activation, the largest tensor of the network, is never stored
        for i, L in enumerate(layers):
            if L["type"] != "maxpool" or (L["size"], L["stride"]) != (2, 2) or pad_input:
                continue
            p = inputs[i][0]
            if p < 0 or layers[p]["type"] != "conv" or uses[p] != 1 or p in head_of or p in absorbed:
                continue
            if stem_layers is not None and p in stem_layers:
                continue
            P = layers[p]
            if inputs[p] != [-1] or in_c != 3 or (P["size"], P["stride"]) != (3, 1) or shapes[p][0] % 2 or shapes[p][1] % 2:
                continue
            fused_into[i] = p
            absorbed[p] = i

    rb_env = os.environ.get("YOLO_AMD_RESBLOCK_C")                                        # A/B runs: "256,128", "128"
    rb_channels = tuple(int(v) for v in rb_env.split(",") if v) if rb_env else None
    resblocks = find_resblocks(layers, inputs, uses, head_of, shapes, batch, dtype, resblock, rb_channels) if fuse else {}
    resblocks = {i3: v for i3, v in resblocks.items() if fused_into.get(v[2]) == i3}     # the shortcut rides the 3x3
    if resblock is None:
        # resblock64_kernel sums its 1x1 in conv_stem's k-slot order (csrc/conv_resblock64.hip), which differs from
        # y3_conv2d_fused's by one bf16 ulp in ~1 of 1e4 outputs.  The automatic mode promises the same bits at every
        # batch size, so it fuses a 64-channel block only where the small-batch plan computes that 1x1 in the stem's
        # launch (stem on, no padded input, the block directly behind the stem).  resblock=True fuses regardless.
        resblocks = {i3: v for i3, v in resblocks.items()
                     if shapes[inputs[v[0]][0]][2] != 64 or (stem_layers is not None and stem_layers[2] == v[0])}
    rb_internal = {v[0] for v in resblocks.values()}      # 1x1 outputs that never leave LDS: no buffer, no launch
    if stem_layers is not None and stem_layers[2] in rb_internal:
        stem_layers = (stem_layers[0], stem_layers[1], None)                             # the block's launch computes the 1x1

    # ---- storage: aliases, concat slices, own buffers ---------------------------------------
    bufs = []            # dict(elems, es, kind)
    tens = [None] * n    # dict(buf, off, ld, c)  (off/ld in elements)

    def new_buf(elems, esize=es, kind="arena"):
        bufs.append(dict(elems=elems, es=esize, kind=kind))
        return len(bufs) - 1

    def pixels(i):
        return batch * shapes[i][0] * shapes[i][1]

    claimed = {}
    concat_buf = {}
    for i, L in enumerate(layers):
        if L["type"] != "route" or len(L["ids"]) < 2:
            continue
        c_total = shapes[i][2]
        ld = _ceil_to(c_total, epc)
        b = new_buf(pixels(i) * ld)
        concat_buf[i] = (b, ld)
        off = 0
        for j in L["ids"]:
            cj = shapes[j][2]
            ok = (j not in claimed and j not in head_of and j not in absorbed
                  and layers[j]["type"] in ("conv", "maxpool", "upsample", "shortcut")
                  and off % epc == 0 and cj % epc == 0 and L["ids"].count(j) == 1)
            if ok:
                claimed[j] = (b, off, ld)
            off += cj

    # In-place shortcuts.  A convolution with a fused shortcut reads its residual operand R and writes the sum element for
    # element (same lane, load before store, in every epilogue), so when the sum is R's last use the two share storage:
    # the stores hit lines the loads just brought in, and the working set of a residual stage shrinks by one tensor
    # (at 76x76, batch 64: 473 -> 284 MB against the 256 MiB Infinity Cache).  Not for the fused block kernel, whose
    # tiles read their neighbours' pixels of x (halo).
    inplace = {}
    if fuse and inplace_shortcut:
        # readers per STORAGE: a single-input route is an alias of its source (tens[j] = dict(tens[src]) below), so
        # whoever reads the route reads the source's bytes
        root = list(range(n))
        for j, Lj in enumerate(layers):
            if Lj["type"] == "route" and len(Lj["ids"]) == 1 and Lj["ids"][0] >= 0:
                root[j] = root[Lj["ids"][0]]
        users = [[] for _ in range(n)]
        for j, ins in enumerate(inputs):
            if root[j] != j:
                continue                       # the alias itself reads nothing
            for q in ins:
                if q is not None and q >= 0:
                    users[root[q]].append(j)
        for i, p in fused_into.items():
            L = layers[i]
            if L["type"] != "shortcut" or p in resblocks:
                continue
            src = L["src"]
            if src < 0 or src in claimed or src in head_of or i in claimed or i in head_of or shapes[src] != shapes[i]:
                continue
            if layers[src]["type"] == "route":
                continue
            if src in absorbed or src in rb_internal or root[src] != src or any(q >= 0 and root[q] == src for q in inputs[p]):
                continue                       # (the 3x3 itself reads R, directly or through a route alias: halo reads would race)
            if any(j != i and j >= p for j in users[src]):
                continue                       # someone reads R after the sum is written (a route to a head branch)
            inplace[i] = src

    for i, L in enumerate(layers):
        t = L["type"]
        if t == "yolo" or i in absorbed or i in rb_internal or (stem_layers is not None and i == stem_layers[0]):
            continue
        c = shapes[i][2]
        if t == "route":
            if len(L["ids"]) == 1:
                src = tens[L["ids"][0]]
                tens[i] = dict(src)
            else:
                b, ld = concat_buf[i]
                tens[i] = dict(buf=b, off=0, ld=ld, c=c)
            continue
        if i in head_of:
            ld = _ceil_to(c, 8)
            b = new_buf(pixels(i) * ld, 4, kind=("head", head_of[i]))
            tens[i] = dict(buf=b, off=0, ld=ld, c=c)
        elif i in claimed:
            b, off, ld = claimed[i]
            tens[i] = dict(buf=b, off=off, ld=ld, c=c)
        elif i in inplace:
            tens[i] = dict(tens[inplace[i]])     # the shortcut's sum overwrites its (dead) shortcut operand
        else:
            ld = _ceil_to(c, epc)
            tens[i] = dict(buf=new_buf(pixels(i) * ld), off=0, ld=ld, c=c)

    in_tensor = dict(buf="in", off=0, ld=in_c, c=in_c)
    # network input with fewer channels than one 16-byte chunk: stage it once, zero padded to a
    # chunk, so the first conv takes the MFMA path instead of the direct kernel
    padded_in = None
    if pad_input and in_c % epc != 0 and in_c < epc:
        padded_in = dict(buf=new_buf(batch * in_h * in_w * epc), off=0, ld=epc, c=epc)

    def tin(j):
        return in_tensor if j < 0 else tens[j]

    # ---- emit ops ------------------------------------------------------------------------
    plan = Plan()
    plan.batch, plan.dtype, plan.shapes = batch, dtype, shapes
    ops = plan.ops
    pad_emitted = False
    for i, L in enumerate(layers):
        t = L["type"]
        if t == "conv" and stem_layers is not None and i in stem_layers:
            plan.conv_layers.append(i)
            if i == stem_layers[0]:
                i0, i1, i2 = stem_layers
                ops.append(dict(kind="stem", layer=i0, layers=stem_layers, x=in_tensor, in_hw=(in_h, in_w),
                                out_hw=shapes[i1][:2], y=tens[i1], y2=tens[i2] if i2 is not None else None,
                                leaky=tuple(layers[j]["activation"] == "leaky" for j in stem_layers if j is not None),
                                convs=[dict(out_hw=shapes[j][:2], out_c=shapes[j][2], ksize=layers[j]["size"], in_c=layers[j]["cin"])
                                       for j in stem_layers if j is not None]))
        elif t == "conv" and i in rb_internal:
            plan.conv_layers.append(i)                 # computed inside the block's launch (emitted at its 3x3)
        elif t == "conv" and i in resblocks:
            plan.conv_layers.append(i)
            i1, i3, isc = resblocks[i]
            src = inputs[i1][0]
            h, w, c = shapes[src]
            ops.append(dict(kind="resblock", layer=i1, layers=(i1, i3, isc), x=tens[src], y=tens[isc], hw=(h, w), c=c,
                            leaky=(layers[i1]["activation"] == "leaky", layers[i3]["activation"] == "leaky"),
                            writes_layer=isc, tiles=resblock_tiles(batch, h, w, c)))
        elif t == "conv":
            plan.conv_layers.append(i)
            src = inputs[i][0]
            ih, iw, ic = (in_h, in_w, in_c) if src < 0 else shapes[src]
            oh, ow, oc = shapes[i]
            xin, in_f32 = tin(src), src < 0
            if src < 0 and padded_in is not None:
                if not pad_emitted:
                    ops.append(dict(kind="padin", layer=i, x=in_tensor, y=padded_in, pixels=batch * in_h * in_w, c=in_c))
                    pad_emitted = True
                xin, in_f32, ic = padded_in, False, epc
            op = dict(kind="conv", layer=i, x=xin, in_hw=(ih, iw), in_c=ic, out_hw=(oh, ow), out_c=oc,
                      ksize=L["size"], stride=L["stride"], leaky=L["activation"] == "leaky",
                      in_f32=in_f32, out_f32=(i in head_of), res=None, ups=False, pool=False)
            tgt = i
            if i in absorbed:
                tgt = absorbed[i]
                if layers[tgt]["type"] == "shortcut":
                    op["res"] = tens[layers[tgt]["src"]]
                elif layers[tgt]["type"] == "maxpool":
                    op["pool"] = True
                else:
                    op["ups"] = True
            op["y"] = tens[tgt]
            op["writes_layer"] = tgt
            ops.append(op)
        elif t == "maxpool" and i not in fused_into:
            src = inputs[i][0]
            ops.append(dict(kind="maxpool", layer=i, x=tin(src), y=tens[i], in_hw=shapes[src][:2], c=shapes[i][2],
                            size=L["size"], stride=L["stride"]))
        elif t == "upsample" and i not in fused_into:
            src = inputs[i][0]
            ops.append(dict(kind="upsample", layer=i, x=tin(src), y=tens[i], in_hw=shapes[src][:2], c=shapes[i][2]))
        elif t == "shortcut" and i not in fused_into:
            a, b2 = inputs[i]
            ops.append(dict(kind="add", layer=i, a=tin(a), b=tin(b2), y=tens[i], pixels=pixels(i), c=shapes[i][2]))
        elif t == "route" and len(L["ids"]) > 1:
            b, ld = concat_buf[i]
            off = 0
            for j in L["ids"]:
                cj = shapes[j][2]
                if not (tens[j]["buf"] == b and tens[j]["off"] == off):
                    if off % epc or cj % epc:
                        raise ValueError("route %d: channel slice %d+%d is not 16-byte aligned" % (i, off, cj))
                    ops.append(dict(kind="copy", layer=i, x=tens[j], y=dict(buf=b, off=off, ld=ld, c=cj),
                                    pixels=pixels(i), c=cj))
                off += cj

    # ---- liveness + arena layout ------------------------------------------------------------
    def refs(op, keys):
        return [op[k]["buf"] for k in keys if op.get(k) is not None and op[k]["buf"] != "in"]

    birth, death = {}, {}
    for k, op in enumerate(ops):
        for b in refs(op, ("y", "y2")):
            birth.setdefault(b, k)
            death[b] = max(death.get(b, k), k)
        for b in refs(op, ("x", "a", "b", "res")):
            if b not in birth:
                raise AssertionError("op %d reads buffer %d before it is written" % (k, b))
            death[b] = max(death.get(b, k), k)
    placed = []
    offsets = {}
    for b in sorted(birth, key=lambda q: (birth[q], -bufs[q]["elems"])):
        if bufs[b]["kind"] != "arena":
            continue
        size = _ceil_to(bufs[b]["elems"] * bufs[b]["es"], 256)
        taken = sorted((o, o + s) for (o, s, b0, d0) in placed if not (d0 < birth[b] or death[b] < b0))
        pos = 0
        for lo, hi in taken:
            if pos + size <= lo:
                break
            pos = max(pos, hi)
        placed.append((pos, size, birth[b], death[b]))
        offsets[b] = pos
        plan.arena_bytes = max(plan.arena_bytes, pos + size)

    heads = [None] * len(out_index)
    for b, info in enumerate(bufs):
        if info["kind"] != "arena":
            k = info["kind"][1]
            li = out_index[k]
            heads[k] = dict(h=shapes[li][0], w=shapes[li][1], c=shapes[li][2], ld=tens[li]["ld"], layer=li)
    plan.heads = heads

    def resolve(tn):
        if tn is None or tn["buf"] == "in":
            return tn
        info = bufs[tn["buf"]]
        where = ("arena", offsets[tn["buf"]]) if info["kind"] == "arena" else info["kind"]
        return dict(buf=where, off=tn["off"], ld=tn["ld"], c=tn["c"])

    for op in ops:
        for k in ("x", "y", "y2", "a", "b", "res"):
            if k in op:
                op[k] = resolve(op[k])
    # Snake order: consecutive convolutions walk their output tiles in opposite directions, so each one starts
    # on the part of its input that the previous one wrote last (still in L2 / Infinity Cache) -- same results.
    k = 0
    for op in ops:
        if op["kind"] in ("conv", "stem", "resblock"):
            op["reverse"] = bool(snake and (k & 1) and op["kind"] != "stem")
            k += 1
    return plan


def op_flops(op, batch):
    """Algorithmic FLOPs (2*MACs) of one conv / stem launch."""
    if op["kind"] == "conv":
        oh, ow = op["out_hw"]
        return 2.0 * batch * oh * ow * op["out_c"] * op["ksize"] ** 2 * op["in_c"]
    if op["kind"] == "stem":
        return sum(2.0 * batch * c["out_hw"][0] * c["out_hw"][1] * c["out_c"] * c["ksize"] ** 2 * c["in_c"] for c in op["convs"])
    if op["kind"] == "resblock":                 # the two convolutions' algorithmic FLOPs (halo recomputation not counted)
        h, w = op["hw"]
        c = op["c"]
        return 2.0 * batch * h * w * (c * (c // 2) + 9 * (c // 2) * c)
    return 0.0


def op_bytes(op, batch, es):
    """Algorithmic HBM bytes of one conv / stem launch: input + output(s) (+ residual) + weights."""
    if op["kind"] == "conv":
        oh, ow = op["out_hw"]; ih, iw = op["in_hw"]
        byt = batch * ih * iw * op["in_c"] * (4 if op["in_f32"] else es)
        byt += batch * oh * ow * op["out_c"] * (4 if op["out_f32"] else es) * (4 if op["ups"] else 1) // (4 if op.get("pool") else 1)
        byt += op["out_c"] * op["ksize"] ** 2 * op["in_c"] * es
        if op["res"] is not None:
            byt += batch * oh * ow * op["out_c"] * es
        return float(byt)
    if op["kind"] == "resblock":                 # x once, out once, both weight matrices
        h, w = op["hw"]
        c = op["c"]
        return float(2 * batch * h * w * c * es + (c * (c // 2) + 9 * (c // 2) * c) * es)
    if op["kind"] == "stem":
        ih, iw = op["in_hw"]; oh, ow = op["out_hw"]
        byt = batch * ih * iw * 3 * 4
        for c in op["convs"][1:]:
            byt += batch * oh * ow * c["out_c"] * es
        return float(byt)
    return 0.0


def conv_gflops(layers, in_h, in_w):
    """2*MACs of all convolutions for one image (matches Darknet's printed BFLOPs)."""
    shapes = infer_shapes(layers, in_h, in_w)
    total = 0
    for i, L in enumerate(layers):
        if L["type"] == "conv":
            h, w, c = shapes[i]
            total += 2 * h * w * c * L["size"] ** 2 * L["cin"]
    return total / 1e9
