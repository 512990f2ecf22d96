"""Helpers for the -m gpu parity tests: call the C-ABI on torch cuda tensors."""
import ctypes

import numpy as np
import torch

from yolo_amd import _lib
from yolo_amd.build_model import pack_conv


def bf16_round(x):
    """fp32 numpy -> nearest bf16, back to fp32 (numpy)."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_dev(x, dtype):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    return t.to(torch.bfloat16) if dtype == "bf16" else t


def run_conv(x, kernel, bias, stride, dtype, leaky=True, residual=None, ups=False, out_f32=False, in_f32=False,
             in_pad=(0, 0), out_pad=(0, 0), splitk=False, info=None, pool=False, reverse=False, ws=False, extra_flags=0):
    """x NHWC fp32 numpy; kernel HWIO; returns the NHWC fp32 numpy output.  in_pad/out_pad =
    (channels before, channels after) embed the tensor in a wider buffer (route slices)."""
    lib = _lib.load()
    B, H, W, C = x.shape
    ks, _, _, O = kernel.shape
    pad = 1 if ks == 3 else 0
    Ho = H if stride == 1 else (H + pad - ks) // stride + 1
    Wo = W if stride == 1 else (W + pad - ks) // stride + 1
    wt, bt, fam, k_pad, n_pad = pack_conv(kernel, bias, dtype, in_f32, "cuda")
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    in_ld = in_pad[0] + C + in_pad[1]
    xin = torch.full((B, H, W, in_ld), 7.0, dtype=torch.float32 if in_f32 else tdt, device="cuda")
    xin[..., in_pad[0]:in_pad[0] + C] = torch.from_numpy(x).cuda().to(xin.dtype)
    o8 = (O + 7) // 8 * 8
    out_ld = out_pad[0] + o8 + out_pad[1]
    odt = torch.float32 if (out_f32 or dtype == "fp32") else torch.bfloat16
    Hs, Ws = (2 * Ho, 2 * Wo) if ups else (Ho, Wo)
    if pool:            # Y3_CONV_POOL2: the launch writes the 2x2 stride-2 max pool of its output
        Hs, Ws = Ho // 2, Wo // 2
    out = torch.full((B, Hs, Ws, out_ld), -9.0, dtype=odt, device="cuda")
    res_t, res_ld = None, 0
    if residual is not None:
        res_ld = o8 + 8
        res_t = torch.zeros((B, Ho, Wo, res_ld), dtype=tdt, device="cuda")
        res_t[..., :O] = torch.from_numpy(residual).cuda().to(tdt)
    zeros = torch.zeros(256, dtype=torch.uint8, device="cuda")
    d = _lib.ConvDesc()
    d.dtype = _lib.Y3_BF16 if dtype == "bf16" else _lib.Y3_F32
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, C, in_ld
    d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld = Ho, Wo, O, out_ld, res_ld
    d.ksize, d.stride, d.k_pad, d.n_pad = ks, stride, k_pad, n_pad
    d.flags = ((_lib.CONV_LEAKY if leaky else 0) | (_lib.CONV_RESIDUAL if residual is not None else 0)
               | (_lib.CONV_UPS2X if ups else 0) | (_lib.CONV_OUT_F32 if out_f32 else 0) | (_lib.CONV_IN_F32 if in_f32 else 0)
               | (_lib.CONV_POOL2 if pool else 0) | (_lib.CONV_REVERSE if reverse else 0) | extra_flags)
    es_in = xin.element_size()
    es_out = out.element_size()
    if splitk:          # y3_conv2d_splitk with the workspace its plan asks for (splits == 1: same as y3_conv2d_fused)
        splits, ws_bytes = ctypes.c_int32(1), ctypes.c_int64(0)
        _lib.check(lib.y3_conv_splitk_plan(ctypes.byref(d), ctypes.byref(splits), ctypes.byref(ws_bytes)))
        ws = torch.full((max(int(ws_bytes.value), 16),), 0x7f, dtype=torch.uint8, device="cuda")
        if info is not None:
            info["splits"] = int(splits.value)
        rc = lib.y3_conv2d_splitk(ctypes.byref(d), xin.data_ptr() + in_pad[0] * es_in, wt.data_ptr(), bt.data_ptr(),
                                  res_t.data_ptr() if res_t is not None else None,
                                  out.data_ptr() + out_pad[0] * es_out, zeros.data_ptr(), ws.data_ptr(), ws.numel(), stream())
    elif ws:            # y3_conv2d_fused_ws with the workspace it asks for: poisoned slabs, zeroed flags
        nbytes = ctypes.c_int64(0)
        _lib.check(lib.y3_conv_workspace_bytes(ctypes.byref(d), ctypes.byref(nbytes)))
        if info is not None:
            info["slab_bytes"] = int(nbytes.value)
        slabs = torch.full((max(int(nbytes.value), 16),), 0xff, dtype=torch.uint8, device="cuda")
        flags = torch.zeros(256, dtype=torch.int32, device="cuda")
        rc = lib.y3_conv2d_fused_ws(ctypes.byref(d), xin.data_ptr() + in_pad[0] * es_in, wt.data_ptr(), bt.data_ptr(),
                                    res_t.data_ptr() if res_t is not None else None,
                                    out.data_ptr() + out_pad[0] * es_out, zeros.data_ptr(), slabs.data_ptr(), slabs.numel(),
                                    flags.data_ptr(), stream())
        if info is not None:
            _lib.check(rc)
            torch.cuda.synchronize()
            info["flags_raised"] = int((flags != 0).sum().item())
    else:
        rc = lib.y3_conv2d_fused(ctypes.byref(d), xin.data_ptr() + in_pad[0] * es_in, wt.data_ptr(), bt.data_ptr(),
                                 res_t.data_ptr() if res_t is not None else None,
                                 out.data_ptr() + out_pad[0] * es_out, zeros.data_ptr(), stream())
    _lib.check(rc)
    torch.cuda.synchronize()
    full = out.float().cpu().numpy()
    # channels outside the slice must be untouched
    if out_pad[0]:
        assert np.all(full[..., :out_pad[0]] == -9.0)
    if out_pad[1]:
        assert np.all(full[..., out_pad[0] + o8:] == -9.0)
    return full[..., out_pad[0]:out_pad[0] + O]


def run_stem(x, kernels, biases, leaky=(True, True, True), out_pad=(0, 0)):
    """Fused stem (y3_conv_stem) on an fp32 NHWC image; kernels/biases for conv A, B and optionally C
    (HWIO).  Returns (out1, out2 or None) as fp32 numpy."""
    lib = _lib.load()
    B, H, W, _ = x.shape
    Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
    w0, b0, fam0, _, n0_pad = pack_conv(kernels[0], biases[0], "bf16", True, "cuda")
    w1, b1, fam1, k1_pad, _ = pack_conv(kernels[1], biases[1], "bf16", False, "cuda")
    assert fam0 == 1 and fam1 == 0
    has_c = len(kernels) > 2
    d = _lib.StemDesc()
    d.batch, d.in_h, d.in_w = B, H, W
    d.c0, d.c1, d.c2 = 32, 64, 32 if has_c else 0
    d.n0_pad, d.k1_pad = n0_pad, k1_pad
    ld1 = out_pad[0] + 64 + out_pad[1]
    out1 = torch.full((B, Ho, Wo, ld1), -9.0, dtype=torch.bfloat16, device="cuda")
    d.out1_ld = ld1
    w2 = b2 = out2 = None
    if has_c:
        w2, b2, fam2, k2_pad, _ = pack_conv(kernels[2], biases[2], "bf16", False, "cuda")
        assert fam2 == 0
        d.k2_pad, d.out2_ld = k2_pad, 40
        out2 = torch.full((B, Ho, Wo, 40), -9.0, dtype=torch.bfloat16, device="cuda")
    d.flags = sum(1 << j for j, lk in enumerate(leaky) if lk)
    xin = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    rc = lib.y3_conv_stem(ctypes.byref(d), xin.data_ptr(), w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                          w2.data_ptr() if has_c else None, b2.data_ptr() if has_c else None,
                          out1.data_ptr() + out_pad[0] * 2, out2.data_ptr() if has_c else None, stream())
    _lib.check(rc)
    torch.cuda.synchronize()
    f1 = out1.float().cpu().numpy()
    if out_pad[0]:
        assert np.all(f1[..., :out_pad[0]] == -9.0)
    if out_pad[1]:
        assert np.all(f1[..., out_pad[0] + 64:] == -9.0)
    f2 = None
    if has_c:
        f2 = out2.float().cpu().numpy()
        assert np.all(f2[..., 32:] == -9.0)
        f2 = f2[..., :32]
    return f1[..., out_pad[0]:out_pad[0] + 64], f2


def records_from_dense(boxes, scores, thr):
    """Build the decode kernel's outputs (records, cand_idx, cand_count, class_count) from a
    dense reference-style (boxes [N,4], box_scores [N,C]) pair for ONE image."""
    N, C = scores.shape
    rec = np.zeros((N, 8), dtype=np.float32)
    rec[:, :4] = boxes
    best = scores.max(axis=1)
    rec[:, 4] = best
    mask = np.zeros((N, 3), dtype=np.uint32)
    for c in range(C):
        hit = scores[:, c] >= best                      # all tied argmax classes (utils/utils.py:151)
        mask[:, c >> 5] |= (hit.astype(np.uint32) << np.uint32(c & 31))
    rec[:, 5:8] = mask.view(np.float32)
    cand = np.nonzero(best >= np.float32(thr))[0].astype(np.int32)
    class_count = np.zeros(C, dtype=np.int32)
    for c in range(C):
        class_count[c] = int((((mask[cand, c >> 5] >> np.uint32(c & 31)) & 1) != 0).sum())
    return rec, cand, class_count


def run_nms(per_image, num_classes, max_boxes, iou_thr, with_rank=True):
    """Runs BOTH NMS entry points (y3_nms: workgroup kernel scanning the candidate list; y3_nms_bucketed:
    class buckets + wavefront-per-pair kernel) on the same input, asserts that they agree bit for bit, and
    returns the result.  per_image: list of (rec [N,8], cand [n], class_count [C]) with a common N."""
    plain = _run_nms(per_image, num_classes, max_boxes, iou_thr, with_rank, bucketed=False)
    fast = _run_nms(per_image, num_classes, max_boxes, iou_thr, with_rank, bucketed=True)
    np.testing.assert_array_equal(plain["kept_count"], fast["kept_count"])
    for b in range(plain["kept_count"].shape[0]):
        for c in range(num_classes):
            k = plain["kept_count"][b, c]
            np.testing.assert_array_equal(plain["kept_idx"][b, c, :k], fast["kept_idx"][b, c, :k])
            if with_rank:
                np.testing.assert_array_equal(plain["kept_rank"][b, c, :k], fast["kept_rank"][b, c, :k])
    for key in ("boxes", "scores", "classes", "count"):
        np.testing.assert_array_equal(plain[key], fast[key])
    return fast


def _run_nms(per_image, num_classes, max_boxes, iou_thr, with_rank, bucketed):
    """per_image: list of (rec [N,8], cand [n], class_count [C]) with a common N."""
    lib = _lib.load()
    B = len(per_image)
    N = per_image[0][0].shape[0]
    rec = torch.from_numpy(np.stack([p[0] for p in per_image])).cuda()
    cand = np.zeros((B, N), dtype=np.int32)
    cc = np.zeros(B, dtype=np.int32)
    clc = np.zeros((B, num_classes), dtype=np.int32)
    for b, (_, c, k) in enumerate(per_image):
        cand[b, :len(c)] = c
        cc[b] = len(c)
        clc[b] = k
    cand_t, cc_t, clc_t = (torch.from_numpy(a).cuda() for a in (cand, cc, clc))
    kept_idx = torch.full((B, num_classes, max_boxes), -1, dtype=torch.int32, device="cuda")
    kept_rank = torch.full((B, num_classes, max_boxes), -1, dtype=torch.int32, device="cuda")
    kept_count = torch.full((B, num_classes), -1, dtype=torch.int32, device="cuda")
    if bucketed:
        wsp = torch.full((int(lib.y3_nms_workspace_ints(B, N)),), 12345, dtype=torch.int32, device="cuda")
        _lib.check(lib.y3_nms_bucketed(rec.data_ptr(), cand_t.data_ptr(), cc_t.data_ptr(), clc_t.data_ptr(), B, N,
                                       num_classes, max_boxes, float(np.float32(iou_thr)), wsp.data_ptr(),
                                       kept_idx.data_ptr(), kept_rank.data_ptr() if with_rank else None,
                                       kept_count.data_ptr(), stream()))
    else:
        _lib.check(lib.y3_nms(rec.data_ptr(), cand_t.data_ptr(), cc_t.data_ptr(), clc_t.data_ptr(), B, N, num_classes,
                              max_boxes, float(np.float32(iou_thr)), kept_idx.data_ptr(),
                              kept_rank.data_ptr() if with_rank else None, kept_count.data_ptr(), stream()))
    cap = num_classes * max_boxes
    ob = torch.zeros((B, cap, 4), dtype=torch.float32, device="cuda")
    os_ = torch.zeros((B, cap), dtype=torch.float32, device="cuda")
    oc = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    on = torch.zeros(B, dtype=torch.int32, device="cuda")
    _lib.check(lib.y3_pack_outputs(rec.data_ptr(), kept_idx.data_ptr(), kept_count.data_ptr(), B, N, num_classes,
                                   max_boxes, ob.data_ptr(), os_.data_ptr(), oc.data_ptr(), on.data_ptr(), stream()))
    torch.cuda.synchronize()
    return dict(kept_idx=kept_idx.cpu().numpy(), kept_rank=kept_rank.cpu().numpy(), kept_count=kept_count.cpu().numpy(),
                boxes=ob.cpu().numpy(), scores=os_.cpu().numpy(), classes=oc.cpu().numpy(), count=on.cpu().numpy())


def run_nms_bitmask(per_image, num_classes, max_boxes, iou_thr, mask_words=None):
    """The experimental all-pairs bitmask NMS (libyolo3hip_exp.so: y3_nms_bitmask) on the input format of run_nms.
    Returns kept_idx [B,C,max_boxes], kept_count [B,C], status [B]."""
    with _lib.experimental() as lib:
        B = len(per_image)
        N = per_image[0][0].shape[0]
        rec = torch.from_numpy(np.stack([p[0] for p in per_image])).cuda()
        cand = np.zeros((B, N), dtype=np.int32)
        cc = np.zeros(B, dtype=np.int32)
        clc = np.zeros((B, num_classes), dtype=np.int32)
        for b, (_, c, k) in enumerate(per_image):
            cand[b, :len(c)] = c
            cc[b] = len(c)
            clc[b] = k
        if mask_words is None:
            mask_words = int(max(1, (clc.astype(np.int64) * ((clc + 63) // 64)).sum(axis=1).max()))
        cand_t, cc_t, clc_t = (torch.from_numpy(a).cuda() for a in (cand, cc, clc))
        kept_idx = torch.full((B, num_classes, max_boxes), -1, dtype=torch.int32, device="cuda")
        kept_count = torch.full((B, num_classes), -1, dtype=torch.int32, device="cuda")
        status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        wsp = torch.full((int(lib.y3_nms_workspace_ints(B, N)),), 12345, dtype=torch.int32, device="cuda")
        mask = torch.full((B * mask_words,), -1, dtype=torch.int64, device="cuda")
        _lib.check(lib.y3_nms_bitmask(rec.data_ptr(), cand_t.data_ptr(), cc_t.data_ptr(), clc_t.data_ptr(), B, N, num_classes,
                                      max_boxes, float(np.float32(iou_thr)), wsp.data_ptr(), mask.data_ptr(), mask_words,
                                      status.data_ptr(), kept_idx.data_ptr(), kept_count.data_ptr(), stream()))
        torch.cuda.synchronize()
        return dict(kept_idx=kept_idx.cpu().numpy(), kept_count=kept_count.cpu().numpy(), status=status.cpu().numpy())
