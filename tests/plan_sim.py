"""Test-only numpy interpreter of a planner.Plan: executes every op's *descriptor
semantics* (leading dimensions, channel slices, padding rule, residual, 2x store, packed
weight layout) on the CPU, with the arena simulated as one real byte buffer so that a
liveness / aliasing bug corrupts results.  It checks the planner and the weight packing
against the oracle without a GPU; it is not a fallback for the product path."""
import numpy as np

from yolo_amd import darknet_weights as dw


def _pack(kernel, bias):
    kh, kw, ci, co = kernel.shape
    return np.transpose(kernel, (3, 0, 1, 2)).reshape(co, -1), bias


class Sim(object):
    def __init__(self, plan, layers, params, images):
        assert plan.dtype == "fp32"
        self.plan, self.layers = plan, layers
        self.arena = np.zeros(plan.arena_bytes // 4 + 64, dtype=np.float32)
        self.arena[:] = np.nan
        self.inp = np.ascontiguousarray(images, dtype=np.float32)
        B = plan.batch
        self.heads = [np.full((B * h["h"] * h["w"] * h["ld"],), np.nan, dtype=np.float32) for h in plan.heads]
        self.w = {li: _pack(*dw.fold_batch_norm(params[li])) for li in params}

    def _flat(self, ref):
        buf = ref["buf"]
        if buf == "in":
            return self.inp.reshape(-1), ref["off"]
        if buf[0] == "arena":
            assert buf[1] % 16 == 0
            return self.arena, buf[1] // 4 + ref["off"]
        return self.heads[buf[1]], ref["off"]

    def read(self, ref, pixels):
        flat, off = self._flat(ref)
        idx = off + np.arange(pixels)[:, None] * ref["ld"] + np.arange(ref["c"])[None, :]
        return flat[idx]

    def write(self, ref, pix_idx, values):
        flat, off = self._flat(ref)
        c = values.shape[1]
        idx = off + np.asarray(pix_idx)[:, None] * ref["ld"] + np.arange(c)[None, :]
        flat[idx] = values

    def run(self):
        B = self.plan.batch
        for op in self.plan.ops:
            k = op["kind"]
            if k == "conv":
                ih, iw = op["in_hw"]; oh, ow = op["out_hw"]; ic, oc = op["in_c"], op["out_c"]
                ks, st = op["ksize"], op["stride"]
                pad = 1 if ks == 3 else 0
                x = self.read(op["x"], B * ih * iw).reshape(B, ih, iw, ic)
                xp = np.zeros((B, ih + 2 * pad + 2, iw + 2 * pad + 2, ic), dtype=np.float32)
                xp[:, pad:pad + ih, pad:pad + iw] = x
                cols = []
                for kh in range(ks):
                    for kw in range(ks):
                        cols.append(xp[:, kh:kh + (oh - 1) * st + 1:st, kw:kw + (ow - 1) * st + 1:st, :])
                a = np.concatenate(cols, axis=-1).reshape(B * oh * ow, ks * ks * ic)
                w, bias = self.w[op["layer"]]
                if w.shape[1] != ks * ks * ic:          # first layer: input channels zero-padded to one chunk
                    w = np.pad(w.reshape(oc, ks * ks, -1), ((0, 0), (0, 0), (0, ic - w.shape[1] // (ks * ks)))).reshape(oc, -1)
                y = a @ w.T + bias
                if op["leaky"]:
                    y = np.where(y > 0, y, np.float32(0.1) * y)
                if op["res"] is not None:
                    y = y + self.read(op["res"], B * oh * ow)
                y = y.astype(np.float32)
                if op.get("pool"):                      # 2x2 stride-2 max pool fused into the first convolution
                    y = y.reshape(B, oh // 2, 2, ow // 2, 2, oc).max(axis=(2, 4)).reshape(-1, oc)
                    self.write(op["y"], np.arange(B * (oh // 2) * (ow // 2)), y)
                elif op["ups"]:
                    b, r = np.divmod(np.arange(B * oh * ow), oh * ow)
                    ho, wo = np.divmod(r, ow)
                    for dy in (0, 1):
                        for dx in (0, 1):
                            self.write(op["y"], (b * 2 * oh + 2 * ho + dy) * 2 * ow + 2 * wo + dx, y)
                else:
                    self.write(op["y"], np.arange(B * oh * ow), y)
            elif k == "resblock":                     # 1x1 -> 3x3 -> + x, the 1x1's output staying inside the op
                h, w = op["hw"]; c = op["c"]
                i1, i3, _ = op["layers"]
                x = self.read(op["x"], B * h * w)
                w1, b1 = self.w[i1]
                w3, b3 = self.w[i3]
                y = x @ w1.T + b1
                if op["leaky"][0]:
                    y = np.where(y > 0, y, np.float32(0.1) * y)
                yp = np.zeros((B, h + 2, w + 2, c // 2), dtype=np.float32)
                yp[:, 1:1 + h, 1:1 + w] = y.reshape(B, h, w, c // 2)
                a = np.concatenate([yp[:, kh:kh + h, kw:kw + w, :] for kh in range(3) for kw in range(3)], axis=-1)
                z = a.reshape(B * h * w, 9 * (c // 2)) @ w3.T + b3
                if op["leaky"][1]:
                    z = np.where(z > 0, z, np.float32(0.1) * z)
                self.write(op["y"], np.arange(B * h * w), (x + z).astype(np.float32))
            elif k == "padin":
                x = self.read(op["x"], op["pixels"])
                y = np.zeros((op["pixels"], op["y"]["c"]), dtype=np.float32)
                y[:, :op["c"]] = x
                self.write(op["y"], np.arange(op["pixels"]), y)
            elif k == "maxpool":
                ih, iw = op["in_hw"]; c, s, z = op["c"], op["stride"], op["size"]
                oh, ow = -(-ih // s), -(-iw // s)
                ph = max((oh - 1) * s + z - ih, 0); pw = max((ow - 1) * s + z - iw, 0)
                x = self.read(op["x"], B * ih * iw).reshape(B, ih, iw, c)
                xp = np.full((B, ih + ph, iw + pw, c), -np.inf, dtype=np.float32)
                xp[:, ph // 2:ph // 2 + ih, pw // 2:pw // 2 + iw] = x
                y = np.full((B, oh, ow, c), -np.inf, dtype=np.float32)
                for dy in range(z):
                    for dx in range(z):
                        y = np.maximum(y, xp[:, dy:dy + (oh - 1) * s + 1:s, dx:dx + (ow - 1) * s + 1:s])
                self.write(op["y"], np.arange(B * oh * ow), y.reshape(-1, c))
            elif k == "upsample":
                ih, iw = op["in_hw"]; c = op["c"]
                x = self.read(op["x"], B * ih * iw).reshape(B, ih, iw, c)
                y = np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)
                self.write(op["y"], np.arange(B * 4 * ih * iw), y.reshape(-1, c))
            elif k == "copy":
                self.write(op["y"], np.arange(op["pixels"]), self.read(op["x"], op["pixels"]))
            elif k == "add":
                self.write(op["y"], np.arange(op["pixels"]), self.read(op["a"], op["pixels"]) + self.read(op["b"], op["pixels"]))
            else:
                raise AssertionError(k)
        out = []
        for hbuf, h in zip(self.heads, self.plan.heads):
            out.append(hbuf.reshape(B, h["h"], h["w"], h["ld"])[..., :h["c"]])
        return out
