// One box of the YOLO head decode, for a lane that owns the box: shared by decode_lane_kernel (decode.hip) and the
// head convolutions' fused epilogue (conv_igemm.hip), so that the two produce the same records bit for bit.
//   reference utils/utils.py:84-92 (get_head), 118-126 (correct_boxes), 150-154 (box_scores, arg-max classes).
// fp32 throughout, no multiply-add contraction (both files are built with -ffp-contract=off), IEEE division,
// operation order as in the reference.
#pragma once
#include "common.h"

struct DecodedBox {
    float ymin, xmin, ymax, xmax, score;
    unsigned m0, m1, m2;                 // classes whose probability equals the maximum (bit c of word c >> 5)
};

__device__ __forceinline__ float y3_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// tl: the box's 5 + C head values in LDS (fp32: tx, ty, tw, th, tobj, class logits); the logits that can reach the
// maximum are overwritten by their sigmoids.  Must be called by every lane of the wavefront (wave-level ballots);
// lanes without a box pass live = false and any readable tl.
__device__ __forceinline__ DecodedBox y3_decode_box(float *tl, bool live, int C, int gx, int gy, float stride_x, float stride_y,
                                                    float anchor_w, float anchor_h, float scale, float off_h, float off_w) {
    // (unrolled by 16: the LDS reads of a batch are in flight together instead of one round trip per class)
    float tmax = -__builtin_inff();
    {
        int c = 0;
        for (; c + 16 <= C; c += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = tl[5 + c + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) tmax = vmax_raw(tmax, v[u]);
        }
        for (; c < C; ++c) tmax = fmaxf(tmax, tl[5 + c]);
    }
    // sigmoid is increasing, so only logits close enough to the largest one that their fp32 sigmoid could
    // still reach the maximum need the exact evaluation: a logit more than `win` below tmax has a true
    // sigmoid at least 8 ulp below sigmoid(tmax) (win * sigmoid'(tmax) = 1e-6 * sigmoid(tmax)); the fp32
    // evaluation errs by 2-3.  Where sigmoid underflows (tmax < -80) every class is evaluated.
    const float win = tmax < -80.0f ? __builtin_inff() : 1e-6f * (1.0f + __expf(tmax)) + 1e-6f * fabsf(tmax);
    const float tcut = tmax - win;
    unsigned need[3] = {0u, 0u, 0u};
    {
        int c = 0;
        for (; c + 16 <= C; c += 16) {               // 16 never straddles a 32-bit mask word
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = tl[5 + c + u];
            unsigned bits = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) bits |= (v[u] >= tcut ? 1u : 0u) << u;
            const unsigned sh = bits << (c & 31);
            if ((c >> 5) == 0) need[0] |= sh; else if ((c >> 5) == 1) need[1] |= sh; else need[2] |= sh;
        }
        for (; c < C; ++c) {
            const unsigned bit = (tl[5 + c] >= tcut ? 1u : 0u) << (c & 31);
            if ((c >> 5) == 0) need[0] |= bit; else if ((c >> 5) == 1) need[1] |= bit; else need[2] |= bit;
        }
    }
    if (!live) need[0] = need[1] = need[2] = 0u;
    // exact sigmoid of the needed classes (usually one per box), written back over the logit
    float pmax = -1.0f;
    unsigned todo[3] = {need[0], need[1], need[2]};
    while (__builtin_amdgcn_ballot_w64((todo[0] | todo[1] | todo[2]) != 0u)) {
        const int w = todo[0] ? 0 : (todo[1] ? 1 : 2);
        const unsigned m = w == 0 ? todo[0] : (w == 1 ? todo[1] : todo[2]);
        if (m) {
            const int c = w * 32 + __ffs(m) - 1;
            const float pv = y3_sigmoid(tl[5 + c]);                                 // utils.py:87
            tl[5 + c] = pv;
            pmax = fmaxf(pmax, pv);                                                // utils.py:150
            if (w == 0) todo[0] = m & (m - 1); else if (w == 1) todo[1] = m & (m - 1); else todo[2] = m & (m - 1);
        }
    }
    DecodedBox o;
    o.m0 = o.m1 = o.m2 = 0u;
    todo[0] = need[0]; todo[1] = need[1]; todo[2] = need[2];
    while (__builtin_amdgcn_ballot_w64((todo[0] | todo[1] | todo[2]) != 0u)) {
        const int w = todo[0] ? 0 : (todo[1] ? 1 : 2);
        const unsigned m = w == 0 ? todo[0] : (w == 1 ? todo[1] : todo[2]);
        if (m) {
            const int bit = __ffs(m) - 1;
            if (tl[5 + w * 32 + bit] >= pmax) {                                    // utils.py:151
                if (w == 0) o.m0 |= 1u << bit; else if (w == 1) o.m1 |= 1u << bit; else o.m2 |= 1u << bit;
            }
            if (w == 0) todo[0] = m & (m - 1); else if (w == 1) todo[1] = m & (m - 1); else todo[2] = m & (m - 1);
        }
    }

    // utils.py:84-92
    const float bx = (y3_sigmoid(tl[0]) + (float)gx) * stride_x;
    const float by = (y3_sigmoid(tl[1]) + (float)gy) * stride_y;
    const float bw = expf(tl[2]) * anchor_w;
    const float bh = expf(tl[3]) * anchor_h;
    const float conf = y3_sigmoid(tl[4]);
    // utils.py:118-126
    const float hw = bw / 2.0f, hh = bh / 2.0f;
    o.ymin = ((by - hh) - off_h / 2.0f) / scale;
    o.xmin = ((bx - hw) - off_w / 2.0f) / scale;
    o.ymax = ((by + hh) - off_h / 2.0f) / scale;
    o.xmax = ((bx + hw) - off_w / 2.0f) / scale;
    o.score = conf * (pmax * 1.0f);                                                // utils.py:153-154
    return o;
}
