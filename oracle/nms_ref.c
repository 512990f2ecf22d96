/* Oracle (test infrastructure): greedy per-class non-max suppression, plain C fp32.
 *
 * Restates the CPU kernel behind tf.image.non_max_suppression as the reference
 * calls it (inference.py:148-152; tensorflow 1.12.0,
 * core/kernels/non_max_suppression_op.cc -- a pinned third-party dependency,
 * requirements.txt:23, absent from /root/reference, so restated from its
 * published algorithm).  PARITY UNPINNED: no golden vectors exist upstream.
 *
 *   - candidates visited in descending score; equal scores: lower index first
 *     (TF 1.12's priority_queue leaves this unspecified; later TF pins it so);
 *   - a candidate is selected iff IoU(candidate, s) > thr is false for every s
 *     already selected (strict >);
 *   - stop once max_out are selected;
 *   - IoU in fp32, corners normalised by min/max, area <= 0 => IoU = 0.
 * Build with -ffp-contract=off so no multiply-add is fused.
 */
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int idx; } cand_t;

static int cand_cmp(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

float y3o_iou(const float *bi, const float *bj) {
    const float ymin_i = fminf_(bi[0], bi[2]), xmin_i = fminf_(bi[1], bi[3]);
    const float ymax_i = fmaxf_(bi[0], bi[2]), xmax_i = fmaxf_(bi[1], bi[3]);
    const float ymin_j = fminf_(bj[0], bj[2]), xmin_j = fminf_(bj[1], bj[3]);
    const float ymax_j = fmaxf_(bj[0], bj[2]), xmax_j = fmaxf_(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iymin = fmaxf_(ymin_i, ymin_j), ixmin = fmaxf_(xmin_i, xmin_j);
    const float iymax = fminf_(ymax_i, ymax_j), ixmax = fminf_(xmax_i, xmax_j);
    const float inter = fmaxf_(iymax - iymin, 0.0f) * fmaxf_(ixmax - ixmin, 0.0f);
    return inter / (area_i + area_j - inter);
}

/* boxes [n][4] = (y1,x1,y2,x2); returns the number selected, indices in `selected`. */
int y3o_nms(const float *boxes, const float *scores, int n, int max_out, float iou_thr, int *selected) {
    if (n <= 0 || max_out <= 0) return 0;
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)n);
    if (!c) return -1;
    for (int i = 0; i < n; ++i) { c[i].score = scores[i]; c[i].idx = i; }
    qsort(c, (size_t)n, sizeof(cand_t), cand_cmp);
    int ns = 0;
    for (int k = 0; k < n && ns < max_out; ++k) {
        const float *b = boxes + 4 * (size_t)c[k].idx;
        int keep = 1;
        for (int j = ns - 1; j >= 0; --j) {
            if (y3o_iou(b, boxes + 4 * (size_t)selected[j]) > iou_thr) { keep = 0; break; }
        }
        if (keep) selected[ns++] = c[k].idx;
    }
    free(c);
    return ns;
}
