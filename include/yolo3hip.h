/* yolo3hip.h -- C-ABI of libyolo3hip.so: the MI355X (gfx950) kernels under the
 * YOLOv3 inference hot path of viplix3/YOLO.
 *
 * The reference has no FFI layer of its own (it is pure Python over TensorFlow
 * 1.12 ops); each entry point below replaces the TF op call
 * site(s) cited beside it.  Conventions:
 *   - plain pointers and ints only; every pointer is a DEVICE pointer unless it
 *     is a descriptor struct (host) or says "host";
 *   - the library never allocates or frees device memory and keeps no mutable state
 *     besides a thread-local error string (plus read-only per-device facts it caches: the
 *     compute-unit count, which kernels already have their LDS attribute set); there is no
 *     setter, knob or mode switch in it -- the tile/variant overrides used by A/B runs and by
 *     the tile-equality tests exist only in the -DY3_EXPERIMENTS flavour of the library
 *     (libyolo3hip_exp.so, include/yolo3hip_experimental.h); the caller owns every buffer;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     re-entrant across streams/devices, and does not synchronise;
 *   - return 0 on success, <0 on error (Y3_E_*); y3_last_error() describes it.
 * Activations are NHWC.  dtype codes: Y3_F32 = 0, Y3_BF16 = 1.
 */
#ifndef YOLO3HIP_H
#define YOLO3HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y3_VERSION 103

enum { Y3_F32 = 0, Y3_BF16 = 1 };

enum {
    Y3_OK = 0,
    Y3_E_INVALID = -1,     /* bad argument / unsupported shape            */
    Y3_E_ALIGN = -2,       /* pointer or leading dimension not aligned     */
    Y3_E_LAUNCH = -3       /* hipLaunch / runtime error                    */
};

enum {
    Y3_CONV_LEAKY = 1,     /* leaky-ReLU alpha=0.1 (build_model.py:174-178)                  */
    Y3_CONV_RESIDUAL = 2,  /* + residual after the activation (shortcut, :218-227)           */
    Y3_CONV_UPS2X = 4,     /* store each output pixel 2x2 (nearest upsample, :231-239)       */
    Y3_CONV_OUT_F32 = 8,   /* store fp32 regardless of compute dtype (head tensors)          */
    Y3_CONV_IN_F32 = 16,   /* input is fp32 regardless of compute dtype (first layer image)  */
    Y3_CONV_REVERSE = 32,  /* scheduling hint, same result: output tiles are walked last to first */
    Y3_CONV_POOL2 = 64     /* first layer only (3x3 s1, 3 fp32 channels, even size): `out` receives the 2x2 stride-2 max pool
                              of the activated output (tf.nn.max_pool, build_model.py:203-212), [B, out_h/2, out_w/2, out_ld]  */
};

/* One fused convolution: conv (+folded BN) + bias [+ leaky] [+ residual] [2x upsampled store].
 * Replaces tf.pad + tf.layers.conv2d + tf.layers.batch_normalization + tf.nn.leaky_relu
 * (build_model.py:151-178), optionally tf.math.add (:224) and
 * tf.image.resize_nearest_neighbor (:236); route/concat (:190-193) is expressed through
 * in_ld/out_ld (a tensor may be a channel slice of a wider buffer).
 * Padding rule (build_model.py:124,151-154): ksize 3 -> one zero row/col before the
 * first input row/col, and whatever the window needs after the last (stride 1: SAME;
 * stride 2: top/left pad then VALID); ksize 1 -> none. */
typedef struct y3_conv_desc {
    int32_t dtype;        /* Y3_F32 or Y3_BF16: element type of in/w/res/out and of the MFMA    */
    int32_t batch;
    int32_t in_h, in_w, in_c;
    int32_t in_ld;        /* elements between consecutive pixels of `in`  (>= in_c)             */
    int32_t out_h, out_w, out_c;
    int32_t out_ld;       /* elements between consecutive pixels of `out` (>= out_c rounded to 8)*/
    int32_t res_ld;       /* same for `residual`                                                */
    int32_t ksize;        /* 1 or 3                                                             */
    int32_t stride;       /* 1 or 2                                                             */
    int32_t k_pad;        /* row length of the packed weight matrix, see y3_conv_weight_layout  */
    int32_t n_pad;        /* rows of the packed weight matrix                                   */
    int32_t flags;        /* Y3_CONV_*                                                          */
} y3_conv_desc;

/* Packed weights: w[n][k], n < n_pad, k < k_pad, k = (kh*ksize + kw)*in_c + c, zero filled
 * beyond out_c / ksize^2*in_c; bias[n_pad] fp32.  The host folds BN.
 * Given in_c/out_c/ksize/dtype this fills k_pad, n_pad and returns which kernel family
 * will run (0 = MFMA implicit GEMM, 1 = direct).  Host-only, no GPU needed. */
int y3_conv_weight_layout(y3_conv_desc *desc);

/* `zeros` is a >=256-byte zero-filled device buffer (source of padding taps). */
int y3_conv2d_fused(const y3_conv_desc *desc, const void *in, const void *w, const float *bias,
                    const void *residual, void *out, const void *zeros, void *stream);

/* Split-K form of the same convolution for launches with few output tiles and a long k (small batches: the
 * reference runs one image per step, inference.py:243): `splits` workgroups per tile write fp32 partial sums to
 * the caller's workspace, a second kernel adds them in a fixed order (deterministic) and applies bias,
 * activation, shortcut and the store.  y3_conv_splitk_plan (host only) says how many splits this descriptor
 * gets (1 = none: y3_conv2d_splitk then is y3_conv2d_fused) and how large the workspace must be.  Results equal
 * y3_conv2d_fused's up to fp32 summation order. */
int y3_conv_splitk_plan(const y3_conv_desc *desc, int32_t *splits, int64_t *workspace_bytes);
int y3_conv2d_splitk(const y3_conv_desc *desc, const void *in, const void *w, const float *bias,
                     const void *residual, void *out, const void *zeros, void *workspace,
                     int64_t workspace_bytes, void *stream);

/* y3_conv2d_fused with a caller-owned workspace.  The 256 x 256-tile kernel (bf16, bf16 output: the 3x3 layers of 128
 * and more channels, build_model.py:151-178) walks whole tiles on one persistent workgroup per CU, which leaves a
 * partial last round (1 444 tiles on 256 CUs at 76 x 76, batch 64).  Given `slabs` it deals the k-steps evenly instead
 * and sums a tile that two workgroups share in order, the second continuing from the first one's fp32 accumulators:
 * results are bit-identical to y3_conv2d_fused.  y3_conv_workspace_bytes (host only) says how many bytes `slabs` must
 * hold for a descriptor (0: the launch does not use it; slabs may be NULL then, and always: NULL = y3_conv2d_fused).
 * `flags`: 1 KiB that is ZERO when the call is issued (e.g. one hipMemsetAsync per forward pass over one such block
 * per layer) and not shared with a launch that may overlap this one; `slabs` may be shared by launches on one stream.
 * Both 16-byte aligned. */
int y3_conv_workspace_bytes(const y3_conv_desc *desc, int64_t *slab_bytes);
int y3_conv2d_fused_ws(const y3_conv_desc *desc, const void *in, const void *w, const float *bias, const void *residual,
                       void *out, const void *zeros, void *slabs, int64_t slab_bytes, void *flags, void *stream);

/* Fused network stem, bf16 mode: the first three convolutions of Darknet-53 in one launch, the
 * 32-channel full-resolution tensor between the first two never leaving LDS:
 *   conv A 3x3 s1 3->c0 on the fp32 NHWC image, conv B 3x3 s2 c0->c1, optional conv C 1x1 c1->c2
 * (build_model.py:113-179 applied to yolov3.cfg layers 0,1,2), each + bias [+ leaky].
 * Only c0 = 32, c1 = 64, c2 = 0 or 32.  Weights are in the layouts y3_conv_weight_layout gives
 * the three layers on their own: w0 fp32 [27][n0_pad] (direct family), w1 bf16 [64][k1_pad],
 * w2 bf16 [32][k2_pad].  out1 = conv B output (bf16, out1_ld), out2 = conv C output. */
typedef struct y3_stem_desc {
    int32_t batch, in_h, in_w;
    int32_t c0, c1, c2;
    int32_t n0_pad, k1_pad, k2_pad;
    int32_t out1_ld, out2_ld;
    int32_t flags;        /* bit 0/1/2: leaky-ReLU after conv A/B/C */
} y3_stem_desc;
int y3_conv_stem(const y3_stem_desc *desc, const float *image, const float *w0, const float *b0,
                 const void *w1, const float *b1, const void *w2, const float *b2,
                 void *out1, void *out2, void *stream);

/* Fused residual block, bf16: out = x + act2(conv3x3(act1(conv1x1(x; w1) + b1); w2) + b2) in one launch -- the
 * [convolutional 1x1 c -> c/2], [convolutional 3x3 c/2 -> c] and [shortcut from=-3] sections that make up a Darknet-53
 * residual block (build_model.py:113-179 twice, then :218-227).  The c/2-channel tensor between the two convolutions
 * never leaves LDS and x is fetched once.  w1/b1 and w2/b2 are the two layers' packed weights exactly as
 * y3_conv_weight_layout / y3_conv2d_fused take them; the result is bit-identical to y3_conv2d_fused(1x1) followed by
 * y3_conv2d_fused(3x3, Y3_CONV_RESIDUAL) on the same operands.  Supported: dtype Y3_BF16, tensors below 4 GiB, and
 * c = 256 with w a multiple of 19 (the 76 x 76 maps of a 608 input), c = 128 at any width, c = 64 at any width.
 * c = 64 is Darknet-53's first block, whose 1x1 y3_conv_stem otherwise computes as its third convolution: there the
 * 1x1 sums its 64 input channels in y3_conv_stem's order (bit-identical to y3_conv_stem(c2 = 32) followed by the 3x3;
 * against y3_conv2d_fused(1x1), which adds the same products in channel order, about one output in 10^4 differs by one
 * bf16 ulp), so a host may move that 1x1 between the two launches freely.  flags: 1 = leaky-ReLU after the 1x1,
 * 2 = after the 3x3, Y3_CONV_REVERSE = scheduling hint.  y3_resblock_tiles (host only) says whether a shape is
 * supported (Y3_OK) and how many workgroup tiles it has, so that a planner can decide when fusing pays.
 * x and out must not overlap (a tile reads its neighbours' halo pixels of x): overlapping ranges are refused with
 * Y3_E_INVALID.  y3_conv2d_fused, by contrast, allows residual == out (each element is loaded before it is stored, by
 * the same lane). */
typedef struct y3_resblock_desc {
    int32_t dtype;        /* Y3_BF16                                                                  */
    int32_t batch, h, w;  /* NHWC activations, same size in and out                                    */
    int32_t c;            /* channels of x and out; the 1x1 produces c/2                               */
    int32_t x_ld, out_ld; /* elements between consecutive pixels                                       */
    int32_t k1_pad, n1_pad, k2_pad, n2_pad;   /* packed layouts of the two weight matrices             */
    int32_t flags;
} y3_resblock_desc;
int y3_resblock_tiles(const y3_resblock_desc *desc, int64_t *tiles);
int y3_resblock_fused(const y3_resblock_desc *desc, const void *x, const void *w1, const float *b1,
                      const void *w2, const float *b2, void *out, const void *zeros, void *stream);

/* Name (prefix) of the kernel a descriptor will run on, as rocprofv3 reports it. Host-only. */
int y3_conv_kernel_name(const y3_conv_desc *desc, char *buf, int len);

/* tf.nn.max_pool(padding='SAME'), build_model.py:207-210.  c % (16/elsize) == 0. */
int y3_maxpool2d(int dtype, const void *in, void *out, int batch, int in_h, int in_w, int c,
                 int in_ld, int out_ld, int size, int stride, void *stream);

/* Fallbacks for cfgs the planner cannot fuse: nearest 2x upsample (build_model.py:236),
 * channel-slice copy for route/concat (:191), residual add (:224). */
int y3_upsample2x(int dtype, const void *in, void *out, int batch, int in_h, int in_w, int c,
                  int in_ld, int out_ld, void *stream);
int y3_copy_channels(int dtype, const void *in, void *out, int64_t pixels, int c,
                     int in_ld, int out_ld, void *stream);
int y3_shortcut_add(int dtype, const void *a, const void *b, void *out, int64_t pixels, int c,
                    int a_ld, int b_ld, int out_ld, void *stream);

/* ---- decode -------------------------------------------------------------------------
 * Fused get_head + correct_boxes + get_boxes_and_scores + score threshold
 * (utils/utils.py:43-160, inference.py:124-127) over up to 3 scales.
 * Per image the pooled box index is p = scale_base + (gy*g + gx)*num_anchors + a with the
 * scales in the order given (coarse first, inference.py:110-125).
 * records[b][p] (32 bytes): f32 y_min, x_min, y_max, x_max, score, u32 class bitmask[3]
 *   score = conf * max_c(prob_c); bit c set iff prob_c >= max (all tied classes,
 *   utils/utils.py:150-154).  num_classes <= 96.
 * cand_idx[b][0..cand_count[b]) : pooled indices with score >= score_threshold (any order).
 * class_count[b][c] : number of those candidates whose bitmask has class c.
 * cand_count and class_count must be zeroed by the caller before the call. */
typedef struct y3_decode_desc {
    int32_t batch;
    int32_t num_scales;          /* 1..3                                                     */
    int32_t num_classes;
    int32_t num_anchors;         /* anchors per scale (3)                                    */
    int32_t grid_h[3], grid_w[3];
    int32_t head_ld[3];          /* floats between consecutive cells of head l               */
    float   anchors[3][3][2];    /* [scale][anchor](w,h), already mask-selected              */
    float   stride_x[3], stride_y[3];   /* floor(input_w / grid_w), floor(input_h / grid_h)  */
    float   score_threshold;
    int32_t boxes_per_image;     /* sum over scales of gh*gw*num_anchors                     */
} y3_decode_desc;

/* letterbox[b] = {scale, offset_h, offset_w, 0} per image (utils/utils.py:114-116). */
int y3_decode(const y3_decode_desc *desc, const float *head0, const float *head1, const float *head2,
              const float *letterbox, float *records, int32_t *cand_idx, int32_t *cand_count,
              int32_t *class_count, void *stream);

/* A head convolution (build_model.py:113-179 on the 1x1 that feeds a [yolo] section) with the decode of ITS scale
 * (utils/utils.py:43-160, inference.py:110-127: exactly what y3_decode does for scale `scale` of `desc`) fused into
 * the epilogue: the finished fp32 head values are decoded where they are staged for the store, with y3_decode's own
 * per-box code, so records / candidates / counts are those y3_decode would produce (candidate order aside, which
 * nothing depends on).  `out` (fp32 [B,g,g,n_pad]) may be NULL: then the head tensor itself is never written.
 * Calling it for every scale of `desc` replaces the y3_decode call; cand_count / class_count zeroed beforehand as
 * for y3_decode.  Only for launches that run on the 256- or 192-pixel x 256-channel tiles (bf16 network, 3 anchors,
 * <= 80 classes, large batch): y3_conv_head_decode_ok (host only) tells, and a host falls back to
 * y3_conv2d_fused + y3_decode for ALL scales when any of them does not qualify. */
int y3_conv_head_decode_ok(const y3_conv_desc *conv, const y3_decode_desc *desc, int scale);
int y3_conv2d_head_decode(const y3_conv_desc *conv, const void *in, const void *w, const float *bias, void *out,
                          const void *zero_page, const y3_decode_desc *desc, int scale, const float *letterbox,
                          float *records, int32_t *cand_idx, int32_t *cand_count, int32_t *class_count, void *stream);

/* Dense forms kept for API parity with the reference's helper functions. */
/* get_head (utils/utils.py:43-96) for one scale: xy,wh [cells*A*2], conf [cells*A], cls [cells*A*C]. */
int y3_get_head(const float *head, int batch, int grid_h, int grid_w, int head_ld, int num_anchors,
                int num_classes, const float *anchors_wh, float stride_x, float stride_y,
                float *box_xy, float *box_wh, float *conf, float *cls, void *stream);
/* correct_boxes (utils/utils.py:100-129): n boxes, one letterbox triple. */
int y3_correct_boxes(const float *box_xy, const float *box_wh, int64_t n, float scale, float off_h,
                     float off_w, float *boxes, void *stream);
/* records -> dense boxes [n][4] and box_scores [n][num_classes] (utils/utils.py:148-159). */
int y3_expand_scores(const float *records, int64_t n, int num_classes, float *boxes, float *box_scores,
                     void *stream);

/* ---- NMS ----------------------------------------------------------------------------
 * Per-(image, class) greedy NMS == 80 x tf.image.non_max_suppression (inference.py:136-152):
 * candidates of class c = cand_idx entries whose record has bit c; visited by descending
 * score (ties: lower pooled index first); selected iff IoU > iou_threshold with no earlier
 * selected box; at most max_boxes (<= 256).  Persistent workgroups walk the (image, class)
 * pairs; the class's candidate list and its alive map are LDS-resident (so
 * boxes_per_image <= 30720), IoU tests are wave-parallel.
 * kept_idx[b][c][k]  pooled index of the k-th selected box (k < kept_count[b][c]);
 * kept_rank (nullable) the same box as an index into the class-masked array (tf nms_index). */
int y3_nms(const float *records, const int32_t *cand_idx, const int32_t *cand_count,
           const int32_t *class_count, int batch, int boxes_per_image, int num_classes, int max_boxes,
           float iou_threshold, int32_t *kept_idx, int32_t *kept_rank, int32_t *kept_count,
           void *stream);
/* Same result as y3_nms, faster: the candidates are first bucketed by class (one pass per image) into
 * `workspace` (y3_nms_workspace_ints(batch, boxes_per_image) int32, caller-owned, contents don't matter),
 * then every (image, class) pair of <= 128 candidates is suppressed by ONE wavefront with its candidates in
 * registers, pairs of <= 1024 by one 256-thread workgroup (four candidates per thread, one barrier per round), both with the greedy
 * order and predicate above; larger pairs take the LDS-resident workgroup kernel of y3_nms. */
int64_t y3_nms_workspace_ints(int batch, int boxes_per_image);
int y3_nms_bucketed(const float *records, const int32_t *cand_idx, const int32_t *cand_count,
                    const int32_t *class_count, int batch, int boxes_per_image, int num_classes, int max_boxes,
                    float iou_threshold, int32_t *workspace, int32_t *kept_idx, int32_t *kept_rank,
                    int32_t *kept_count, void *stream);
/* y3_nms_bucketed without the workspace argument, for callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * class-bucketed candidate lists live in a device array built into the library instead: 8 MiB
 * (Y3_NMS_STATIC_MAX_INTS int32) that every GPU the library runs on holds until the process ends.  The call
 * allocates and frees nothing; it returns Y3_E_INVALID when batch * (boxes_per_image + 128) exceeds
 * Y3_NMS_STATIC_MAX_INTS (64 images of 608x608 need 1 463 744).  The first call on a device looks the array's
 * address up (make it outside graph capture).  All calls on one GPU share that array: calls that overlap on one
 * GPU (two streams, two threads) race -- serialise them or use y3_nms_bucketed with workspaces of your own.
 * Same results as y3_nms_bucketed / y3_nms, bit for bit. */
#define Y3_NMS_STATIC_MAX_INTS (2 * 1024 * 1024)
int y3_nms_bucketed_static(const float *records, const int32_t *cand_idx, const int32_t *cand_count,
                           const int32_t *class_count, int batch, int boxes_per_image, int num_classes, int max_boxes,
                           float iou_threshold, int32_t *kept_idx, int32_t *kept_rank, int32_t *kept_count,
                           void *stream);
/* dynamic LDS the NMS kernel takes for this pool size (host-only helper). */
int y3_nms_lds_bytes(int boxes_per_image);

/* gather + class-major concat (inference.py:153-167): per image up to num_classes*max_boxes rows.
 * out_boxes [b][cap][4], out_scores [b][cap], out_classes [b][cap], out_count [b]; cap = num_classes*max_boxes. */
int y3_pack_outputs(const float *records, const int32_t *kept_idx, const int32_t *kept_count, int batch,
                    int boxes_per_image, int num_classes, int max_boxes, float *out_boxes,
                    float *out_scores, int32_t *out_classes, int32_t *out_count, void *stream);

/* Letterbox pre-process on the device: resize_image (utils/utils.py:7-40: aspect-preserving INTER_CUBIC
 * resize to (int(w*scale), int(h*scale)), centre pad with 128) + the /255. of inference.py:242.
 * image: uint8 RGB [h][w][3] on the device, rows row_stride_bytes apart; out: fp32 [in_h][in_w][3]. */
int y3_letterbox_u8(const uint8_t *image, int h, int w, int row_stride_bytes, float *out, int in_h, int in_w,
                    void *stream);

/* Network input staging: fp32 [pixels][cin] (cin <= 8 bf16 / 4 fp32) -> dtype [pixels][one 16-byte chunk],
 * zero padded, so that the first convolution (cin = 3, build_model.py:86) runs on the MFMA kernel. */
int y3_pad_input(int dtype, const float *in, void *out, int64_t pixels, int cin, void *stream);

/* fp32 -> bf16 round-to-nearest-even on device (weight/activation staging). */
int y3_cast_f32_to_bf16(const float *in, void *out, int64_t n, void *stream);
int y3_cast_bf16_to_f32(const void *in, float *out, int64_t n, void *stream);

int y3_version(void);
const char *y3_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* YOLO3HIP_H */
