/* yolo3hip_experimental.h -- knobs of libyolo3hip_exp.so, the -DY3_EXPERIMENTS flavour of libyolo3hip.so.
 *
 * NOT part of the product ABI and absent from libyolo3hip.so.  The experimental flavour is the same source built
 * with -DY3_EXPERIMENTS; it exports everything include/yolo3hip.h declares plus the functions below; the setters write
 * plain process-global variables read by every later launch on every thread (set them from one thread, between
 * launches).  Users: the tile-equality and kernel-variant tests (tests/test_gpu_conv.py, tests/test_gpu_decode.py),
 * tools/layer_report.py --tile/--phase and the A/B scripts under tools/.  The reference has no counterpart.
 */
#ifndef YOLO3HIP_EXPERIMENTAL_H
#define YOLO3HIP_EXPERIMENTAL_H

#include "yolo3hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Force the implicit-GEMM tile of y3_conv2d_fused: low 16 bits = tile (0 = heuristic, 128, 129, 130, 131, 192, 256,
 * 64, 257, 258, 260, 261: see the table at the top of csrc/conv_igemm.hip; 512 / 704 = conv_pp_kernel's 256- / 192-pixel
 * tile wherever it applies, while a forced 256 / 192 keeps conv_igemm_kernel); high 16 bits = debug bits (2 no
 * interleave, 4 ring-pipelined main loop, 64 LDS epilogue, 128 one workgroup per tile; 1/8/16/32 are TIMING
 * experiments that skip loads or the epilogue and produce garbage results). */
int y3_set_conv_tile(int tile);
/* Start offset of the de-phased persistent workgroups in percent of a tile period (default 50, 0 = off). */
int y3_set_conv_phase(int percent);
/* CUs one convolution launch may occupy (default: all; two streams with 128 each overlap two layers). */
int y3_set_conv_cus(int cus);
/* 1 = always the 16-lanes-per-box decode kernel, 0 = automatic (lane-per-box when shapes allow). */
int y3_decode_variant(int v);

/* The all-pairs IoU bitmask form of per-class NMS (BASELINE north_star's sketch), kept for the A/B against the greedy
 * kernels of y3_nms_bucketed (tools/nms_ab.py): bucket -> bitonic sort of each pair's candidates -> one
 * 64-bit mask word per (row, 64-candidate column block) -> one wavefront per pair scans with the suppressed set in
 * registers.  Same kept indices as y3_nms_bucketed (tests/test_gpu_nms.py).  workspace: y3_nms_workspace_ints() ints;
 * mask: batch * mask_words_per_image 64-bit words, an image needs sum over classes of n * ceil(n / 64); status[b] = 1
 * (and nothing kept for image b) when its masks do not fit, a class has more than 8192 candidates or bucketing failed. */
int y3_nms_bitmask(const float *records, const int32_t *cand_idx, const int32_t *cand_count, const int32_t *class_count,
                   int batch, int boxes_per_image, int num_classes, int max_boxes, float iou_threshold,
                   int32_t *workspace, uint64_t *mask, int64_t mask_words_per_image, int32_t *status,
                   int32_t *kept_idx, int32_t *kept_count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO3HIP_EXPERIMENTAL_H */
