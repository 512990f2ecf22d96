/* yolo3hip_experimental.h -- knobs of libyolo3hip_exp.so, the -DY3_EXPERIMENTS flavour of libyolo3hip.so.
 *
 * NOT part of the product ABI and absent from libyolo3hip.so.  The experimental flavour is the same source built
 * with -DY3_EXPERIMENTS; it exports everything include/yolo3hip.h declares plus the setters below, which write
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
 * 64, 257, 258, 260, 261: see the table at the top of csrc/conv_igemm.hip); high 16 bits = debug bits (2 no
 * interleave, 4 ring-pipelined main loop, 64 LDS epilogue, 128 one workgroup per tile; 1/8/16/32 are TIMING
 * experiments that skip loads or the epilogue and produce garbage results). */
int y3_set_conv_tile(int tile);
/* Start offset of the de-phased persistent workgroups in percent of a tile period (default 50, 0 = off). */
int y3_set_conv_phase(int percent);
/* CUs one convolution launch may occupy (default: all; two streams with 128 each overlap two layers). */
int y3_set_conv_cus(int cus);
/* 1 = always the 16-lanes-per-box decode kernel, 0 = automatic (lane-per-box when shapes allow). */
int y3_decode_variant(int v);

#ifdef __cplusplus
}
#endif
#endif /* YOLO3HIP_EXPERIMENTAL_H */
