#!/bin/bash
# Round-3 ablation of the dominant kernel (libyolo3hip_exp.so debug bits, conv_igemm.hip): which part of a tile's
# time is the LDS-DMA of activations (bit 8), of weights (16), the epilogue (32).  Results under gpurun_out/r03a/.
set -e
mkdir -p gpurun_out/r03a
for dbg in 0 32 40 48 56 8 16; do
  t=$(( (dbg << 16) | 0 ))
  if [ $dbg = 0 ]; then args=""; else args="--tile $t"; fi
  timeout -k 10 150 python tools/layer_report.py --reps 7 $args > gpurun_out/r03a/lr_dbg$dbg.txt 2>&1
  echo dbg=$dbg $(tail -n 1 gpurun_out/r03a/lr_dbg$dbg.txt)
done
