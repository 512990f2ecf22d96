#!/bin/bash
# Ablation of the ping-pong kernel (exp library debug bits: 8 no X pieces, 16 no W pieces, 32 no epilogue).
set -e
mkdir -p gpurun_out/r03c
for dbg in 0 32 40 48 56 24; do
  t=$(( (dbg << 16) | 512 ))
  timeout -k 10 150 python tools/layer_report.py --reps 5 --tile $t > gpurun_out/r03c/lr_dbg$dbg.txt 2>&1
  echo dbg=$dbg $(tail -n 1 gpurun_out/r03c/lr_dbg$dbg.txt)
done
