#!/bin/bash
# Round-3 A/B on one device: the ping-pong main loop (default for the 256x256 tile, or --tile 512 in the exp library)
# against conv_igemm_kernel<256,256> (--tile 256).  Results under gpurun_out/r03b/.
set -e
mkdir -p gpurun_out/r03b
for r in 1 2; do
  for t in 512 256; do
    timeout -k 10 150 python tools/layer_report.py --reps 7 --tile $t > gpurun_out/r03b/lr_tile${t}_$r.txt 2>&1
    echo tile=$t round=$r $(tail -n 1 gpurun_out/r03b/lr_tile${t}_$r.txt)
  done
done
timeout -k 10 150 python tools/layer_report.py --reps 7 > gpurun_out/r03b/lr_default.txt 2>&1
echo default $(tail -n 1 gpurun_out/r03b/lr_default.txt)
