#!/bin/bash
# per-layer report for several forced tile variants on one device: tools/ab_tiles.sh <tile>...
set -e
mkdir -p gpurun_out
for t in "$@"; do
  timeout -k 10 120 python tools/layer_report.py --reps 7 --tile $t > gpurun_out/tile_$t.txt 2>&1
  echo $t $(tail -1 gpurun_out/tile_$t.txt)
done
