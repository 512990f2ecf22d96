#!/bin/bash
set -e
mkdir -p gpurun_out/r03d
timeout -k 10 300 python -u -m pytest tests/test_gpu_conv.py -m gpu -x -q --timeout 200 --timeout-method thread -k "pingpong" > gpurun_out/r03d/test.log 2>&1 || { tail -30 gpurun_out/r03d/test.log; exit 1; }
tail -2 gpurun_out/r03d/test.log
for t in 512 256 $(( (32<<16)|512 )) $(( (56<<16)|512 )) $(( (40<<16)|512 )) $(( (48<<16)|512 )); do
  timeout -k 10 150 python tools/layer_report.py --reps 5 --tile $t > gpurun_out/r03d/lr_$t.txt 2>&1
  echo tile=$t $(tail -n 1 gpurun_out/r03d/lr_$t.txt)
done
