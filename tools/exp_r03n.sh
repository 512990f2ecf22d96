#!/bin/bash
# conv_pp_kernel<3> (192-pixel tiles) in the plan: tests, then same-box A/B against the library without it (608 and 416)
set -e
mkdir -p gpurun_out/r03n
timeout -k 10 600 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_e2e.py tests/test_gpu_layerwise.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r03n/test.log 2>&1 || { tail -30 gpurun_out/r03n/test.log; exit 1; }
tail -2 gpurun_out/r03n/test.log
for r in 1 2 3; do
  for v in main pp0; do
    if [ "$v" = main ]; then unset YOLO3HIP_LIB; else export YOLO3HIP_LIB=$PWD/gpurun_ab/libyolo3hip_$v.so; fi
    timeout -k 10 120 python tools/layer_report.py --reps 7 > gpurun_out/r03n/lr608_${v}_$r.txt 2>&1
    timeout -k 10 120 python tools/layer_report.py --reps 7 --size 416 > gpurun_out/r03n/lr416_${v}_$r.txt 2>&1
    echo $v $r 608: $(tail -n 1 gpurun_out/r03n/lr608_${v}_$r.txt | cut -c1-16) 416: $(tail -n 1 gpurun_out/r03n/lr416_${v}_$r.txt | cut -c1-16)
  done
done
