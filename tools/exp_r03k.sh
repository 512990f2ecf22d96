#!/bin/bash
# two kernels (plain + split) against the single kernel that carried both, and against the pre-split library: one box
set -e
mkdir -p gpurun_out/r03k
timeout -k 10 500 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_e2e.py -m gpu -x -q --timeout 200 --timeout-method thread -k "pingpong or workspace or balanced" > gpurun_out/r03k/test.log 2>&1 || { tail -30 gpurun_out/r03k/test.log; exit 1; }
tail -2 gpurun_out/r03k/test.log
bash tools/ab_libs.sh 3 main pp0
cp gpurun_out/ab_main_2.txt gpurun_out/r03k/lr_two_kernels.txt
cp gpurun_out/ab_pp0_2.txt gpurun_out/r03k/lr_before_split.txt
