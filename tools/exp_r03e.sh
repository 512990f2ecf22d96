#!/bin/bash
# balanced split in the plan: parity tests, then per-layer report and bench with / without the workspace (same box)
set -e
mkdir -p gpurun_out/r03l
timeout -k 10 500 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_e2e.py tests/test_gpu_layerwise.py -m gpu -x -q --timeout 200 --timeout-method thread -k "pingpong or workspace or balanced or e2e or teacher" > gpurun_out/r03l/test.log 2>&1 || { tail -30 gpurun_out/r03l/test.log; exit 1; }
tail -2 gpurun_out/r03l/test.log
timeout -k 10 200 python tools/pp_split_probe.py 20 | tee gpurun_out/r03l/probe.txt
for b in 0 1 0 1; do
  YOLO_AMD_BALANCED=$b timeout -k 10 150 python tools/layer_report.py --reps 5 > gpurun_out/r03l/lr_bal$b.txt 2>&1
  echo balanced=$b $(tail -n 1 gpurun_out/r03l/lr_bal$b.txt)
done
for b in 0 1; do
  YOLO_AMD_BALANCED=$b timeout -k 10 280 python bench.py > gpurun_out/r03l/bench_bal$b.json 2> gpurun_out/r03l/bench_bal$b.err
  echo balanced=$b $(python -c "import json,sys; d=json.loads(open('gpurun_out/r03l/bench_bal$b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'])")
done
