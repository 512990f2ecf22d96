#!/bin/bash
# Bench lines of the BASELINE configurations on one GPU: tools/bench_configs.sh <tag>  ->  gpurun_out/<tag>/bench_*_line.json
set -e
TAG=${1:-cfg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $OUT/bench_${name}_line.json 2> $OUT/bench_${name}.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_${name}_line.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$name', d['value'], d['ms_per_step'], r.get('kernel'), r.get('achieved'))"; }
run b64_608_bf16 --steps 20 --warmup 5
run b64_608_bf16_unfused_decode --steps 20 --warmup 5 --unfused-decode --no-cpu-baseline
run b64_608_bf16_hipgraph --steps 20 --warmup 5 --graph --no-cpu-baseline
run config3_416_b64_bf16 --steps 20 --warmup 5 --size 416 --no-cpu-baseline
run config2_tiny416_b32_fp32 --steps 20 --warmup 5 --model yolov3-tiny --size 416 --batch 32 --dtype fp32 --no-cpu-baseline
run tiny416_b64_bf16 --steps 20 --warmup 5 --model yolov3-tiny --size 416 --no-cpu-baseline
run config5_dense_stress --steps 20 --warmup 5 --score-threshold 0.001 --no-cpu-baseline
