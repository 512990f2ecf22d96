#!/bin/bash
# The round's evidence in one call: per-layer reports and the bench lines of every BASELINE configuration that fits one GPU.
# usage: tools/round_evidence.sh <tag>   ->  gpurun_out/<tag>/
set -e
TAG=${1:-evidence}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 150 python tools/layer_report.py --reps 7 > $OUT/layer_report_b64_608.txt 2>&1
timeout -k 10 150 python tools/layer_report.py --reps 7 --decode > $OUT/layer_report_b64_608_decode.txt 2>&1
timeout -k 10 150 python tools/layer_report.py --reps 7 --size 416 > $OUT/layer_report_b64_416.txt 2>&1
timeout -k 10 280 python bench.py --steps 20 > $OUT/bench_b64_608_bf16_line.json 2> $OUT/bench.err
timeout -k 10 200 python bench.py --steps 20 --graph --no-cpu-baseline > $OUT/bench_b64_608_bf16_hipgraph_line.json 2>> $OUT/bench.err
timeout -k 10 200 python bench.py --steps 20 --size 416 --no-cpu-baseline > $OUT/bench_config3_416_b64_bf16_line.json 2>> $OUT/bench.err
timeout -k 10 200 python bench.py --steps 20 --model yolov3-tiny --size 416 --batch 32 --dtype fp32 --no-cpu-baseline > $OUT/bench_config2_tiny416_b32_fp32_line.json 2>> $OUT/bench.err
timeout -k 10 200 python bench.py --steps 20 --model yolov3-tiny --size 416 --batch 64 --no-cpu-baseline > $OUT/bench_tiny416_b64_bf16_line.json 2>> $OUT/bench.err
timeout -k 10 200 python bench.py --steps 20 --score-threshold 0.001 --no-cpu-baseline > $OUT/bench_config5_dense_b64_608_bf16_line.json 2>> $OUT/bench.err
for f in $OUT/bench_*_line.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("achieved"), d.get("roofline", {}).get("frac"))
PY
done
tail -n 1 $OUT/layer_report_b64_608.txt $OUT/layer_report_b64_416.txt
