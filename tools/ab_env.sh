#!/bin/bash
# A/B of an environment switch on one device: tools/ab_env.sh <rounds> VAR v1 v2 ...   (layer_report per value)
set -e
rounds=$1; var=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    env $var=$v timeout -k 10 120 python tools/layer_report.py --reps 7 > gpurun_out/env_${var}_${v}_$r.txt 2>&1
    echo $var=$v $r $(tail -1 gpurun_out/env_${var}_${v}_$r.txt)
  done
done
