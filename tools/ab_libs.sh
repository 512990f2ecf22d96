#!/bin/bash
# A/B of library builds in ONE process sequence on one device: tools/ab_libs.sh <rounds> <lib|main>...
# (box-to-box spread is larger than most kernel changes; YOLO3HIP_LIB selects the build).
set -e
rounds=$1; shift
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset YOLO3HIP_LIB; else export YOLO3HIP_LIB=$PWD/gpurun_ab/libyolo3hip_$v.so; fi
    timeout -k 10 120 python tools/layer_report.py --reps 7 > gpurun_out/ab_${v}_$r.txt 2>&1
    echo $v $r $(tail -n 1 gpurun_out/ab_${v}_$r.txt)
  done
done
