#!/bin/bash
# why is the balanced split slower inside the network?  Infinity-Cache pollution probe + non-temporal slab accesses
set -e
mkdir -p gpurun_out/r03g
for cfg in "0 0" "64 0" "0 1" "64 1" "0 0" "0 1"; do
  set -- $cfg
  timeout -k 10 150 python tools/pollute_probe.py $1 $2 > gpurun_out/r03g/pol_$1_$2.txt 2>&1
  echo "== pollute $1 balanced $2"; grep -E "@76\+r|@38\+r|@19\+r|total" gpurun_out/r03g/pol_$1_$2.txt
done
export YOLO3HIP_LIB=$PWD/gpurun_ab/libyolo3hip_nt.so
timeout -k 10 150 python tools/pollute_probe.py 0 1 > gpurun_out/r03g/pol_nt.txt 2>&1
echo "== nt slabs, balanced 1"; grep -E "@76\+r|@38\+r|@19\+r|total" gpurun_out/r03g/pol_nt.txt
