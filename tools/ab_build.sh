#!/bin/bash
# Build an A/B flavour of the library: tools/ab_build.sh <name> <extra hipcc flags...>  ->  gpurun_ab/libyolo3hip_<name>.so
# (sources as they are in the tree; compare with tools/ab_libs.sh <rounds> main <name>).
set -e
name=$1; shift
cd "$(dirname "$0")/../yolo_amd/csrc"
mkdir -p build_ab_$name ../../gpurun_ab
for f in $(ls *.hip | sed "s/.hip$//"); do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -c $f.hip -o build_ab_$name/$f.o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o ../../gpurun_ab/libyolo3hip_$name.so build_ab_$name/*.o
ls -la ../../gpurun_ab/libyolo3hip_$name.so
