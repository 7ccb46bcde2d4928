#!/bin/bash
# build_prof_variant.sh <name> <unit.hip> <-DFLAG ...>: a diagnostic build of ONE unit linked with the product's other objects
# -> build/prof/libloamx_<name>.so (git-ignored; travels to the GPU box); use with LOAMX_LIB=build/prof/libloamx_<name>.so
set -eu
name=$1; unit=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/loam_velodyne_amd/csrc
make -s -j8
mkdir -p $root/build/prof
obj=$root/build/prof/${unit%.hip}_$name.o
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-result -I/opt/rocm/include -c $unit -o $obj
objs=$(ls *.o | grep -v "^${unit%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/prof/libloamx_$name.so $objs $obj -L/opt/rocm/lib -lrocprofiler-sdk-roctx -lrccl -lhsa-runtime64 -Wl,-rpath,/opt/rocm/lib
echo built build/prof/libloamx_$name.so
