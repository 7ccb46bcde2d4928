#!/bin/bash
# build_prof_variant.sh <name> <unit.hip[,unit2.hip...]> <-DFLAG ...>: a diagnostic build of some units linked with the product's other
# objects -> build/prof/libloamx_<name>.so (git-ignored; travels to the GPU box); use with LOAMX_LIB=build/prof/libloamx_<name>.so
set -eu
name=$1; units=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/loam_velodyne_amd/csrc
make -s -j8
mkdir -p $root/build/prof
objs=$(ls *.o)
new=""
for unit in ${units//,/ }; do
  obj=$root/build/prof/${unit%.hip}_$name.o
  /opt/rocm/bin/hipcc "$@" -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-result -I/opt/rocm/include -c $unit -o $obj &
  objs=$(echo "$objs" | grep -v "^${unit%.hip}.o$")
  new="$new $obj"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/prof/libloamx_$name.so $objs $new -L/opt/rocm/lib -lrocprofiler-sdk-roctx -lrccl -lhsa-runtime64 -Wl,-rpath,/opt/rocm/lib
echo built build/prof/libloamx_$name.so
