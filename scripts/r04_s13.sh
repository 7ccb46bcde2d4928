#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s13; mkdir -p $out
export TMPDIR=/tmp
for v in "" cap11 cap10; do
  lib=$root/loam_velodyne_amd/libloamx.so; [ -n "$v" ] && lib=$root/build/prof/libloamx_$v.so
  LOAMX_LIB=$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab ";" > $out/ab_$v.json 2> $out/ab_$v.err
  echo "variant [$v]"; grep "^\[ab\]" $out/ab_$v.err
done
LOAMX_LIB=$root/build/prof/libloamx_vb.so LOAMX_NO_LOOKAHEAD=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pcie --repeat 1 > $out/vb.json 2> $out/vb.err
grep -A5 "k_vb_reduce" $out/vb.err | tail -12
timeout 300 python -m pytest tests/test_gpu_voxbucket.py -m gpu -x -q 2>&1 | tail -2
