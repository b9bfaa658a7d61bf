#!/bin/bash
cd "$(dirname "$0")/../../.."
cp raytracers_amd/libray_mi355x.so /tmp/keep.so
for v in new d3 d4 new; do
  if [ $v = new ]; then cp /tmp/keep.so raytracers_amd/libray_mi355x.so; else cp build/lib_$v/libray_mi355x.so raytracers_amd/libray_mi355x.so; fi
  timeout 100 python profiles/r03/exp/part_d.py $v 2>&1 | grep -v amdgpu
done | tee gpurun_out/part_d.txt
cp /tmp/keep.so raytracers_amd/libray_mi355x.so
