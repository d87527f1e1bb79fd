#!/bin/bash
# Cooperative-padding wait variants of sent_lane_kernel (csrc/_ab/libgtok_<name>.so built by variant_build.sh) against the in-tree
# build and round 3's library: headline form alternated (time_lane_ab.py), then the padded forms at scale (time_r04_variants.py).
#   gpurun -- 'bash profiles/tools/coop_sweep.sh "r03 w127 w64 w32 s64"'
ab=glearning-benchmark_amd/csrc/_ab
for round in 1 2 3; do
  python3 profiles/tools/time_lane_ab.py 2>&1 | grep -v amdgpu.ids
  for v in $1; do GTOK_LIB=$PWD/$ab/libgtok_$v.so python3 profiles/tools/time_lane_ab.py 2>&1 | grep -v amdgpu.ids; done
done
python3 profiles/tools/time_r04_variants.py 2>&1 | grep -v amdgpu.ids
for v in $1; do [ $v = r03 ] && continue; GTOK_LIB=$PWD/$ab/libgtok_$v.so python3 profiles/tools/time_r04_variants.py 2>&1 | grep -v amdgpu.ids; done
