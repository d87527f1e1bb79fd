#!/bin/bash
# A/B on the GPU box: <script> alternately with the library under test (in-tree build) and GTOK_LIB=<baseline .so>, <n> rounds.
# Usage: gpurun -- 'bash profiles/tools/ab_time.sh glearning-benchmark_amd/csrc/_ab/libgtok_base.so profiles/tools/time_lane_ab.py 3'
base=$1; script=$2; n=${3:-3}
for i in $(seq $n); do
  echo "--- round $i: baseline"; GTOK_LIB=$PWD/$base python3 $script 2>&1 | grep -v amdgpu.ids
  echo "--- round $i: new";      python3 $script 2>&1 | grep -v amdgpu.ids
done
