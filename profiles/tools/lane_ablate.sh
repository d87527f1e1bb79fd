#!/bin/bash
# Time-based phase budget of sent_lane_kernel: the library is built once per -DGTOK_ABLATE_<phase> (the phase is cut out: tokens are
# WRONG, only the time means something) into csrc/_ab/, then profiles/tools/time_lane_ab.py runs with each.  Two steps:
#   bash profiles/tools/lane_ablate.sh build                      (here: hipcc cross-compiles)
#   gpurun -- 'bash profiles/tools/lane_ablate.sh run'            (on the GPU box)
root=$(git rev-parse --show-toplevel 2>/dev/null || pwd)
ab=$root/glearning-benchmark_amd/csrc/_ab
variants="STORES PHILOX PICK BRACKET"
if [ "$1" = build ]; then
  mkdir -p $ab
  for v in $variants; do
    ( tmp=$(mktemp -d)
      for f in gtok_sent gtok_ibtt gtok_rows; do
        obj=$root/glearning-benchmark_amd/csrc/_obj/$f.hip.o
        if [ $f = gtok_sent ]; then
          /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c -DGTOK_ABLATE_$v -I$root/include -o $tmp/$f.o $root/glearning-benchmark_amd/csrc/$f.hip
        else cp $obj $tmp/$f.o; fi
      done
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $ab/libgtok_no_$v.so $tmp/*.o; rm -rf $tmp ) &
  done
  wait; ls -la $ab
else
  python3 profiles/tools/time_lane_ab.py 2>&1 | grep -v amdgpu.ids
  for v in $variants; do GTOK_LIB=$ab/libgtok_no_$v.so python3 profiles/tools/time_lane_ab.py 2>&1 | grep -v amdgpu.ids; done
  python3 profiles/tools/time_lane_ab.py 2>&1 | grep -v amdgpu.ids
fi
