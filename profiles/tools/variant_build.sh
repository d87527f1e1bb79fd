#!/bin/bash
# Build the WORKING TREE's libgtok.so with extra -D flags on gtok_sent.hip into csrc/_ab/libgtok_<name>.so (A/B timing of a
# kernel variant in one GPU call: GTOK_LIB=<that file> python3 <timing script>).  Usage: variant_build.sh <name> -DFLAG[=v] ...
set -e
name=$1; shift
root=$(git rev-parse --show-toplevel 2>/dev/null || pwd)
ab=$root/glearning-benchmark_amd/csrc/_ab; mkdir -p $ab
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c "$@" -I$root/include -o $tmp/gtok_sent.o $root/glearning-benchmark_amd/csrc/gtok_sent.hip
for f in gtok_ibtt gtok_rows gtok_csr; do cp $root/glearning-benchmark_amd/csrc/_obj/$f.hip.o $tmp/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $ab/libgtok_$name.so $tmp/*.o
rm -rf $tmp
echo $ab/libgtok_$name.so
