#!/bin/bash
# Build libgtok.so as of a git ref into glearning-benchmark_amd/csrc/_ab/libgtok_<name>.so (A/B timing of kernel changes in one
# GPU call: run a timing script once with GTOK_LIB=<that file> and once without).  Usage: ab_build.sh <git-ref> <name>
set -e
ref=$1; name=$2
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C "$root" archive "$ref" glearning-benchmark_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/glearning-benchmark_amd/csrc/_ab"
for f in gtok_sent gtok_ibtt gtok_rows; do
  [ -f "$tmp/glearning-benchmark_amd/csrc/$f.hip" ] || continue
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c -I"$tmp/include" -o "$tmp/$f.o" "$tmp/glearning-benchmark_amd/csrc/$f.hip" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$root/glearning-benchmark_amd/csrc/_ab/libgtok_$name.so" "$tmp"/*.o
rm -rf "$tmp"
echo "$root/glearning-benchmark_amd/csrc/_ab/libgtok_$name.so"
