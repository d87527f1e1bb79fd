#!/bin/bash
# Tokens-only rows (GTOK_SENT_NO_PAD) of sent_lane_kernel with the token window leaving in bursts of 1 / 2 / 4 windows
# (-DGTOK_LANE_SECTOR_GROUPS: 16 / 32 / 64 bytes of a row per burst): time of the padded headline and WRITE_SIZE of the tokens-only
# flavour.  Variant libraries csrc/_ab/libgtok_sg{1,4}.so are built beforehand (hipcc -DGTOK_LANE_SECTOR_GROUPS=n on gtok_sent.hip);
# the in-tree library is the default (2).   gpurun -- 'bash profiles/tools/sector_groups.sh'
export TMPDIR=/tmp
out=gpurun_out/sector_groups; mkdir -p $out
ab=$PWD/glearning-benchmark_amd/csrc/_ab
for sg in 2 1 4; do
  lib=$ab/libgtok_sg$sg.so; [ $sg = 2 ] && lib=$PWD/glearning-benchmark_amd/csrc/libgtok.so
  echo "== sector groups $sg"
  GTOK_LIB=$lib timeout -k 10 200 python3 profiles/tools/time_lane_ab.py 2>&1 | grep -v amdgpu.ids || exit 1
  GTOK_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex sent_lane --output-format csv -d $out/w$sg -o p -- python3 bench.py --steps 5 --warmup 1 --rows unpadded --no-cpu-baseline --no-unpadded --no-boundary --no-sustained --no-ibtt > $out/w$sg.log 2>&1 || exit 1
  find $out -name '*_kernel_trace.csv' -delete
  python3 profiles/tools/pmc_summary.py $out/w$sg
done
