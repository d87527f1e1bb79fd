#!/bin/bash
# Same bench command under rocprofv3 --pmc with two libraries (GTOK_LIB baseline vs in-tree): core-clock cycles and instruction
# counts per launch of the lane kernel, beside its duration - tells a real difference in work from a difference in clocks.
# Usage: gpurun -- 'bash profiles/tools/ab_counters.sh glearning-benchmark_amd/csrc/_ab/libgtok_r03.so'
export TMPDIR=/tmp
base=$1
out=gpurun_out/ab_counters; rm -rf $out; mkdir -p $out
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ibtt --no-unpadded --no-boundary --no-sustained"
for tag in ${TAGS:-base new}; do
  lib=$PWD/glearning-benchmark_amd/csrc/libgtok.so; [ $tag = base ] && lib=$PWD/$base
  for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
    name=$(echo $grp | cut -d' ' -f1)
    GTOK_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex sent_lane --output-format csv -d $out/${tag}_$name -o p -- $B > $out/${tag}_$name.log 2>&1 || { tail -5 $out/${tag}_$name.log; exit 1; }
  done
  GTOK_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --kernel-include-regex sent_lane --output-format csv -d $out/${tag}_stats -o p -- $B > $out/${tag}_stats.log 2>&1
  echo "== $tag"; python3 profiles/tools/pmc_summary.py $out/${tag}_GRBM_GUI_ACTIVE $out/${tag}_SQ_INSTS_VALU $out/${tag}_SQ_WAIT_INST_ANY
  find $out -name '*kernel_stats.csv' -path "*${tag}_stats*" | head -1 | xargs cat | cut -d, -f1-8 | head -3
done
find $out -name '*_kernel_trace.csv' -delete; find $out -name '*_counter_collection.csv' -delete; find $out -name '*agent_info.csv' -delete
