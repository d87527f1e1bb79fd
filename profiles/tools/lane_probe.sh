#!/bin/bash
# Quick look at sent_lane_kernel on the ZINC-full-shaped corpus: launch time (HIP events), then one rocprofv3 PMC
# pass per counter group.  Usage (repo root):  gpurun --timeout 600 -- 'bash profiles/tools/lane_probe.sh <tag> [pmc]'
set -e -o pipefail
tag=${1:-probe}
out=gpurun_out/probe_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 200 python3 profiles/tools/time_sent_zinc.py > $out/time.txt 2>&1
cat $out/time.txt
if [ "$2" = "pmc" ]; then
  B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ibtt"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
    name=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$name -o p -- $B > $out/pmc_$name.log 2>&1
  done
  find $out -name '*_kernel_trace.csv' -delete; find $out -name '*_agent_info.csv' -delete
  python3 profiles/tools/pmc_summary.py $out > $out/pmc_summary.json
  python3 - $out/pmc_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, c in d.items():
    if "sent_lane" in k:
        print(k)
        for n, v in c.items():
            print(f"  {n:24s} {v:16.1f}")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            print(f"  HBM bytes (2*FETCH+WRITE)  {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e6:10.1f} MB")
PY
  find $out -name '*_counter_collection.csv' -delete
fi
