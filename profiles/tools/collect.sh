#!/bin/bash
# Collects the per-round evidence on the GPU box: bench JSON lines, rocprofv3 kernel stats, and the separate PMC
# passes (FETCH_SIZE, WRITE_SIZE, SQ instruction mix, SQ waits) for both workloads.  Usage (from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/tools/collect.sh r02'
# Everything lands in gpurun_out/collect_<tag>/; profiles/tools/collect_merge.py turns it into profiles/<tag>/.
set -o pipefail
tag=${1:-r03}
out=gpurun_out/collect_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 3"
ok=1       # after a step fails or is killed at its limit no further GPU step is started; what was collected is still trimmed below
step() { [ $ok = 1 ] || return 0; "$@" || { ok=0; echo "FAILED: $*"; }; }
run_to() { local o=$1; shift; "$@" > $o 2> ${o%.json}.err; }
step run_to $out/bench_zinc_full.json timeout -k 10 400 $B
step run_to $out/bench_synth_er.json timeout -k 10 400 $B --workload synth_er
step run_to $out/bench_synth_mix.json timeout -k 10 600 $B --workload synth_mix --no-cpu-baseline
step run_to $out/bench_zinc_subset.json timeout -k 10 400 $B --workload zinc_subset
prof() { local log=$1; shift; "$@" > $log 2>&1; local rc=$?; find $out -name '*_kernel_trace.csv' -delete; return $rc; }
# counters are collected for OUR kernels only (--kernel-include-regex gtok): the corpora are sampled by thousands of torch
# launches, which made counter collection of synth_mix impractical before
for wl in zinc_full synth_er synth_mix zinc_subset; do
  # (the sustained leg stays in: it is part of the command that prints the bench line, and its launches are the steady state the K steps run in)
  step prof $out/stats_$wl.log timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o s -- $B --workload $wl --no-cpu-baseline --no-unpadded --no-boundary
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"; do
    name=$(echo $grp | cut -d' ' -f1)
    step prof $out/pmc_${wl}_$name.log timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex gtok --output-format csv -d $out/pmc_${wl}_$name -o p -- python3 bench.py --steps 5 --warmup 1 --workload $wl --no-cpu-baseline --no-unpadded --no-boundary --no-sustained
  done
done
# the tokens-only flavour of the headline kernel (GTOK_SENT_NO_PAD): what reaches the L2's memory side when no pad tail is written
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  step prof $out/pmc_nopad_$grp.log timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex sent_lane --output-format csv -d $out/pmc_nopad_$grp -o p -- python3 bench.py --steps 5 --warmup 1 --rows unpadded --no-cpu-baseline --no-unpadded --no-boundary --no-sustained --no-ibtt
done
# keep what collect_merge.py reads: our kernels' counter rows and the stats tables (gpurun returns <= 64 MiB)
find $out -name '*_kernel_trace.csv' -delete; find $out -name '*_agent_info.csv' -delete
python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "gtok" in r["Kernel_Name"]]
    with open(f, "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys() if rows else ["Kernel_Name"])
        w.writeheader(); w.writerows(keep)
PY
du -sh $out
echo collected
