#!/bin/bash
# Collects the per-round evidence on the GPU box: bench JSON lines, rocprofv3 kernel stats, and the separate PMC
# passes (FETCH_SIZE, WRITE_SIZE, SQ instruction mix, SQ waits) for both workloads.  Usage (from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/tools/collect.sh r02'
# Everything lands in gpurun_out/collect_<tag>/; profiles/tools/collect_merge.py turns it into profiles/<tag>/.
set -o pipefail
tag=${1:-r04}
out=gpurun_out/collect_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 3"
ok=1       # after a step fails or is killed at its limit no further GPU step is started; what was collected is still trimmed below
step() { [ $ok = 1 ] || return 0; "$@" || { ok=0; echo "FAILED: $*"; }; }
run_to() { local o=$1; shift; "$@" > $o 2> ${o%.json}.err; }
WLS=${WLS:-zinc_full synth_er synth_mix zinc_subset}      # (WLS="zinc_subset" re-collects one workload)
# zinc_subset runs 24 epochs per launch: K = 48 steps are two launches; its counter passes time one launch of 24 behind a warm-up of one
steps_of() { [ $1 = zinc_subset ] && echo "--steps 48 --warmup 5" || echo "--steps 20 --warmup 3"; }
pmc_steps_of() { [ $1 = zinc_subset ] && echo "--steps 24 --warmup 24" || echo "--steps 5 --warmup 1"; }
for wl in $WLS; do
  extra=""; [ $wl = synth_mix ] && extra="--no-cpu-baseline"
  step run_to $out/bench_$wl.json timeout -k 10 600 python3 bench.py $(steps_of $wl) --workload $wl $extra
done
if [ "${BENCH_ONLY:-0}" = 1 ]; then du -sh $out; echo collected; exit 0; fi      # (the bench lines alone: after a change that no counter pass would see)
prof() { local log=$1; shift; "$@" > $log 2>&1; local rc=$?; find $out -name '*_kernel_trace.csv' -delete; return $rc; }
# counters are collected for OUR kernels only (--kernel-include-regex gtok): the corpora are sampled by thousands of torch
# launches, which made counter collection of synth_mix impractical before
for wl in $WLS; do
  # (the sustained leg stays in: it is part of the command that prints the bench line, and its launches are the steady state the K steps run in)
  step prof $out/stats_$wl.log timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o s -- python3 bench.py $(steps_of $wl) --workload $wl --no-cpu-baseline --no-unpadded --no-boundary
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"; do
    name=$(echo $grp | cut -d' ' -f1)
    step prof $out/pmc_${wl}_$name.log timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex gtok --output-format csv -d $out/pmc_${wl}_$name -o p -- python3 bench.py $(pmc_steps_of $wl) --workload $wl --no-cpu-baseline --no-unpadded --no-boundary --no-sustained
  done
done
# the other row flavours of the headline kernel: what reaches the L2's memory side when no pad tail is written (GTOK_SENT_NO_PAD),
# and with rows of 16-bit ids (GTOK_SENT_U16, with and without padding)
for rows in ${ROWS-unpadded u16 u16padded}; do
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    step prof $out/pmc_${rows}_$grp.log timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex sent_lane --output-format csv -d $out/pmc_${rows}_$grp -o p -- python3 bench.py --steps 5 --warmup 1 --rows $rows --no-cpu-baseline --no-unpadded --no-boundary --no-sustained --no-ibtt
  done
done
# beyond one round of resident waves: 1,000,000 molecules in one launch (bit-exact against the oracle, then timed), and the
# timing matrix of round 4 (row flavours x K epochs per launch x corpus sizes)
if [ "${SKIP_EXTRAS:-0}" != 1 ]; then      # (gpurun's limit is 20 minutes per call: SKIP_EXTRAS=1 leaves these two to a call of their own)
  step run_to $out/check_1m.txt timeout -k 10 500 python3 profiles/tools/check_1m.py
  step run_to $out/time_r04.txt timeout -k 10 400 python3 profiles/tools/time_r04.py
fi
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
