#!/bin/bash
# Collects the per-round evidence on the GPU box: bench JSON lines, rocprofv3 kernel stats, and the separate PMC
# passes (FETCH_SIZE, WRITE_SIZE, SQ instruction mix, SQ waits) for both workloads.  Usage (from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/tools/collect.sh r02'
# Everything lands in gpurun_out/collect_<tag>/; profiles/tools/collect_merge.py turns it into profiles/<tag>/.
set -e -o pipefail
tag=${1:-r02}
out=gpurun_out/collect_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 3"
timeout -k 10 400 $B > $out/bench_zinc_full.json 2> $out/bench_zinc_full.err
timeout -k 10 400 $B --workload synth_er > $out/bench_synth_er.json 2> $out/bench_synth_er.err
timeout -k 10 600 $B --workload synth_mix --no-cpu-baseline > $out/bench_synth_mix.json 2> $out/bench_synth_mix.err
timeout -k 10 400 $B --workload zinc_subset > $out/bench_zinc_subset.json 2> $out/bench_zinc_subset.err
for wl in zinc_full synth_er; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o s -- $B --workload $wl --no-cpu-baseline --no-unpadded > $out/stats_$wl.log 2>&1
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"; do
    name=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_${wl}_$name -o p -- python3 bench.py --steps 5 --warmup 1 --workload $wl --no-cpu-baseline --no-unpadded > $out/pmc_${wl}_$name.log 2>&1
  done
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
