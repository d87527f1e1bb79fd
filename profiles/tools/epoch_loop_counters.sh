#!/bin/bash
# Counters of the dataset classes' epoch on ZINC-full (E epochs per launch [4], 16-bit rows, no padding): how close to its
# vector-issue bound the lane kernel runs when staging and padding overlap with the walks.
#   gpurun -- 'E=16 STEPS=48 bash profiles/tools/epoch_loop_counters.sh'
export TMPDIR=/tmp
out=gpurun_out/epoch_loop_counters; rm -rf $out; mkdir -p $out
B="python3 bench.py --steps ${STEPS:-16} --warmup ${E:-4} --rows u16 --epochs-per-launch ${E:-4} --no-cpu-baseline --no-ibtt --no-unpadded --no-boundary --no-sustained"
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex sent_lane --output-format csv -d $out/$name -o p -- $B > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --kernel-include-regex sent_lane --output-format csv -d $out/stats -o p -- $B > $out/stats.log 2>&1
python3 - $out <<'PY'
import collections, csv, glob, json, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "true>(gtok::SentLaneArgs)" in r["Kernel_Name"]:          # the 16-bit instantiation
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {c: {"calls": len(v), "mean_of_last_4": sum(v[-4:]) / len(v[-4:])} for c, v in sorted(agg.items())}
import os
E = os.environ.get("E", "4")
print(json.dumps({"command": f"rocprofv3 --kernel-trace --pmc <group> --kernel-include-regex sent_lane -- python3 bench.py --steps {os.environ.get('STEPS', '16')} --warmup {E} --rows u16 "
                             f"--epochs-per-launch {E} ... (sent_lane_kernel<true, 4, true, true, true>: {E} epochs of ZINC-full per launch, 16-bit rows, no padding)",
                  f"counters_per_launch_of_{E}_epochs": out}, indent=1))
PY
find $out -name '*kernel_stats.csv' | head -1 | xargs cat | cut -d, -f1-8 | head -4
find $out -name '*_kernel_trace.csv' -delete; find $out -name '*_counter_collection.csv' -delete; find $out -name '*agent_info.csv' -delete
