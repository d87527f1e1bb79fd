#!/bin/bash
# gtok_sent_packed on the ZINC-full-shaped corpus, K epochs per launch [16] of 16-bit rows: rocprofv3 kernel stats and the memory-side
# traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the plain walk, the walk that packs beside the slab, and the walk that packs
# alone (GTOK_SENT_PACK_ONLY: rows staged in 64 rows per resident wave).
#   gpurun -- 'K=16 bash profiles/tools/packed_counters.sh'
export TMPDIR=/tmp
out=gpurun_out/packed_counters; rm -rf $out; mkdir -p $out
K=${K:-16}
cat > $out/run.py <<PY
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")
G, ld, K = 249456, 176, $K
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
ids = torch.empty((K * G, ld), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
pk = gtok.ops.PackedRows(K * G, K * G * 96, True, dev)
mode = sys.argv[1]
for k in range(6):
    if mode == "plain":
        gtok.ops.sent(b, 37, 1024, 0, k * K, ld=ld, out=(ids, ln), pad=False, epochs=K, u16=True, **kw)
    elif mode == "beside":
        gtok.ops.sent(b, 37, 1024, 0, k * K, ld=ld, out=(ids, ln), pad=False, epochs=K, u16=True, packed=pk, **kw)
    else:
        gtok.ops.sent(b, 37, 1024, 0, k * K, ld=ld, epochs=K, u16=True, packed=pk, slab=False, **kw)
torch.cuda.synchronize()
assert mode == "plain" or (pk.fused and int(pk.status()) == 0)
PY
for mode in plain beside alone; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --kernel-include-regex sent_lane --output-format csv -d $out/stats_$mode -o s -- python3 $out/run.py $mode > $out/stats_$mode.log 2>&1 || { tail -5 $out/stats_$mode.log; exit 1; }
  for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    name=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex sent_lane --output-format csv -d $out/pmc_${mode}_$name -o p -- python3 $out/run.py $mode > $out/pmc_${mode}_$name.log 2>&1 || { tail -5 $out/pmc_${mode}_$name.log; exit 1; }
  done
done
python3 - $out $K <<'PY'
import collections, csv, glob, json, sys
out, K = sys.argv[1], int(sys.argv[2])
res = {}
for mode in ("plain", "beside", "alone"):
    r = {}
    for f in glob.glob(f"{out}/stats_{mode}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "sent_lane" in row["Name"]:
                r["kernel"] = row["Name"].split("(")[0]; r["calls"] = int(row["Calls"]); r["avg_us_per_launch"] = round(float(row["AverageNs"]) / 1e3, 1)
                r["us_per_epoch"] = round(float(row["AverageNs"]) / 1e3 / K, 2)
    for f in glob.glob(f"{out}/pmc_{mode}_*/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "sent_lane" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in agg.items():
            r[c + "_per_epoch"] = round(sum(v[-3:]) / len(v[-3:]) / K, 1)
    # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts half of wide coalesced reads on gfx950 (x 2)
    if "FETCH_SIZE_per_epoch" in r:
        r["hbm_read_MB_per_epoch"] = round(2 * r["FETCH_SIZE_per_epoch"] * 1024 / 1e6, 1)
    if "WRITE_SIZE_per_epoch" in r:
        r["hbm_write_MB_per_epoch"] = round(r["WRITE_SIZE_per_epoch"] * 1024 / 1e6, 1)
    res[mode] = r
print(json.dumps({"command": f"K={K} bash profiles/tools/packed_counters.sh (ZINC-full-shaped corpus, {K} epochs per launch, 16-bit rows, GTOK_SENT_NO_PAD; "
                             "plain = gtok_sent, beside = gtok_sent_packed + the slab, alone = gtok_sent_packed with GTOK_SENT_PACK_ONLY)", "per_mode": res}, indent=1))
PY
find $out -name '*_kernel_trace.csv' -delete; find $out -name '*_counter_collection.csv' -delete; find $out -name '*agent_info.csv' -delete
