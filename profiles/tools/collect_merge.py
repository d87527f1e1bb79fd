"""Turn gpurun_out/collect_<tag>/ (profiles/tools/collect.sh) into the committed summaries under profiles/<tag>/:
bench JSON lines, per-kernel stats CSVs, one PMC summary per workload, and profiles/pmc_traffic.json
(HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE KiB: gfx950 FETCH_SIZE counts half the fetched bytes,
/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, os, shutil, sys
import subprocess
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
src, dst = f"gpurun_out/collect_{tag}", f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
traffic = {}
for wl in ("zinc_full", "synth_er", "synth_mix", "zinc_subset"):
    if not os.path.exists(f"{src}/bench_{wl}.json"):
        continue
    shutil.copy(f"{src}/bench_{wl}.json", f"{dst}/bench_{wl}_final.json")
    for f in glob.glob(f"{src}/stats_{wl}/**/*_kernel_stats.csv", recursive=True):
        shutil.copy(f, f"{dst}/bench_{wl}_kernel_stats_final.csv")
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{src}/pmc_{wl}_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gtok" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    kern = {}
    for (k, c), v in sorted(agg.items()):
        kern.setdefault(k, {})[c] = {"calls": len(v), "mean": sum(v) / len(v)}
    for k, c in kern.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024)
    if not kern:
        continue
    json.dump({"command": "profiles/tools/collect.sh: rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 bench.py "
                          f"--steps 5 --warmup 1 --workload {wl} --no-cpu-baseline, one pass per counter group",
               "note": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; hbm_bytes_per_launch = (2*FETCH + WRITE) * 1024",
               "kernels": kern}, open(f"{dst}/pmc_summary_{wl}_final.json", "w"), indent=1)
    bench = json.loads(open(f"{src}/bench_{wl}.json").read().strip().splitlines()[-1])
    G, mode = bench["config"]["graphs_per_gpu"], bench["config"]["slab_width_mode"]
    for k, c in kern.items():
        if "hbm_bytes_per_launch" not in c:
            continue
        leg = "sent" if "sent" in k else ("ibtt" if "ibtt" in k else None)
        if leg:
            E = bench["config"].get("epochs_per_launch", 1)
            key = (f"{leg}:{wl}:{G}:{mode}" + (f":x{E}" if E > 1 else "")) if leg == "sent" else f"{leg}:{wl}:{G}"
            label = bench["roofline"]["kernel"] if leg == "sent" else bench.get("ibtt", {}).get("kernel")
            if not label or label.split("<")[0] not in k:      # another kernel of the same leg (the no-mirror run, decode): not the one the line reports
                continue
            traffic[key] = {"hbm_bytes_per_launch": c["hbm_bytes_per_launch"], "kernel": k.replace("void ", ""),
                            "kernel_label": label, "commit": commit, "source": f"{dst}/pmc_summary_{wl}_final.json"}
# the other row flavours of the headline kernel (bench.py --rows unpadded | u16 | u16padded): a small summary each
for rows, fname in (("unpadded", "nopad"), ("u16", "u16"), ("u16padded", "u16padded"), ("nopad", "nopad")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{src}/pmc_{rows}_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sent_lane_kernel" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    if not agg:
        continue
    kern = {}
    for (k, c), v in sorted(agg.items()):
        # the first launches of the run are padded int32 ones (slab-width probe: another instantiation); of the timed kernel
        # keep the later values
        kern.setdefault(k, {})[c] = {"calls": len(v), "mean_of_last_5": sum(v[-5:]) / len(v[-5:])}
    for k, c in kern.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"]["mean_of_last_5"] + c["WRITE_SIZE"]["mean_of_last_5"]) * 1024)
            c["write_bytes_per_launch"] = int(c["WRITE_SIZE"]["mean_of_last_5"] * 1024)
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-include-regex sent_lane -- python3 bench.py --steps 5 --warmup 1 "
                          f"--rows {rows} ... (the timed launches run in that flavour; the <.., false> / <.., true> instantiations are the int32 / 16-bit rows)",
               "kernels": kern}, open(f"{dst}/pmc_summary_zinc_full_{fname}.json", "w"), indent=1)
for extra in ("check_1m.txt", "time_r04.txt"):
    if os.path.exists(f"{src}/{extra}"):
        shutil.copy(f"{src}/{extra}", f"{dst}/{extra}")
old = json.load(open("profiles/pmc_traffic.json")) if os.path.exists("profiles/pmc_traffic.json") else {}
old = {k: v for k, v in old.items() if v.get("source", "").startswith(dst + "/")}      # entries of this round that were not re-collected
old.update(traffic)
traffic = old
json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
