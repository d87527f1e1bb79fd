"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (gtok kernels only)."""
import collections
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gtok" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            out.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
print(json.dumps(out, indent=1))
