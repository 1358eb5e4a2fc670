"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per
dispatch. Usage: python tools/pmc_summary.py <dir-with-csv> [<dir> ...] > profiles/xxx.json"""
import csv, glob, json, os, sys
from collections import defaultdict
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            short = k.replace("(anonymous namespace)::", "")[:110]
            for c, vals in cs.items():
                out.setdefault(short, {})[c] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
print(json.dumps(out, indent=1, sort_keys=True))
