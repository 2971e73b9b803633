"""Summarise a rocprofv3 --pmc run: per (kernel, counter) the number of dispatches and the mean value per dispatch.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [out.csv]
(rocprofv3 emits one row per dispatch x counter x instance dimension; rows of one dispatch are summed.)"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
per = defaultdict(float)  # (kernel, counter, dispatch) -> value
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
            k = re.sub(r"<.*$", "", k)
            per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
agg = defaultdict(list)
for (k, c, _), v in per.items():
    agg[(k, c)].append(v)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write("kernel,counter,dispatches,mean_per_dispatch\n")
for (k, c), vs in sorted(agg.items()):
    out.write(f"\"{k}\",{c},{len(vs)},{sum(vs)/len(vs):.6g}\n")
