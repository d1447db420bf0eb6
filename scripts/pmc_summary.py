#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: scripts/pmc_summary.py gpurun_out/pmc_<tag> > profiles/<name>.csv"""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
names = sorted({c for k in agg for c in agg[k]})
w = csv.writer(sys.stdout); w.writerow(["kernel", "dispatches"] + names)
for k in sorted(agg):
    w.writerow([k, max(len(v) for v in calls[k].values())] + [("%.6g" % agg[k][c]) if c in agg[k] else "" for c in names])
