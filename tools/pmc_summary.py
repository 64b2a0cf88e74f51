#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel and counter, the mean value per dispatch."""
import csv, glob, sys, collections, json, os
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if "echo" in k}
print(json.dumps(out, indent=1))
