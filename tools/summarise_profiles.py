#!/usr/bin/env python3
"""Condenses a tools/collect_profiles.sh output directory into the small files kept under profiles/.

Writes <dir>/kernel_stats.csv (rocprofv3 --stats kernel summary, the library's kernels only), <dir>/hbm_traffic.json
(per kernel: mean FETCH_SIZE / WRITE_SIZE per launch, converted to bytes) and prints both.

Counter units: rocprofv3 reports FETCH_SIZE and WRITE_SIZE in kilobytes.  On gfx950 FETCH_SIZE under-counts by a
factor of two against known copy sizes (64-byte requests counted as 32); WRITE_SIZE matches.  Both corrections
are applied here and recorded in the output.
"""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
FETCH_FIX = 2.0


def counter_means(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


stats = []
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    stats = [r for r in rows if "echo" in r["Name"]]
if stats:
    with open(os.path.join(root, "kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(stats[0].keys()))
        w.writeheader()
        for r in stats:
            r["Name"] = r["Name"].split("(")[0]
            w.writerow(r)
    for r in stats:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]}%')

fetch, write = counter_means("fetch"), counter_means("write")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if "echo" not in k:
        continue
    fb = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * FETCH_FIX
    wb = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    traffic[k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
json.dump({"units": "bytes per launch; FETCH_SIZE KB x1024 x2 (gfx950 correction), WRITE_SIZE KB x1024",
           "kernels": traffic}, open(os.path.join(root, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
