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


for sub, name in (("trace", "kernel_stats.csv"), ("trace_serial", "kernel_stats_serial.csv")):
    stats = []
    for f in glob.glob(os.path.join(root, sub, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        stats = [r for r in rows if "echo" in r["Name"]]
    if stats:
        with open(os.path.join(root, name), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(stats[0].keys()))
            w.writeheader()
            for r in stats:
                r["Name"] = r["Name"].split("(")[0]
                w.writerow(r)
        print(name)
        for r in stats:
            print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]}%')

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

# SQ counters: per kernel and counter the mean per launch, plus the derived issue-busy and LDS-busy fractions
sq = {}
for sub in ("sq1", "sq2"):
    for k, cs in counter_means(sub).items():
        if "echo" in k:
            sq.setdefault(k, {}).update(cs)
for k, c in sq.items():
    d = {}
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
        # SQ_BUSY_CYCLES sums over the 32 shader engines; SQ_ACTIVE_INST_VALU counts 4-cycle issue slots over 1024 SIMDs
        kernel_cycles = c["SQ_BUSY_CYCLES"] / 32.0
        d["valu_issue_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / kernel_cycles
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_array_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / kernel_cycles
    if c.get("SQ_LDS_IDX_ACTIVE") and c.get("SQ_LDS_BANK_CONFLICT") is not None:
        d["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
        d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    c["derived"] = d
if sq:
    json.dump({"units": "mean per launch; derived: valu_issue_busy_frac = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / "
                        "(SQ_BUSY_CYCLES / 32 shader engines), lds_array_busy_frac = SQ_LDS_IDX_ACTIVE / 256 CUs / same",
               "kernels": sq}, open(os.path.join(root, "sq_counters.json"), "w"), indent=1)
    for k, c in sq.items():
        print(k[:60], {a: round(b, 3) for a, b in c["derived"].items()})

